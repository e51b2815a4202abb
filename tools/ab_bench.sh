#!/bin/bash
# same-box comparison of builds of the library on bench.py: tools/ab_bench.sh [bench args --] lib_a.so lib_b.so ... ("-" = in-tree)
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
if [ "$1" = "--" ]; then shift; else set -- "${args[@]}"; args=(); fi
show='import sys,json; d=json.loads(sys.stdin.read()); print("%-24s step %8.4f ms  kernel %8.4f ms  exhaustive %8.3f ms  hits %s" % (sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms"], d["exhaustive"]["kernel_ms"], d["config"]["hits_per_rank"]))'
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset NSM_HIP_LIBRARY; else export NSM_HIP_LIBRARY=$lib; fi
    timeout -k 10 600 python bench.py --no-cpu-baseline "${args[@]}" 2>/dev/null | python -c "$show" "$(basename $lib)" || exit 1
  done
done
