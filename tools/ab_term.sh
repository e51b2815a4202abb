#!/bin/bash
# same-box comparison of builds of the library on the term workload (bench.py --workload term):
#   tools/ab_term.sh [--threshold T] lib_a.so lib_b.so ...      ("-" = the in-tree build)
thr=""
if [ "$1" = "--threshold" ]; then thr="--threshold $2"; shift 2; fi
show='import sys,json; d=json.loads(sys.stdin.read()); print("%-28s step %7.2f ms  kernel %7.2f ms  hits %d" % (sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["hits_per_rank"]))'
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset NSM_HIP_LIBRARY; else export NSM_HIP_LIBRARY=$lib; fi
  timeout -k 10 300 python bench.py --workload term --steps 5 --warmup 2 --no-cpu-baseline $thr 2>/dev/null | python -c "$show" "$(basename $lib) $thr" || exit 1
done
