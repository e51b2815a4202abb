#!/bin/bash
# same-box comparison of builds of the library on configs[2] (bench.py --workload c3):
#   tools/ab_c3.sh lib_a.so lib_b.so ...      ("-" = the in-tree build)   ->  name, step ms, kernel ms
show='import sys,json; d=json.loads(sys.stdin.read()); print("%-28s step %7.3f ms  kernel %7.3f ms  hits %d" % (sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["hits_per_rank"]))'
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset NSM_HIP_LIBRARY; else export NSM_HIP_LIBRARY=$PWD/$lib; fi
  timeout -k 10 300 python bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$show" "$(basename $lib)" || exit 1
done
