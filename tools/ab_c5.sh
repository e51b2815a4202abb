#!/bin/bash
# same-box comparison of builds of the library on configs[4] (bench.py --workload c5):
#   tools/ab_c5.sh lib_a.so lib_b.so ...      ("-" = the in-tree build)   ->  name, step ms, fuzzy grids ms, Jaccard grids ms
show='import sys,json; d=json.loads(sys.stdin.read()); print("%-24s step %7.1f ms  fuzzy %7.1f ms  jaccard %6.1f ms" % (sys.argv[1], d["ms_per_step"], d["config"]["fuzzy_grids_ms_per_step"], d["config"]["jaccard_grids_ms_per_step"]))'
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset NSM_HIP_LIBRARY; else export NSM_HIP_LIBRARY=$lib; fi
  timeout -k 10 300 python bench.py --workload c5 --steps 3 --no-cpu-baseline 2>/dev/null | python -c "$show" "$(basename $lib)" || exit 1
done
