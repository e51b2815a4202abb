#!/bin/bash
# same-box comparison of builds of the library on the levels-mode bench:
#   tools/ab_levels.sh [--rows N] lib_a.so lib_b.so ...      ("-" = the in-tree build)
rows=100000
if [ "$1" = "--rows" ]; then rows=$2; shift 2; fi
show='import sys,json; d=json.loads(sys.stdin.read()); print("%-28s fuzzy %8.2f ms  jaccard %7.3f ms" % (sys.argv[1], d["fuzzy_match"]["ms_per_3_grids"], d["intersection_vs_union"]["ms_per_3_grids"]))'
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset NSM_HIP_LIBRARY; else export NSM_HIP_LIBRARY=$lib; fi
    timeout -k 10 600 python tools/bench_levels.py --rows $rows --steps 4 2>/dev/null | python -c "$show" "$(basename $lib)" || exit 1
  done
done
