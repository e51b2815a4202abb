#!/bin/bash
# same-box A/B of two builds of the library on the levels-mode bench: tools/ab_levels.sh old.so [rounds]
old=$1; rounds=${2:-2}
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "fuzzy %.2f ms  jaccard %.3f ms" % (d["fuzzy_match"]["ms_per_3_grids"], d["intersection_vs_union"]["ms_per_3_grids"]))'
for r in $(seq $rounds); do
  NSM_HIP_LIBRARY=$old timeout -k 10 300 python tools/bench_levels.py --rows 100000 --steps 5 2>/dev/null | python -c "$show" old || exit 1
  timeout -k 10 300 python tools/bench_levels.py --rows 100000 --steps 5 2>/dev/null | python -c "$show" new || exit 1
done
