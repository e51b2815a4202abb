#!/bin/bash
# same-box experiments on the one-word fuzzy levels split path (scan -> queue -> finish) against the fused park kernel:
#   tools/ab_split.sh        -> gpurun_out/split/ab.txt
# needs these variant builds (made here, they travel to the GPU box with the tree):
#   tools/build_variant.sh scanstats indel_levels.hip "-DNSM_SCAN_STATS"
#   tools/build_variant.sh thr0      indel_levels.hip "-DNSM_SPLIT_MIN_THRESHOLD=0.0"
#   tools/build_variant.sh w5        indel_levels.hip "-DNSM_SPLIT_WAVES=5"      (the default now; 4 / 6 for the comparison)
# results of the round's runs: profiles/r03_split_ab.txt
V=napkon-string-matching_amd/csrc/variants
out=gpurun_out/split; mkdir -p $out
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-44s fuzzy %8.2f ms / 3 grids  hits %s %s" % (sys.argv[1], d["fuzzy_match"]["ms_per_3_grids"], d["fuzzy_match"]["hits"], json.dumps(d.get("scan_stats_first_grid", ""))))'
lev() {  # label, lib, threshold, flags, extra
  if [ "$2" = "-" ]; then unset NSM_HIP_LIBRARY; else export NSM_HIP_LIBRARY=$V/libnsm_$2.so; fi
  timeout -k 10 200 python3 tools/bench_levels.py --rows 100000 --steps 3 --threshold $3 --indel-flags $4 $5 2>/dev/null | python3 -c "$show" "$1" || exit 1
}
{
lev "split 0.7" - 0.7 1
lev "fused 0.7" - 0.7 17
lev "split 0.7 (stats build)" scanstats 0.7 1 --scan-stats
lev "split 0.6 (min threshold 0)" thr0 0.6 1
lev "fused 0.6" - 0.6 17
lev "split 0.5 (min threshold 0)" thr0 0.5 1
lev "fused 0.5" - 0.5 17
lev "split 0.8" - 0.8 1
lev "fused 0.8" - 0.8 17
unset NSM_HIP_LIBRARY
bash tools/ab_c5.sh - $V/libnsm_w5.so
# (round 4: the queue is the caller's workspace -- python3 bench.py --workload c5 --split-workspace-mb 512 | 8192)
} 2>&1 | tee $out/ab.txt
