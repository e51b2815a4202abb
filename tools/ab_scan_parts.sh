#!/bin/bash
# where the one-word scan's time goes: variant builds with parts compiled out (csrc/indel_levels_park.hpp NSM_X_*) on 3 x 100k^2
# digit-token grids at 0.7 -> gpurun_out/scanparts/out.txt
set -e
mkdir -p gpurun_out/scanparts
V=napkon-string-matching_amd/csrc/variants
run() {
  echo "== $1" >> gpurun_out/scanparts/out.txt
  NSM_HIP_LIBRARY=$2 timeout -k 10 300 python tools/bench_levels.py --rows 100000 --threshold 0.7 --steps 3 $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  fuzzy ms/3 grids %.2f hits %s' % (d['fuzzy_match']['ms_per_3_grids'], d['fuzzy_match']['hits']))" >> gpurun_out/scanparts/out.txt
}
run base $PWD/napkon-string-matching_amd/csrc/libnsm_hip.so
run EMPTY $PWD/$V/libnsm_xs_EMPTY.so
run HONLY $PWD/$V/libnsm_xs_HONLY.so
run base-words $PWD/napkon-string-matching_amd/csrc/libnsm_hip.so --words
run HONLY-words $PWD/$V/libnsm_xs_HONLY.so --words
cat gpurun_out/scanparts/out.txt
