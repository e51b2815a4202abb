#!/bin/bash
# copy what tools/refresh_profiles.sh measured (gpurun_out/final) into profiles/ under this round's names
set -e
r=${1:-r01}; f=gpurun_out/final
for w in c2 c3; do
  cp $f/${w}_bench.json profiles/${r}_${w}_bench.json
  cp $f/${w}_kernel_stats.csv profiles/${r}_${w}_kernel_stats.csv
  cp $f/${w}_hbm_pmc.txt profiles/${r}_${w}_hbm_pmc.txt
done
cp $f/levels_bench.json profiles/${r}_levels_bench.json
cp $f/levels_kernel_stats.csv profiles/${r}_levels_kernel_stats.csv
cp $f/terms_bench.json profiles/${r}_terms_bench.json
for w in c4 c5; do [ -s $f/${w}_bench.json ] && cp $f/${w}_bench.json profiles/${r}_${w}_bench.json; done
python tools/make_traffic.py c2 profiles/${r}_c2_hbm_pmc.txt 'jaccard_raw_kernel<16, true>'
python tools/make_traffic.py c3 profiles/${r}_c3_hbm_pmc.txt 'indel_raw_kernel<true>'
