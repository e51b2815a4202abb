#!/bin/bash
# copy what tools/refresh_profiles.sh measured (gpurun_out/final) into profiles/ under this round's names
set -e
r=${1:-r04}; f=gpurun_out/final
for w in all c2 c2low c3 c4 c5 c5w term levels levels01 terms; do
  [ -s $f/${w}_bench.json ] && cp $f/${w}_bench.json profiles/${r}_${w}_bench.json
  [ -s $f/${w}_kernel_stats.csv ] && cp $f/${w}_kernel_stats.csv profiles/${r}_${w}_kernel_stats.csv
  [ -s $f/${w}_sq_pmc.txt ] && cp $f/${w}_sq_pmc.txt profiles/${r}_${w}_sq_pmc.txt
  [ -s $f/pmc_${w}.json ] && cp $f/pmc_${w}.json profiles/pmc_${w}.json
done
ls profiles
