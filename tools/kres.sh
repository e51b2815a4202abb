#!/bin/bash
# Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage):
#   tools/kres.sh napkon-string-matching_amd/csrc/indel_levels.hip [extra hipcc flags]
src=$1; shift
dir=$(dirname $src)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$dir/../../include -I$dir \
  -Wno-unused-command-line-argument -c $src -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
sed -E 's/ *\[-Rpass-analysis=kernel-resource-usage\]//' |
awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /SGPRs:/ {s=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {o=$NF}
     /LDS Size/ {printf "%s vgpr %s sgpr %s scratch %s occ %s lds %s\n", name, v, s, sc, o, $NF}' | c++filt | sed -E 's/\(int const\*.*\) vgpr/ vgpr/; s/^void //'
