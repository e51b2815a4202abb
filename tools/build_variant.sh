#!/bin/bash
# Build a variant of libnsm_hip.so with extra compiler flags for ONE source (same-box A/B runs, tools/ab_*.sh):
#   tools/build_variant.sh <name> <source.hip> "<flags>"   ->  napkon-string-matching_amd/csrc/variants/libnsm_<name>.so
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/../napkon-string-matching_amd/csrc"
make -s -j8 > /dev/null
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wno-unused-command-line-argument \
  $flags -c $src -o variants/${name}_${src%.hip}.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs variants/${name}_${src%.hip}.o -o variants/libnsm_${name}.so
rm variants/${name}_${src%.hip}.o
echo variants/libnsm_${name}.so
