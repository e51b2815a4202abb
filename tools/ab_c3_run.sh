#!/bin/bash
# one GPU call: parity of a C3 kernel build ($1, "-" = in-tree), then tools/ab_c3.sh over all the builds given -> gpurun_out/c3ab/
set -e
mkdir -p gpurun_out/c3ab
if [ "$1" != "-" ]; then export NSM_HIP_LIBRARY=$PWD/$1; fi
timeout -k 10 600 python -m pytest tests/test_gpu_grids.py -q -x -k "indel_raw" > gpurun_out/c3ab/tests.txt 2>&1 || { tail -30 gpurun_out/c3ab/tests.txt; exit 1; }
unset NSM_HIP_LIBRARY
tail -2 gpurun_out/c3ab/tests.txt
bash tools/ab_c3.sh "$@" > gpurun_out/c3ab/out.txt 2>&1
cat gpurun_out/c3ab/out.txt
