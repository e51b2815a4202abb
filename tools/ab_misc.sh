#!/bin/bash
# fuzz campaigns at the current kernels -> gpurun_out/fuzz/
set -e
mkdir -p gpurun_out/fuzz
timeout -k 10 560 python tools/fuzz_parity.py --seconds 480 --seed 40000 > gpurun_out/fuzz/parity.txt 2>&1 || { tail -20 gpurun_out/fuzz/parity.txt; exit 1; }
tail -2 gpurun_out/fuzz/parity.txt
timeout -k 10 460 python tools/fuzz_api.py --seconds 380 --seed 50000 > gpurun_out/fuzz/api.txt 2>&1 || { tail -20 gpurun_out/fuzz/api.txt; exit 1; }
tail -2 gpurun_out/fuzz/api.txt
