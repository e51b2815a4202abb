#!/bin/bash
# scratch GPU call -> gpurun_out/misc/
set -e
mkdir -p gpurun_out/misc
timeout -k 10 600 python -m pytest tests/test_gpu_wide.py -q -x > gpurun_out/misc/tests.txt 2>&1 || { tail -40 gpurun_out/misc/tests.txt; exit 1; }
tail -2 gpurun_out/misc/tests.txt
timeout -k 10 300 python tools/fuzz_parity.py --seconds 200 --seed 70000 --family wide > gpurun_out/misc/fuzz_wide.txt 2>&1 || { tail -20 gpurun_out/misc/fuzz_wide.txt; exit 1; }
tail -1 gpurun_out/misc/fuzz_wide.txt
