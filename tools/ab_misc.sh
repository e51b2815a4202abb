#!/bin/bash
# scratch GPU call -> gpurun_out/misc/
set -e
mkdir -p gpurun_out/misc
timeout -k 10 1100 python -m pytest tests/test_gpu_builders.py tests/test_gpu_grids.py tests/test_gpu_wide.py tests/test_gpu_api.py -q -x > gpurun_out/misc/tests.txt 2>&1 || { tail -40 gpurun_out/misc/tests.txt; exit 1; }
tail -3 gpurun_out/misc/tests.txt
