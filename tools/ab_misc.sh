#!/bin/bash
# scratch GPU call -> gpurun_out/misc/
set -e
mkdir -p gpurun_out/misc
timeout -k 10 900 python -m pytest tests/test_gpu_builders.py tests/test_gpu_grids.py tests/test_gpu_bench.py -q -x -k "str_table or indel_raw or indel_known or c3" > gpurun_out/misc/tests.txt 2>&1 || { tail -40 gpurun_out/misc/tests.txt; exit 1; }
tail -3 gpurun_out/misc/tests.txt
timeout -k 10 300 python bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/misc/c3.json 2> gpurun_out/misc/c3.err
python -c "import json; d=json.load(open('gpurun_out/misc/c3.json')); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['exhaustive']['ms_per_step'], d['config']['hits_per_rank'])"
