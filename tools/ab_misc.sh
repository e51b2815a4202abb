#!/bin/bash
# scratch GPU call -> gpurun_out/misc/
set -e
mkdir -p gpurun_out/misc
NSM_HIP_LIBRARY=$PWD/napkon-string-matching_amd/csrc/variants/libnsm_pf2.so timeout -k 10 600 python -m pytest tests/test_gpu_grids.py -q -x -k "global_index or jaccard_raw" > gpurun_out/misc/tests.txt 2>&1 || { tail -40 gpurun_out/misc/tests.txt; exit 1; }
tail -2 gpurun_out/misc/tests.txt
for rep in 1 2; do
for w in c4 c2low c2; do
for lib in napkon-string-matching_amd/csrc/libnsm_hip.so napkon-string-matching_amd/csrc/variants/libnsm_pf2.so; do
NSM_HIP_LIBRARY=$PWD/$lib timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/misc/$w.json 2> gpurun_out/misc/$w.err
python -c "import json; d=json.load(open('gpurun_out/misc/$w.json')); r=d['roofline']; print('$w $(basename $lib)', d['ms_per_step'], r['kernel_ms'], d['config']['hits_per_rank'])"
done; done; done
