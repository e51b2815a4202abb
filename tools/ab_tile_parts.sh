set -e
mkdir -p gpurun_out/tileparts
V=napkon-string-matching_amd/csrc/variants
run() { # name lib extra
  echo "== $1" >> gpurun_out/tileparts/out.txt
  NSM_HIP_LIBRARY=$2 timeout -k 10 300 python tools/bench_levels.py --rows 100000 --words --threshold 0.5 --indel-flags 257 --steps 2 $3 >> gpurun_out/tileparts/out.txt 2>&1
}
run base napkon-string-matching_amd/csrc/libnsm_hip.so
run stats $V/libnsm_stats.so --tile-stats
for v in NODENSE NOWIDE NOAFTER NONEED NOBUILD; do run $v $V/libnsm_x_$v.so; done
