V=napkon-string-matching_amd/csrc/variants
for lib in - $V/libnsm_sl4k.so $V/libnsm_sl16k.so $V/libnsm_sl32k.so $V/libnsm_pm16.so $V/libnsm_pm32.so -; do
  if [ "$lib" = "-" ]; then unset NSM_HIP_LIBRARY; else export NSM_HIP_LIBRARY=$lib; fi
  timeout -k 10 300 python bench.py --workload c5 --steps 3 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"],1), round(d["config"]["fuzzy_grids_ms_per_step"],1), round(d["config"]["jaccard_grids_ms_per_step"],1))' $(basename $lib) || exit 1
done
