#!/bin/bash
# Re-measure everything kept under profiles/ on the GPU box (run through gpurun from the repo root):
#   tools/refresh_profiles.sh        -> gpurun_out/final/*
# bench JSON lines, rocprofv3 kernel-trace stats, and HBM traffic counters (FETCH_SIZE / WRITE_SIZE in
# separate --pmc passes, no other trace domains).
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/final
rm -rf $out && mkdir -p $out
python bench.py > $out/c2_bench.json 2> $out/c2_bench.err
echo "c2 bench done"
python bench.py --workload c3 > $out/c3_bench.json 2> $out/c3_bench.err
echo "c3 bench done"
python tools/bench_levels.py --rows 100000 --steps 5 --check 300 > $out/levels_bench.json 2> $out/levels_bench.err
echo "levels bench done"
python tools/bench_terms.py --rows 50000 --check 200 > $out/terms_bench.json 2> $out/terms_bench.err
echo "terms bench done"
for w in c2 c3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -- python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $out/prof_$w.log 2>&1
  cp $(ls $out/prof_$w/*/*_kernel_stats.csv | head -1) $out/${w}_kernel_stats.csv
  echo "$w kernel trace done"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch_$w -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_fetch_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write_$w -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_write_$w.log 2>&1
  python tools/pmc_summary.py $out/pmc_fetch_$w $out/pmc_write_$w > $out/${w}_hbm_pmc.txt
  echo "$w pmc done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_levels -- python tools/bench_levels.py --rows 100000 --steps 3 > $out/prof_levels.log 2>&1
cp $(ls $out/prof_levels/*/*_kernel_stats.csv | head -1) $out/levels_kernel_stats.csv
echo "levels kernel trace done"
# keep the merge small: only the summaries travel back
rm -rf $out/prof_* $out/pmc_*
ls -la $out
