#!/bin/bash
# Re-measure everything kept under profiles/ on the GPU box (run through gpurun from the repo root):
#   tools/refresh_profiles.sh <stamp> [part...]        parts: pmc bench trace (default: all, in this order)   -> gpurun_out/final/*
# <stamp> = `git describe --always --dirty` of the tree that was sent (the box has no .git).
#   bench  the JSON lines of bench.py (c2 c3 c4 c5 term) and of the levels / RAW-terms tool benches
#   trace  rocprofv3 --kernel-trace --stats of the default bench.py run, of c5 and of term
#   pmc    hardware counters per kernel (tools/pmc_collect.sh: two SQ passes + FETCH_SIZE + WRITE_SIZE, separate passes)
set -e -o pipefail
export TMPDIR=/tmp
stamp=${1:-unknown}; shift || true
parts=${@:-pmc bench trace}  # pmc first: its pmc_<w>.json are copied into profiles/ of this tree, so that the bench lines that follow carry a roofline that matches the sources
out=gpurun_out/final
mkdir -p $out
for part in $parts; do
  case $part in
  bench)
    # the default run carries every workload as a sub-record (c3 headline + c2 c2low c4 c5 c5w term)
    python3 bench.py > $out/all_bench.json 2> $out/all_bench.err; echo "default bench (all workloads) done"
    python3 tools/bench_levels.py --rows 100000 --steps 5 --check 300 > $out/levels_bench.json 2> $out/levels_bench.err; echo "levels bench done"
    python3 tools/bench_levels.py --rows 100000 --steps 5 --threshold 0.1 > $out/levels01_bench.json 2> $out/levels01_bench.err; echo "levels bench at 0.1 done"
    python3 tools/bench_terms.py --rows 50000 --check 200 > $out/terms_bench.json 2> $out/terms_bench.err; echo "terms bench done"
    ;;
  trace)
    for w in c2 c3 c5 c5w term; do
      steps=10; [ $w = c5 ] && steps=2; [ $w = c5w ] && steps=1
      rm -rf $out/prof_$w
      rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -- python3 bench.py --workload $w --steps $steps --warmup 2 --no-cpu-baseline > $out/prof_$w.log 2>&1
      cp $(ls $out/prof_$w/*/*_kernel_stats.csv | head -1) $out/${w}_kernel_stats.csv
      rm -rf $out/prof_$w
      echo "$w kernel trace done"
    done
    ;;
  pmc)
    tools/pmc_collect.sh $out $stamp c2 c2low c3 c4 c5 c5w term levels
    cp $out/pmc_*.json profiles/
    ;;
  esac
done
ls -la $out
