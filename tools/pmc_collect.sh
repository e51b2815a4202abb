#!/bin/bash
# Hardware-counter passes for the kernels bench.py reports (run on the GPU box through gpurun, from the repo root):
#   tools/pmc_collect.sh <outdir> <stamp> [workloads...]      workloads default: c2 c3 levels term
# Per workload: one SQ pass (8 SQ slots + GRBM_GUI_ACTIVE), a second SQ pass, and FETCH_SIZE / WRITE_SIZE in
# separate passes (MI355X_MICROARCH.md: TCC slots), each with --kernel-trace only (no other trace domains).
# The program stands directly behind `--`.  Summaries: <outdir>/<workload>_sq_pmc.txt and pmc_<workload>.json.
set -e -o pipefail
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc}; stamp=${2:-unknown}; shift 2 || true
loads=${@:-c2 c2low c3 levels term}
mkdir -p $out
SQ_A="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
SQ_B="SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
for w in $loads; do
  case $w in
    levels) cmd="python3 tools/bench_levels.py --rows 100000 --steps 2";;
    c5) cmd="python3 bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline";;
    c5w) cmd="python3 bench.py --workload c5w --steps 1 --warmup 1 --no-cpu-baseline --no-extras";;
    c4) cmd="python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline";;
    *) cmd="python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline";;
  esac
  k=0
  for set in "$SQ_A" "$SQ_B" "FETCH_SIZE" "WRITE_SIZE"; do
    k=$((k + 1))
    rm -rf $out/raw_${w}_$k
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/raw_${w}_$k -- $cmd > $out/raw_${w}_$k.log 2>&1
    echo "$w pass $k done"
  done
  python3 tools/pmc_to_json.py $w $stamp $out $out/raw_${w}_1 $out/raw_${w}_2 $out/raw_${w}_3 $out/raw_${w}_4
  rm -rf $out/raw_${w}_?
done
ls -la $out
