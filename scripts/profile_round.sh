#!/bin/bash
# Per-round profile of the bench command (run ON the GPU box from the repo root):
#   bash scripts/profile_round.sh r01
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (same command line as the driver's, shorter run)
# 2. separate --pmc passes (never combined with tracing): HBM traffic, L2 hit rate, SQ instruction mix
# 3. scripts/summarize_prof.py writes profiles/<round>_bench_kernel_stats.csv and profiles/<round>_pmc_summary.json
set -e
R=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 24 --warmup 8 --no-cpu-baseline --no-secondary --no-parity > "$OUT/trace.log" 2>&1
echo "trace done"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  # a failed pass ends the script (set -e): a summary built from part of the counters would go stale silently
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$i" -- python3 "$ROOT/bench.py" --steps 4 --warmup 4 --no-cpu-baseline --no-secondary --no-parity --no-exact > "$OUT/pmc_$i.log" 2>&1
  echo "pmc pass $i ($c) done"
done
cd "$ROOT"
python3 scripts/summarize_prof.py "$OUT" profiles "$R"
