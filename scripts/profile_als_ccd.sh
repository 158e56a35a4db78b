#!/bin/bash
# Per-round profile of the non-headline trainers (run ON the GPU box from the repo root): ALS (C3), CCD++ (C4), CCD and the
# SVD initialisation under rocprofv3 --kernel-trace --stats; the json lines of scripts/bench_als_ccd.py go next to it.
set -e
R=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_als_ccd_$R
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
export WHAT=als,ccd,cd,svd
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/summary.jsonl" 2> "$OUT/trace.log"
ALS_K=128 ALS_ITERS=3 WHAT=als python3 "$ROOT/scripts/bench_als_ccd.py" >> "$OUT/summary.jsonl" 2>> "$OUT/trace.log"
cat "$OUT/summary.jsonl"
