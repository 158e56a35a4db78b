#!/bin/bash
# Wide-ALS blocked solve at K = 256 on the C2 matrix: wave-cycle split and instruction-cache counters (run ON the GPU box)
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_alsw; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; export WHAT=als ALS_K=${ALS_K:-256} ALS_ITERS=2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$OUT/sq" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d "$OUT/ic" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/ic.log" 2>&1 || echo "icache counters not available"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/mx" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/mx.log" 2>&1 || echo "mix counters not available"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in sorted(acc.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    if m.get("SQ_WAVE_CYCLES", 0) < 1e8: continue
    print(n, "launches", len(c["SQ_WAVE_CYCLES"]))
    print("   " + "  ".join("%s %.4g" % (k, v) for k, v in sorted(m.items())))
PY
