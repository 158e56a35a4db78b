#!/bin/bash
# What the CCD++ kernels wait for at C4 (run ON the GPU box from the repo root): SQ wave-cycle split, instruction counts,
# memory-side bytes and L2 hit rate per launch, separate --pmc passes of the same workload.
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_ccd_sq; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; export WHAT=ccd CCD_NK=4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d "$OUT/sq" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/sq2" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/w.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/t" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/t.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0][:60]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in sorted(acc.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    if m.get("SQ_WAVE_CYCLES", 0) < 1e6: continue
    print(n, "launches", len(c["SQ_WAVE_CYCLES"]))
    print("   " + "  ".join("%s %.4g" % (k, v) for k, v in sorted(m.items())))
PY
