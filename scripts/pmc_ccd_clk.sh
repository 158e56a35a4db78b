#!/bin/bash
# Effective clock and wave-cycle split of the CCD++ pass kernels at C4 (run ON the GPU box from the repo root)
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_ccd_clk; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; export WHAT=ccd CCD_NK=4
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tr" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/tr.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$OUT/sq" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/sq.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d "$OUT/t" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/t.log" 2>&1 || true
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = {}
for f in glob.glob(out + "/tr/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"].split("(")[0][:40]] = float(r["AverageNs"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in sorted(acc.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    if m.get("SQ_WAVE_CYCLES", 0) < 5e7: continue
    d = dur.get(n, 0)
    print("%s: %.1f us" % (n, d / 1e3))
    if d: print("   effective clock %.2f GHz" % (m.get("GRBM_GUI_ACTIVE", 0) / 8 / d))
    print("   " + "  ".join("%s %.4g" % (k, v) for k, v in sorted(m.items())))
PY
