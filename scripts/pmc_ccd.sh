#!/bin/bash
# HBM traffic of the CCD++ kernels at C4 (run ON the GPU box from the repo root): FETCH_SIZE / WRITE_SIZE per launch
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_ccd; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; export WHAT=ccd CCD_NK=4
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/w.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0][:48]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    f = sum(c.get("FETCH_SIZE", [0])) / max(1, len(c.get("FETCH_SIZE", [1]))); w = sum(c.get("WRITE_SIZE", [0])) / max(1, len(c.get("WRITE_SIZE", [1])))
    if f + w > 50000: print("%-50s launches %4d  fetch x2 %8.1f MB  write %8.1f MB" % (n, len(c.get("FETCH_SIZE", [])), 2 * f / 1024, w / 1024))
PY
