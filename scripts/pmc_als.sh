#!/bin/bash
# SQ counters of the ALS kernel on the C2 matrix (run ON the GPU box from the repo root):
#   bash scripts/pmc_als.sh          the accumulation alone (MFX_ALS_NOSOLVE=1: rounds 1 and 3)
#   bash scripts/pmc_als.sh solve    the kernel as it runs, accumulation + in-register solve (round 4: profiles/r04_als_pmc.txt)
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_als_${1:-nosolve}; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
if [ "${1:-nosolve}" != "solve" ]; then export MFX_ALS_NOSOLVE=1; fi
rocprofv3 -L > "$OUT/list.txt" 2>&1 || true
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/$tag" -- python3 "$ROOT/scripts/als_sides.py" > "$OUT/$tag.log" 2>&1 || echo "group failed: $grp"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0][:40]
        if "als_segment" in n: acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    for k, v in sorted(c.items()):
        print("%-40s %-28s n=%3d avg=%.4g min=%.4g max=%.4g" % (n, k, len(v), sum(v) / len(v), min(v), max(v)))
PY
