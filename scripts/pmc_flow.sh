#!/bin/bash
# Counters of the exact replay's kernel (sgd_flow_wide_kernel, C2, rank 64, double bracket) -- separate --pmc passes, never combined with
# tracing -- plus one --kernel-trace --stats pass for its duration:   bash scripts/pmc_flow.sh r04   ->  profiles/r04_flow_pmc.json
set -e
R=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_flow_$R
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
export CONFIGS="tag:4"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/scripts/flow_tune.py" > "$OUT/trace.log" 2>&1
echo "trace done"
i=0
for c in "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_SMEM SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$i" -- python3 "$ROOT/scripts/flow_tune.py" > "$OUT/pmc_$i.log" 2>&1
  echo "pmc pass $i ($c) done"
done
cd "$ROOT"
python3 - "$OUT" "$R" <<'PY'
import csv, glob, json, os, sys
out, rnd = sys.argv[1], sys.argv[2]
vals = {}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "sgd_flow_wide_kernel" in row["Kernel_Name"]:
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in vals.items()}
dur = None
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "sgd_flow_wide_kernel" in row["Name"]:
            dur = {"calls": int(row["Calls"]), "average_ns": float(row["AverageNs"]), "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
line = [l for l in open(os.path.join(out, "trace.log")) if l.startswith("C2 K=")][-1].strip()
n = 20029657
rec = {"kernel": "sgd_flow_wide_kernel<1, REF64> (C2: 20 029 657 ratings, rank 64, 4096 queues)", "launches_sampled": {k: len(v) for k, v in vals.items()},
       "counters_mean_per_launch": m, "kernel_trace": dur, "run_line": line,
       "per_update": {k: m[k] / n for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_LDS") if k in m},
       "note": "separate --pmc passes of scripts/flow_tune.py (CONFIGS=tag:4); wave-instruction counts divided by the ratings of the epoch"}
if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m:
    rec["wait_any_share_of_wave_cycles"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:
    rec["wait_inst_any_share_of_wave_cycles"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
if "SQ_ACTIVE_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:
    rec["active_inst_any_share_of_wave_cycles"] = m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    rec["memory_side_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
os.makedirs(os.path.join(out, "summary"), exist_ok=True)
json.dump(rec, open(os.path.join(out, "summary", "%s_flow_pmc.json" % rnd), "w"), indent=1)
print(json.dumps({k: rec.get(k) for k in ("kernel_trace", "per_update", "wait_any_share_of_wave_cycles", "active_inst_any_share_of_wave_cycles")}))
PY
