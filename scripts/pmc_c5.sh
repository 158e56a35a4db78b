#!/bin/bash
# HBM-side traffic of the tiled SGD round at one GPU's share of config 5 (scripts/c5_shard.py: 1.25 M x 1 M, 125 M train
# ratings, rank 256) -- separate --pmc passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2), never combined with tracing:
#   bash scripts/pmc_c5.sh r03      ->  profiles/r03_c5_pmc.json
set -e
R=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_c5_$R
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d "$OUT/$n" -- python3 "$ROOT/scripts/c5_shard.py" > "$OUT/$n.log" 2>&1
  echo "$c done"
done
cd "$ROOT"
python3 - "$OUT" "$R" <<'PY'
import csv, glob, json, os, sys
out, rnd = sys.argv[1], sys.argv[2]
vals = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "sgd_slots_kernel" in row["Kernel_Name"] and ", false, false, 0" in row["Kernel_Name"]:
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in vals.items()}
line = [l for l in open(os.path.join(out, "FETCH_SIZE.log")) if l.startswith("{")][-1]
run = json.loads(line)
rec = {"kernel": "sgd_slots_kernel<16,4,F32,rounds>", "launches_sampled": {k: len(v) for k, v in vals.items()}, "counters_mean_per_launch": m,
       "hbm_bytes_per_launch": (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024,
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); separate --pmc passes of scripts/c5_shard.py",
       "updates_per_launch": run["train_nnz"] / 8.0, "train_nnz": run["train_nnz"], "K": run["K"], "epoch_ms_profiled": run["epoch_ms"], "round_ms_profiled": run["round_ms"]}
if "TCC_HIT_sum" in m:
    rec["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
json.dump(rec, open(os.path.join("profiles", "%s_c5_pmc.json" % rnd), "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("hbm_bytes_per_launch", "updates_per_launch", "round_ms_profiled")}))
PY
