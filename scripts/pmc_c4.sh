#!/bin/bash
# Memory-side traffic of one CCD++ rank-one step at C4 (run ON the GPU box from the repo root): FETCH_SIZE and WRITE_SIZE per launch
# of every kernel of the step, separate --pmc passes, never combined with tracing:
#   bash scripts/pmc_c4.sh r02   ->   gpurun_out/<round>_c4_pmc.json  (copy to profiles/)
set -e
R=${1:-r02}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_c4_$R; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; export WHAT=ccd CCD_NK=6
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -- python3 "$ROOT/scripts/bench_als_ccd.py" > "$OUT/w.log" 2>&1
cd "$ROOT"
python3 - "$OUT" "$R" <<'PY'
import csv, glob, json, sys, collections
out, rnd = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
# launches per rank-one step in steady state (add_back: the deferred subtract of the previous factor fused with the add-back)
# (round 4: the first sweep of a factor carries the residual update -- ccd_pass_fused_kernel / colpass_fused_kernel and the two pair builders)
fused = any(n.startswith("ccd_pass_fused_kernel") for n in acc)
per_factor = {"ccd_pass_kernel": 4 if fused else 5, "ccd_pass_fused_kernel": 1, "ccd_finish_kernel": 5, "colpass_kernel": 4 if fused else 5,
              "colpass_fused_kernel": 1, "colfinish_kernel": 5, "ccd_pairs_kernel": 1,
              "resid_fused_kernel": 1, "colresid_kernel<2": 1, "colresid_light_kernel<2": 1, "extract_col_kernel": 1, "store_col_kernel": 1}
kern, total = {}, 0.0
for n, c in acc.items():
    key = next((k for k in per_factor if n.startswith(k)), None)
    if key is None: continue
    f = sum(c.get("FETCH_SIZE", [0])) / max(1, len(c.get("FETCH_SIZE", [1])))
    w = sum(c.get("WRITE_SIZE", [0])) / max(1, len(c.get("WRITE_SIZE", [1])))
    b = (2 * f + w) * 1024
    kern[n] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": b, "launches_per_factor": per_factor[key]}
    total += b * per_factor[key]
rec = {"workload": "C4 CCD++ rank-one step (scripts/bench_als_ccd.py, WHAT=ccd)", "kernels": kern, "hbm_bytes_per_factor": total,
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads; an upper bound where the loads are 8 bytes per lane); separate --pmc passes"}
json.dump(rec, open("gpurun_out/%s_c4_pmc.json" % rnd, "w"), indent=1)
print(json.dumps({"hbm_bytes_per_factor": total, "kernels": {k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in kern.items()}}))
PY
