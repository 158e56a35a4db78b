"""Condense the rocprofv3 outputs of scripts/profile_round.sh into profiles/<round>_*.  HBM bytes per launch =
2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes): gfx950 reports half of the wide coalesced reads (MI355X_MICROARCH.md)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

out, dst, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, "%s_bench_kernel_stats.csv" % rnd))


def short(name):
    for k in ("sgd_slots_kernel", "eval_sse_kernel", "eval_norm_kernel", "sgd_hogwild_kernel", "sgd_flow_wide_kernel", "sgd_flow_tag_kernel"):
        if k in name:
            if k == "sgd_slots_kernel":      # <L, C, ARITH, SWEEP, OWN_U>: the leftover sweep launch is reported apart
                args = [a.strip() for a in name.split("<", 1)[1].split(">", 1)[0].split(",")]
                return k + ("_sweep" if len(args) > 3 and args[3] == "true" else "")
            return k
    return None


vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k:
            vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {}
for k, cs in vals.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    e = {"launches_sampled": max(len(v) for v in cs.values()), "counters_mean_per_launch": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
        e["note"] = "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); separate --pmc passes"
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
        e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    summary[k] = e
# updates one launch of the round kernel processed in the profiled command (its JSON line is the last line of trace.log)
try:
    line = [l[l.index('{"metric"'):] for l in open(os.path.join(out, "trace.log")) if '{"metric"' in l][-1]
    upl = json.loads(line)["roofline"]["updates_per_launch"]
    for k in ("sgd_slots_kernel", "sgd_hogwild_kernel"):
        if k in summary:
            summary[k]["updates_per_launch"] = upl
except Exception as e:
    print("no updates_per_launch:", e)
json.dump(summary, open(os.path.join(dst, "%s_pmc_summary.json" % rnd), "w"), indent=1)
print(json.dumps({k: {x: v.get(x) for x in ("hbm_bytes_per_launch", "l2_hit_rate")} for k, v in summary.items()}))
