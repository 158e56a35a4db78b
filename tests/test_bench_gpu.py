"""bench.py's driver contract, exercised end to end at a reduced size: the N = 1 line with its roofline record, and the
N = 2 path (the script starts its own ranks; two processes share the one GPU and exchange through gloo, BENCH_COMM=gloo --
the driver's own N > 1 runs use RCCL on one GPU per rank) in both scaling modes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # ONE JSON line on stdout, everything else on stderr
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    d = _bench(["--steps", "24", "--warmup", "2", "--scale", "0.25", "--no-cpu-baseline", "--no-parity"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 24 and d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
    r = d["roofline"]
    assert r["launches_per_step"] == 8 and r["launches"] == 8 * 3          # events around every 8th epoch's 8 round launches
    assert 0 < r["frac"] <= 1.0 and r["bound"] in ("l2-memory-side", "l2-rows", "valu-issue")
    assert 0 < r["l2_rows"]["frac"] <= 1.0
    assert r["algorithmic"]["frac"] > 0 and r["avg_launch_ms"] > 0
    assert d["value"] > 1e9


@pytest.mark.parametrize("exchange", ["rotate", "allreduce"])
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_on_one_gpu_through_gloo(scaling, exchange):
    d = _bench(["--gpus", "2", "--steps", "6", "--warmup", "1", "--scale", "0.1", "--scaling", scaling, "--exchange", exchange],
               env={"BENCH_COMM": "gloo"})
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["config"]["exchange"] == exchange
    print("N = 2 (%s, %s): val RMSE after %.4f, same epochs without exchange %.4f" % (scaling, exchange, d["val_rmse_after"],
                                                                                   d["val_rmse_after_same_epochs_without_exchange"]))
    if exchange == "rotate":      # every update applied once with its full step: not behind the run without any exchange
        assert d["val_rmse_after"] < d["val_rmse_after_same_epochs_without_exchange"] + 0.02
    assert ("strong" in d["config"]["workload"]) == (scaling == "strong")
    assert d["value"] > 1e8 and d["val_rmse_after"] > 0 and d["val_rmse_after_same_epochs_without_exchange"] > 0
    assert "cpu_baseline" not in d and "secondary" not in d
    # round 4: the line carries the OTHER exchange and (weak-scaling runs) the strong-scaling split of the one named matrix
    other = "allreduce" if exchange == "rotate" else "rotate"
    sr = d["sub_records"]
    assert sr["exchange_" + other]["ms_per_step"] > 0 and sr["exchange_" + other]["ms_per_exchange_alone"] > 0
    assert ("strong_scaling" in sr) == (scaling == "weak")
    if scaling == "weak":
        assert sr["strong_scaling"]["value"] > 1e8 and "ONE" in sr["strong_scaling"]["workload"]
