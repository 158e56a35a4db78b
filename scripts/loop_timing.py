"""End-to-end time per iteration of the host training loops (ModelMF::hogTrain / trainALS / trainCCDPP incl.
Model::isTerminateModel every iteration) on the C2 shape, through matfac_amd/host (mfh_train)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import synth
from tests.test_host_gpu import host_train
shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
d = synth.make(shape, seed=1)
out = {}
for method, K, iters, lr, reg in (("hogsgd", 64, 30, 0.0025, 0.01), ("hogsgd", 64, 330, 0.0025, 0.01), ("als", 64, 6, 0.005, 5.0), ("als", 64, 26, 0.005, 5.0)):
    t0 = time.time(); h2 = host_train(method, d, K, iters, 1, lr, reg, reg); t2 = time.time() - t0
    out[method + str(iters)] = dict(K=K, iters=h2["iters"], ms_per_iter=h2["loop_s"] / h2["iters"] * 1e3, total_s=t2, val_rmse=h2["val"], test_rmse=h2["test"])
a, b = out["hogsgd30"], out["hogsgd330"]
out["hogsgd_steady_ms_per_iter"] = (b["ms_per_iter"] * b["iters"] - a["ms_per_iter"] * a["iters"]) / (b["iters"] - a["iters"])
a, b = out["als6"], out["als26"]
out["als_steady_ms_per_iter"] = (b["ms_per_iter"] * b["iters"] - a["ms_per_iter"] * a["iters"]) / max(1, b["iters"] - a["iters"])
print(json.dumps(out))
