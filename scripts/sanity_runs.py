import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matfac_amd import synth
from tests.test_host_gpu import host_train
shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
d = synth.make(shape, seed=1)
out = {}
for lr in (0.005, 0.01):
    h = host_train("hogsgd", d, 64, 40, 1, lr, 0.01, 0.01)
    out["C2_hogsgd_lr%g" % lr] = dict(val=h["val"], test=h["test"], final_lr=h["lr"], iters=h["iters"])
d1 = synth.make("C1", seed=1)
for m in ("sgd", "hogsgd", "sgdu", "sgdpar", "als", "ccd++"):
    h = host_train(m, d1, 10, 120, 1, 0.005, 0.01 if "sgd" in m else 1.0, 0.01 if "sgd" in m else 1.0)
    out["C1_" + m] = dict(val=h["val"], test=h["test"], final_lr=h["lr"], iters=h["iters"])
print(json.dumps(out))
