"""Run-to-run spread of the lock-free tiled SGD through the host classes (test_host_gpu's hogsgd case)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MFX_NO_SAVE"] = "1"
from test_host_gpu import data, host_train, oracle_train
from oracle import binding as orc
d, K = data(3000, 2000, 300000, seed=2), 16
o = oracle_train(orc.M_SGD, d, K, 40, 1, 0.01, 0.02, 0.02)
print("cpu test %.5f val %.5f" % (o["test"], o["valbest"]))
for m in ("hogsgd", "sgd", "hogsgd", "sgd", "hogsgd"):
    h = host_train(m, d, K, 40, 1, 0.01, 0.02, 0.02)
    print(m, "gpu test %.5f val %.5f" % (h["test"], h["val"]), flush=True)
print("--- sibling models, 40 iterations on the small matrix of test_host_gpu: lock-free default vs MFX_EXACT")
d2 = data()
for m in ("ifwmf:500", "tmf:1.5:-0.2", "tmfd:1.5:-0.2"):
    e = host_train(m, d2, 8, 40, 1, 0.004, 0.02, 0.02, env={"MFX_EXACT": "1"})
    vals = [host_train(m, d2, 8, 40, 1, 0.004, 0.02, 0.02)["val"] for _ in range(4)]
    print(m, "exact val %.5f" % e["val"], "lock-free", " ".join("%.5f" % v for v in vals), flush=True)
