import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
os.chdir(ROOT)
import test_host_gpu as T
from oracle import binding as orc
d, K = T.data(3000, 2000, 300000, seed=2), 16
for iters in (40, 120):
    o = T.oracle_train(T.orc.M_SGD, d, K, iters, 1, 0.01, 0.02, 0.02)
    for m in ("sgd", "hogsgd", "sgdpar", "sgdu"):
        gaps = []
        for rep in range(3):
            h = T.host_train(m, d, K, iters, 1, 0.01, 0.02, 0.02)
            gaps.append(round(h["test"] - o["test"], 4))
        print(iters, m, "cpu", round(o["test"], 4), "gaps", gaps, flush=True)
