"""Round-1 finding: the tiled path produced a first-epoch NaN at lr 0.01 on a 3000 x 2000 matrix (the reference does not).
Forced lock-free schedule (MFX_EXACT=0), 40 iterations, against the oracle's sequential loop; MFX_SGD_WAVES overrides the
number of waves per workgroup that take part."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_host_gpu import host_train, oracle_train, data
from oracle import binding as orc
d, K = data(3000, 2000, 300000, seed=2), 16
o = oracle_train(orc.M_SGD, d, K, 40, 1, 0.01, 0.02, 0.02)
print("oracle: test %.5f val %.5f lr %.5g" % (o["test"], o["valbest"], o["learnRate"]), flush=True)
for w in (os.environ.get("WAVES", "auto,16")).split(","):
    env = {"MFX_EXACT": "0"}
    if w != "auto":
        env["MFX_SGD_WAVES"] = w
    for m in ("sgd", "hogsgd"):
        h = host_train(m, d, K, 40, 1, 0.01, 0.02, 0.02, env=env)
        print("waves %s %s: test %.5f val %.5f final lr %.5g (0.01 = never halved)" % (w, m, h["test"], h["val"], h["lr"]), flush=True)
