"""Converged-model parity of the fast (lock-free, tiled) SGD path: run ModelMF::train on the GPU and the
oracle's sequential restatement of the same loop for the same number of iterations and compare the
best-validation models' test RMSE.  Diagnostic; prints one JSON line per configuration."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import synth
from oracle import binding as orc
from tests.test_host_gpu import host_train, oracle_train

shape = dict(nU=30000, nI=8000, nnz=3_000_000, K=0)
d = synth.make(shape, seed=3)
K, iters, lr, reg = 32, int(os.environ.get("ITERS", 150)), float(os.environ.get("LR", 0.005)), 0.02
t0 = time.time(); o = oracle_train(orc.M_SGD, d, K, iters, 1, lr, reg, reg); tc = time.time() - t0
res = dict(train_nnz=d["train"].nnz, K=K, iters=iters, lr=lr, cpu_s=tc, cpu_test=o["test"], cpu_val=o["valbest"],
           cpu_best_iter=o["bestIter"], cpu_final_lr=o["learnRate"])
for method in ("sgd", "hogsgd"):
    t0 = time.time(); h = host_train(method, d, K, iters, 1, lr, reg, reg); tg = time.time() - t0
    res[method] = dict(gpu_s=tg, test=h["test"], val=h["val"], dtest=h["test"] - o["test"], final_lr=h["lr"])
print(json.dumps(res))
