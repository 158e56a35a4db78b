"""Cost of the every-iteration objective pass (model.cpp:1770-1815) next to one SGD epoch.  SHAPE=C2|C5s"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import Ctx, mfx, synth
which = os.environ.get("SHAPE", "C2")
if which == "C2":
    shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8); K = 64; d = synth.make(shape, seed=1)
else:
    K = 256; shape = dict(nU=1_250_000, nI=1_000_000, nnz=int(125_000_000 / 0.8), K=K); d = synth.make(shape, seed=1, r0_i=0.002)
tr = d["train"]; nU, nI = d["nUsers"], shape["nI"]
U0, V0 = synth.init_factors(1, nU, nI, K)
ctx = Ctx(0)
ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0); ctx.compute_invalid()
def timeit(f, n=5):
    f(); ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    ctx.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ep = [0]
def epoch():
    ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=ep[0]); ep[0] += 1
flat_before = timeit(lambda: ctx.eval(mfx.MAT_TRAIN, with_norms=True))
t_epoch = timeit(epoch)
after = timeit(lambda: ctx.eval(mfx.MAT_TRAIN, with_norms=True))
e = ctx.eval(mfx.MAT_TRAIN, with_norms=True)
print(json.dumps(dict(shape=which, nnz=int(tr.nnz), K=K, objective_ms_before_slots=round(flat_before, 3), sgd_epoch_ms=round(t_epoch, 3),
                      objective_ms_after_slots=round(after, 3), sse=e.sse, n=e.n)))
