"""Epoch time of the sibling models (IFWMF weights, TMF truncated ranks) on the C2 matrix: tiled kernel variants vs the flat
lock-free kernels, next to the plain tiled epoch."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import Ctx, mfx, synth
K = 64
shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
d = synth.make(shape, seed=1); tr, va = d["train"], d["val"]; nU, nI = d["nUsers"], shape["nI"]
U0, V0 = synth.init_factors(1, nU, nI, K)
uf = np.diff(tr.rowptr).astype(np.float64); itf = np.bincount(tr.rowind, minlength=nI).astype(np.float64)
both = np.concatenate([uf, itf]); mean, std = both.mean(), both.std()
def rank(f): return np.clip(np.ceil(K / (1 + np.exp(-1.0 * ((f - mean) / std - 0.0)))), 1, K).astype(np.int32)
ctx = Ctx(0)
ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
ctx.set_model(nU, nI, K); ctx.compute_invalid()
out = {}
def run(name, mode, n=6, **kw):
    ctx.set_factors(U0, V0)
    ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mode, seed=1, epoch=0, **kw); ctx.synchronize()
    t0 = time.perf_counter()
    for ep in range(1, n + 1): ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mode, seed=1, epoch=ep, **kw)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / n
    out[name] = dict(epoch_ms=round(dt * 1e3, 3), G_updates_per_s=round(tr.nnz / dt / 1e9, 2), val_rmse=round(ctx.rmse(mfx.MAT_VAL), 4))
run("plain tiled f32", mfx.SGD_TILED)
run("plain tiled ref64", mfx.SGD_TILED, arith=mfx.ARITH_REF64)
ctx.sgd_set_ifw(uf.astype(np.float32), (uf / uf.sum()).astype(np.float32), itf.astype(np.float32), (itf / itf.sum()).astype(np.float32), 1000.0)
run("IFWMF tiled", mfx.SGD_TILED); run("IFWMF flat", mfx.SGD_HOGWILD)
ctx.sgd_set_ifw()
ctx.set_tmf(uf.astype(np.float32), rank(uf), itf.astype(np.float32), rank(itf))
run("TMF tiled", mfx.SGD_TILED); run("TMF flat", mfx.SGD_HOGWILD)
print(json.dumps(out))
