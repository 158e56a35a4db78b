import json, os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from matfac_amd import Ctx, mfx, synth
K = 64
shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
d = synth.make(shape, seed=1); tr, va = d["train"], d["val"]; nU, nI = d["nUsers"], shape["nI"]
U0, V0 = synth.init_factors(1, nU, nI, K)
ctx = Ctx(0)
ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0); ctx.compute_invalid()
out = {}
for name, mode, order in (("users", mfx.SGD_USERS, mfx.ORDER_NATURAL), ("hogwild_flat", mfx.SGD_HOGWILD, mfx.ORDER_DEVICE), ("tiled", mfx.SGD_TILED, mfx.ORDER_DEVICE)):
    ctx.set_factors(U0, V0)
    ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mode, order=order, seed=1, epoch=0, arith=mfx.ARITH_REF64 if name == "users" else mfx.ARITH_F32); ctx.synchronize()
    t0 = time.perf_counter()
    for ep in range(1, 6): ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mode, order=order, seed=1, epoch=ep, arith=mfx.ARITH_REF64 if name == "users" else mfx.ARITH_F32)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 5
    out[name] = dict(epoch_ms=round(dt * 1e3, 2), G=round(tr.nnz / dt / 1e9, 2), val=round(ctx.rmse(mfx.MAT_VAL), 4))
print(json.dumps(out))
