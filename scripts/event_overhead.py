import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from matfac_amd import Ctx, mfx, synth
shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"]/0.8)
d = synth.make(shape, seed=1); tr = d["train"]; nU, nI = d["nUsers"], shape["nI"]
ctx = Ctx(0)
ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
ctx.set_model(nU, nI, 64)
U0, V0 = synth.init_factors(1, nU, nI, 64); ctx.set_factors(U0, V0); ctx.compute_invalid()
for prof in (False, True, False, True):
    ctx.prof_enable(prof); ctx.prof_reset()
    for ep in range(5): ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=ep)
    ctx.synchronize()
    t0 = time.perf_counter()
    for ep in range(5, 205): ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=ep)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print("prof", prof, "ms/epoch %.4f" % (dt * 1e3), "G upd/s %.2f" % (tr.nnz / dt / 1e9), flush=True)
