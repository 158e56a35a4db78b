"""One-time costs before the first epoch: upload (+ column view) and the slot lists, device vs host builders.
SHAPE=C2|C5s"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import Ctx, mfx, synth
which = os.environ.get("SHAPE", "C2")
if which == "C2":
    d = synth.make("C2", seed=1); K = 64
else:
    K = 256
    d = synth.make(dict(nU=1_250_000, nI=1_000_000, nnz=int(125_000_000 / 0.8), K=K), seed=1, r0_i=0.002)
tr = d["train"]; nU, nI = d["nUsers"], max(d["nItems"], tr.ncols)
out = dict(shape=which, train_nnz=int(tr.nnz), K=K)
for host in (0, 1):
    if host: os.environ["MFX_SLOTS_HOST"] = "1"
    else: os.environ.pop("MFX_SLOTS_HOST", None)
    ctx = Ctx(0)
    ctx.synchronize()
    t0 = time.time(); ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval); t_up = time.time() - t0
    ctx.set_model(nU, nI, K)
    U0, V0 = synth.init_factors(1, nU, nI, K)
    t0 = time.time(); ctx.set_factors(U0, V0); t_fac = time.time() - t0
    t0 = time.time(); ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=0); ctx.synchronize(); t_first = time.time() - t0
    t0 = time.time(); ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=1); ctx.synchronize(); t_second = time.time() - t0
    out["host_slots" if host else "device_slots"] = dict(set_csr_train_s=round(t_up, 4), set_factors_s=round(t_fac, 4),
                                                         first_epoch_s=round(t_first, 4), second_epoch_s=round(t_second, 5))
    ctx.close() if hasattr(ctx, "close") else None
    del ctx
print(json.dumps(out))
