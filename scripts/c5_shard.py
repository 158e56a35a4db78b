"""One GPU's share of config C5 (10M x 1M, 1B ratings, rank 256 over 8 GPUs): 1.25 M users x 1 M items, 125 M
ratings, K=256 -- a working set (U 1.28 GB, V 1.02 GB) far beyond the caches.  SGD epoch time + sanity."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import Ctx, mfx, synth
K = int(os.environ.get("K", 256))
shape = dict(nU=1_250_000, nI=1_000_000, nnz=int(125_000_000 * float(os.environ.get("SCALE", 1.0)) / 0.8), K=K)
t0 = time.time(); d = synth.make(shape, seed=1, r0_i=0.002); gen = time.time() - t0
tr, va = d["train"], d["val"]; nU, nI = d["nUsers"], shape["nI"]
U0, V0 = synth.init_factors(1, nU, nI, K)
ctx = Ctx(0)
t0 = time.time()
ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0); ctx.compute_invalid()
up = time.time() - t0
t0 = time.time(); ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=0); ctx.synchronize(); first = time.time() - t0
for ep in (1, 2, 3):       # (the slot lists of the other three tilings are built by their first epochs)
    ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=ep)
ctx.synchronize()
ctx.prof_enable(True); ctx.prof_reset()
traj = [round(ctx.rmse(mfx.MAT_VAL), 4)]
t0 = time.perf_counter(); n = 5
for ep in range(4, 4 + n):
    ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=ep)
ctx.synchronize(); wall = (time.perf_counter() - t0) / n
traj.append(round(ctx.rmse(mfx.MAT_VAL), 4))
ms, cnt = ctx.prof_get(mfx.K_SGD); ems, ecnt = ctx.prof_get(mfx.K_EVAL)
print(json.dumps(dict(train_nnz=tr.nnz, K=K, datagen_s=gen, upload_s=up, first_epoch_incl_slot_build_s=first,
                      epoch_ms=wall * 1e3, updates_per_s=tr.nnz / wall, algorithmic_GBs=(16 * K + 12) * tr.nnz / wall / 1e9,
                      hbm_roofline_updates_per_s=8e12 / (16 * K + 12), round_ms=ms / max(cnt, 1), val_rmse=traj)))
