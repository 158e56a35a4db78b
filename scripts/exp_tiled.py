"""Diagnostic: time + convergence of MFX_SGD_TILED on the C2 shape."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import Ctx, mfx, synth
shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] * float(os.environ.get("SCALE", 1)) / 0.8)
K = int(os.environ.get("K", 64))
d = synth.make(shape, seed=1); tr, va = d["train"], d["val"]; nU, nI = d["nUsers"], shape["nI"]
U0, V0 = synth.init_factors(1, nU, nI, K)
for blocks in [int(b) for b in os.environ.get("BLOCKS", "2048,1024,512").split(",")]:
    for lr in [float(x) for x in os.environ.get("LRS", "0.0025").split(",")]:
        ctx = Ctx(0)
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
        ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0); ctx.compute_invalid()
        ctx.sgd_epoch(0.0, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=0, blocks=blocks)  # builds the slot lists
        ctx.prof_enable(True); ctx.prof_reset()
        traj = []
        t0 = time.perf_counter()
        for ep in range(int(os.environ.get("EPOCHS", 8))):
            ctx.sgd_epoch(lr, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=ep, blocks=blocks, own=int(os.environ.get('OWN', 0)))
            traj.append(round(ctx.rmse(mfx.MAT_TRAIN), 4))
        wall = time.perf_counter() - t0
        ms, n = ctx.prof_get(mfx.K_SGD); sw, ns = ctx.prof_get(mfx.K_SGD_SWEEP)
        print(json.dumps(dict(own=os.environ.get('OWN','0'), blocks=blocks, lr=lr, round_ms=ms / max(n, 1), sweep_ms=sw / max(ns, 1),
                              kernel_gups=tr.nnz * (n / 8) / (ms + sw) / 1e6, traj=traj)), flush=True)
        ctx.close()
