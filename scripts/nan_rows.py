"""Diagnostic: where do non-finite factor rows appear after one tiled epoch on a tall matrix (1.25 M users x 1 M items)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import Ctx, mfx, synth
K = int(os.environ.get("K", 128))
shape = dict(nU=int(os.environ.get("NU", 1_250_000)), nI=int(os.environ.get("NI", 1_000_000)), nnz=int(float(os.environ.get("NNZ", 6e6))), K=K)
d = synth.make(shape, seed=1, r0_i=0.002)
tr = d["train"]; nU, nI = d["nUsers"], shape["nI"]
U0, V0 = synth.init_factors(1, nU, nI, K)
with Ctx(0) as ctx:
    ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
    ctx.set_model(nU, nI, K); ctx.compute_invalid()
    for lr in (0.0, 0.0025):
        ctx.set_factors(U0, V0)
        ctx.sgd_epoch(lr, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=0)
        U, V = ctx.get_factors()
        bu = ~np.isfinite(U).all(axis=1); bv = ~np.isfinite(V).all(axis=1)
        chU = (U != U0).any(axis=1); chV = (V != V0).any(axis=1)
        deg = np.diff(tr.rowptr)
        print("lr %g: non-finite user rows %d (first %s), item rows %d (first %s); rows changed: users %d of %d rated, items %d; max|U| %.3g max|V| %.3g"
              % (lr, bu.sum(), np.nonzero(bu)[0][:5], bv.sum(), np.nonzero(bv)[0][:5], chU.sum(), (deg > 0).sum(), chV.sum(),
                 np.nanmax(np.abs(U)), np.nanmax(np.abs(V))), flush=True)
        if bu.any():
            idx = np.nonzero(bu)[0]
            print("   non-finite user ids: min %d max %d; their degrees: %s" % (idx.min(), idx.max(), deg[idx[:10]]))
