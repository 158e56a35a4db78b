"""Per-epoch trace of the lr-0.01 case of scripts/nan_check.py (3000 x 2000, 300 k ratings, K = 16, factors drawn from
+-0.01): train RMSE, largest |p|, |q| and the number of non-finite entries after each of the first epochs, for
  seq      the oracle's sequential loop over a shuffled list (what the reference does),
  tile1    the tiled schedule with ONE lane group in flight (MFX_SGD_F_ONE_GROUP): the tile order, no concurrency,
  w<N>     the tiled schedule with N waves per workgroup taking part (MFX_SGD_WAVES),
so that order and concurrency can be told apart."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matfac_amd import mfx, synth
from oracle import binding as orc
import ctypes as C

LR = float(os.environ.get("LR", "0.01"))
EPOCHS = int(os.environ.get("EPOCHS", "4"))
K = int(os.environ.get("K", "16"))
d = synth.make(dict(nU=3000, nI=2000, nnz=300000, K=0), seed=2)
tr = d["train"]
nU, nI = d["nUsers"], d["nItems"]
U0 = np.empty((nU, K), np.float32)
V0 = np.empty((nI, K), np.float32)
synth._host().mfh_init_factors(1, nU, nI, K, U0.ctypes.data_as(C.c_void_p), V0.ctypes.data_as(C.c_void_p))
rows = tr.rowids()


def stat(tag, e, U, V):
    bad = int((~np.isfinite(U)).sum() + (~np.isfinite(V)).sum())
    est = np.einsum("ij,ij->i", U[rows].astype(np.float64), V[tr.rowind].astype(np.float64))
    rm = float(np.sqrt(np.mean((tr.rowval - est) ** 2)))
    print("%-6s epoch %d  train rmse %8.4f  max|p| %9.3g  max|q| %9.3g  max|p|^2 %9.3g max|q|^2 %9.3g  non-finite %d" % (
        tag, e, rm, np.nanmax(np.abs(U)), np.nanmax(np.abs(V)), np.nanmax((U * U).sum(1)), np.nanmax((V * V).sum(1)), bad), flush=True)


U, V = U0.copy(), V0.copy()
rng = np.random.default_rng(5)
for e in range(EPOCHS):
    order = rng.permutation(tr.nnz).astype(np.uint64)
    orc.sgd_pass(U, V, rows, tr.rowind, tr.rowval, order, LR, 0.02, 0.02, orc.ARITH_F32, orc.DOT_SEQ)
    stat("seq", e, U, V)

for tag in os.environ.get("RUNS", "tile1,w1,w4,w16").split(","):
    if tag.startswith("w"):
        os.environ["MFX_SGD_WAVES"] = tag[1:]
    ctx = mfx.Ctx(0)
    ctx.set_csr(mfx.MAT_TRAIN, nU, nI, tr.rowptr, tr.rowind, tr.rowval)
    ctx.set_model(nU, nI, K)
    ctx.set_factors(U0, V0)
    for e in range(EPOCHS):
        ctx.sgd_epoch(LR, 0.02, 0.02, mode=mfx.SGD_TILED, order=mfx.ORDER_DEVICE, arith=mfx.ARITH_F32, seed=1, epoch=e,
                      flags=mfx.SGD_F_ONE_GROUP if tag == "tile1" else 0)
        Ue, Ve = ctx.get_factors()
        stat(tag, e, Ue, Ve)
    ctx.close()
