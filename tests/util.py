"""Shared helpers for the parity tests."""
import numpy as np

from matfac_amd import Ctx, mfx, synth
from oracle import binding as orc


def small(nU=300, nI=200, nnz=6000, K=64, seed=3):
    d = synth.make(dict(nU=nU, nI=nI, nnz=nnz, K=K), seed=seed)
    return d


def load_ctx(ctx, d, K, U0=None, V0=None, with_test=False):
    tr, va = d["train"], d["val"]
    cp, ci, cv = orc.create_col_index(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
    ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval, cp, ci, cv)
    ctx.set_csr(mfx.MAT_VAL, va.nrows, va.ncols, va.rowptr, va.rowind, va.rowval)
    if with_test:
        te = d["test"]
        ctx.set_csr(mfx.MAT_TEST, te.nrows, te.ncols, te.rowptr, te.rowind, te.rowval)
    ctx.set_model(d["nUsers"], d["nItems"], K)
    if U0 is not None:
        ctx.set_factors(U0, V0)
    return ctx.compute_invalid()
