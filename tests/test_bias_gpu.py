"""ModelMFBias (modelMFBias.cpp), the bias-only sibling of ModelMF: the sequential loop replayed by one lane and by the
dataflow schedule, the objective / RMSE sums, and the host class's whole train() against a simulation that drives the
oracle's pass with the same std::shuffle sequence and best-validation bookkeeping."""
import ctypes as C

import numpy as np
import pytest

from matfac_amd import Ctx, mfx, synth
from oracle import binding as orc
from tests.util import load_ctx, small

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [mfx.SGD_SERIAL, mfx.SGD_LEVELS])
def test_bias_epochs_bit_exact_and_eval(mode):
    d = small(nU=900, nI=80, nnz=60_000, K=4, seed=13)          # ~600 ratings per item: long chains
    tr, va = d["train"], d["val"]
    nU, nI = d["nUsers"], d["nItems"]
    rng = np.random.default_rng(5)
    ub0 = rng.normal(0, 0.5, nU).astype(np.float32)
    ib0 = rng.normal(0, 0.5, nI).astype(np.float32)
    U0, V0 = orc.init_factors(1, nU, nI, 4)
    ru = tr.rowids()
    order = np.arange(tr.nnz, dtype=np.uint64)
    mt = orc.MT(3)
    ubo, ibo = ub0.copy(), ib0.copy()
    with Ctx(0) as ctx:
        invU, invI = load_ctx(ctx, d, 4, U0, V0)
        ctx.bias_set(ub0, ib0)
        for ep in range(3):
            mt.shuffle_u64(order)
            ctx.sgd_set_order(order)
            ctx.bias_epoch(0.01, 0.05, 0.02, mode=mode)
            orc.bias_pass(ubo, ibo, ru, tr.rowind, tr.rowval, order, 0.01, 0.05, 0.02)
        ub, ib = ctx.bias_get()
        e_tr = ctx.bias_eval(mfx.MAT_TRAIN)
        e_va = ctx.bias_eval(mfx.MAT_VAL)
        ctx.snapshot_best()
        ctx.bias_set(ub0, ib0)
        ubb, ibb = ctx.bias_get(mfx.SNAP_BEST)
        ctx.restore_best()
        ubr, ibr = ctx.bias_get()
    assert np.array_equal(ub, ubo) and np.array_equal(ib, ibo)
    assert np.abs(ub - ub0).max() > 0.05
    assert np.array_equal(ubb, ubo) and np.array_equal(ubr, ubo) and np.array_equal(ibr, ibo)
    _, sse, cnt, un, inn = orc.bias_eval(ubo, ibo, nU, nI, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, invI)
    assert e_tr.n == cnt and abs(e_tr.sse - sse) <= 1e-12 * sse
    assert abs(e_tr.unorm2 - un) <= 1e-12 * un and abs(e_tr.inorm2 - inn) <= 1e-12 * inn
    _, vsse, vcnt, _, _ = orc.bias_eval(ubo, ibo, nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, invU, invI)
    assert e_va.n == vcnt and abs(e_va.sse - vsse) <= 1e-12 * vsse


def test_model_mf_bias_train_loop_follows_the_reference_loop():
    d = synth.make(dict(nU=500, nI=300, nnz=25000, K=0), seed=12)
    tr, va, te = d["train"], d["val"], d["test"]
    nU, nI, K = d["nUsers"], d["nItems"], 6
    iters, lr, ureg, ireg, seed = 40, 0.01, 0.02, 0.03, 2
    lib = synth._host()
    ubl, ibl, ubb, ibb = (np.empty(n, np.float32) for n in (nU, nI, nU, nI))
    stats = np.zeros(8)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.mfh_train_bias(C.c_int32(tr.nrows), P(tr.rowptr), P(tr.rowind), P(tr.rowval), C.c_int32(tr.ncols), P(va.rowptr), P(va.rowind),
                            P(va.rowval), C.c_int32(va.ncols), P(te.rowptr), P(te.rowind), P(te.rowval), C.c_int32(te.ncols), C.c_int32(K),
                            C.c_int32(iters), C.c_int32(seed), C.c_float(lr), C.c_float(ureg), C.c_float(ireg), P(ubl), P(ibl), P(ubb),
                            P(ibb), P(stats))
    assert rc == 0
    # the reference's constructor stream: uFac, iFac, then uBias, iBias (model.cpp:2331-2362)
    invU, invI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
    ub, ib = orc.init_bias(seed, nU, nI, K)
    ru = tr.rowids()
    mt = orc.MT(seed)
    order = np.arange(tr.nnz, dtype=np.uint64)
    _, s0, c0, _, _ = orc.bias_eval(ub, ib, nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, invU, invI)
    bestval = np.sqrt(s0 / c0)
    bub, bib = ub.copy(), ib.copy()
    for it in range(iters):
        mt.shuffle_u64(order)
        orc.bias_pass(ub, ib, ru, tr.rowind, tr.rowval, order, lr, ureg, ireg)
        _, s, c, _, _ = orc.bias_eval(ub, ib, nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, invU, invI)
        v = np.sqrt(s / c)
        if v < bestval:
            bestval, bub, bib = v, ub.copy(), ib.copy()
    assert np.array_equal(ubl, ub) and np.array_equal(ibl, ib)
    assert np.array_equal(ubb, bub) and np.array_equal(ibb, bib)
    assert abs(stats[2] - bestval) < 1e-12
    assert int(stats[4]) == iters
