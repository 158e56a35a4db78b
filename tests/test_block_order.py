"""Round-1 finding, traced in round 2: at learnrate 0.01 on a 3000 x 2000 matrix (300 k ratings, K = 16, factors drawn from
+-0.01) the tiled schedule leaves its first epoch with non-finite factors while the reference's sequential loop
(ModelMF::train, modelMF.cpp:83-105) trains.  The cause is the ORDER, not the ratings in flight:

  * the reference's OWN block-ordered trainer, trainSGDPar (modelMF.cpp:229-304, sgdUpdateBlockSeq util.cpp:1077-1107;
    oracle orc_strat_epoch), diverges the same way on the same data at the same rate with 2, 4 or 8 parts and ONE thread;
  * the tiled schedule run with ONE lane group in flight (MFX_SGD_F_ONE_GROUP: no concurrency at all) diverges too,
    and so does the oracle's sequential loop over that list.

Mechanism (scripts/nan_trace.py): when a block of users that has not been visited yet meets item rows
that earlier blocks have already grown (|q|^2 ~ 24), every such user's first visit sets p = 2*lr*e*q and pushes q by
(2*lr*e)^2 * q along itself; with no established user in between to pull q back, a few hundred of them in a row carry
|q|^2 past 1/lr, where the user step 2*lr*|q|^2 > 2 is unstable.  A shuffled list interleaves fresh and established users.
Model::isTerminateModel's guard (model.cpp:1486-1510) halves the rate in both programs."""
import ctypes as C

import numpy as np
import pytest

from matfac_amd import synth
from oracle import binding as orc

LR, K = 0.01, 16


@pytest.fixture(scope="module")
def prob():
    d = synth.make(dict(nU=3000, nI=2000, nnz=300000, K=0), seed=2)
    U0, V0 = synth.init_factors(1, d["nUsers"], d["nItems"], K)
    return d, U0, V0


def _finite(U, V):
    return bool(np.isfinite(U).all() and np.isfinite(V).all())


def test_reference_stratified_order_diverges_where_its_sequential_loop_trains(prob):
    d, U0, V0 = prob
    tr = d["train"]
    nU, nI = d["nUsers"], d["nItems"]
    U, V = U0.copy(), V0.copy()
    order = np.arange(tr.nnz, dtype=np.uint64)
    orc.MT(1).shuffle_u64(order)
    orc.sgd_pass(U, V, tr.rowids(), tr.rowind, tr.rowval, order, LR, 0.02, 0.02, orc.ARITH_REF64, orc.DOT_SEQ)
    assert _finite(U, V) and (V * V).sum(1).max() < 1.0 / LR
    inv_u, inv_i = np.zeros(nU, np.uint8), np.zeros(nI, np.uint8)
    for T in (2, 4, 8):
        U, V = U0.copy(), V0.copy()
        orc.time_strat(U, V, tr.rowptr, tr.rowind, tr.rowval, nU, nI, inv_u, inv_i, T, LR, 0.02, 0.02, seed=1, epochs=1)
        assert not _finite(U, V), "trainSGDPar order, %d parts" % T
    # half the rate: the block order trains as well
    U, V = U0.copy(), V0.copy()
    orc.time_strat(U, V, tr.rowptr, tr.rowind, tr.rowval, nU, nI, inv_u, inv_i, 8, LR / 2, 0.02, 0.02, seed=1, epochs=1)
    assert _finite(U, V)


@pytest.mark.gpu
def test_tiled_order_without_any_concurrency_diverges_like_the_references_block_order(prob):
    from matfac_amd import mfx
    d, U0, V0 = prob
    tr = d["train"]
    nU, nI = d["nUsers"], d["nItems"]
    res = {}
    with mfx.Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, nU, nI, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_model(nU, nI, K)
        ctx.compute_invalid()
        for lr in (LR, LR / 2):
            ctx.set_factors(U0, V0)
            ctx.sgd_epoch(lr, 0.02, 0.02, mode=mfx.SGD_TILED, order=mfx.ORDER_DEVICE, arith=mfx.ARITH_F32, seed=1, epoch=0,
                          flags=mfx.SGD_F_ONE_GROUP)
            U, V = ctx.get_factors()
            res[lr] = _finite(U, V)
            if lr == LR:
                # the same list through the reference's sequential loop: the order alone does it
                u, i, r = ctx.debug_epoch_list()
                Uo, Vo = U0.copy(), V0.copy()
                orc.sgd_pass(Uo, Vo, u, i, r, None, lr, 0.02, 0.02, orc.ARITH_F32, orc.DOT_TREE)
                assert not _finite(Uo, Vo)
        # all waves in flight at the halved rate: finite, and the epoch made progress
        ctx.set_factors(U0, V0)
        before = ctx.rmse(mfx.MAT_TRAIN)
        ctx.sgd_epoch(LR / 2, 0.02, 0.02, mode=mfx.SGD_TILED, order=mfx.ORDER_DEVICE, arith=mfx.ARITH_F32, seed=1, epoch=0)
        after = ctx.rmse(mfx.MAT_TRAIN)
    assert not res[LR] and res[LR / 2]
    print("tiled epoch at lr %g: train RMSE %.4f -> %.4f" % (LR / 2, before, after))
    assert np.isfinite(after) and after < before
