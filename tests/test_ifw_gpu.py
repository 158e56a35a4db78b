"""ModelInvPopMF (--algo=IFWMF, modelInvPopMF.cpp): the weighted SGD visit and objective on the device against the
oracle's restatement -- bit-exact in list order (the weight is a pure function of the user's and the item's
popularity pair, which the kernel gathers next to the rows)."""
import numpy as np
import pytest

from matfac_amd import Ctx, mfx, synth
from oracle import binding as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K,rho", [(5, 100.0), (16, 3000.0), (40, 10.0), (64, 1000.0), (100, 500.0)])
def test_weighted_visit_and_objective(K, rho):
    d = synth.make(dict(nU=400, nI=150, nnz=12000, K=K), seed=K)
    tr = d["train"]
    nU, nI = d["nUsers"], max(d["nItems"], tr.ncols)
    rng = np.random.default_rng(K)
    U0 = rng.normal(0, 0.4, (nU, K)).astype(np.float32)
    V0 = rng.normal(0, 0.4, (nI, K)).astype(np.float32)
    oU, oI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
    pop = orc.ifw_pop(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, oU, oI)
    uf, itf, pu, pi = pop
    pad = lambda a, n: np.concatenate([a, np.zeros(n - len(a))]).astype(np.float32)
    order = np.arange(tr.nnz, dtype=np.uint64)
    orc.MT(3).shuffle_u64(order)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_model(nU, nI, K)
        ctx.set_factors(U0, V0)
        ctx.compute_invalid()
        with pytest.raises(mfx.MfxError):
            ctx.eval_ifw()
        ctx.sgd_set_ifw(pad(uf, nU), pad(pu, nU), pad(itf, nI), pad(pi, nI), rho)
        e0 = ctx.eval_ifw()
        ctx.sgd_set_order(order)
        ctx.sgd_epoch(0.004, 0.05, 0.03, mode=mfx.SGD_SERIAL, order=mfx.ORDER_HOST)
        U, V = ctx.get_factors()
        e1 = ctx.eval_ifw()
        ctx.set_factors(U0, V0)                                      # the same epoch on the dataflow schedule: the same bits
        ctx.sgd_epoch(0.004, 0.05, 0.03, mode=mfx.SGD_LEVELS, order=mfx.ORDER_HOST)
        Ul, Vl = ctx.get_factors()
        assert np.array_equal(Ul, U) and np.array_equal(Vl, V)
        with pytest.raises(mfx.MfxError):
            ctx.sgd_epoch(0.004, 0.05, 0.03, mode=mfx.SGD_USERS, order=mfx.ORDER_NATURAL)
        # parallel kernel on a conflict-free batch
        ctx.set_factors(U0, V0)
        first = np.unique(tr.rowids(), return_index=True)[1]
        keep = first[np.unique(tr.rowind[first], return_index=True)[1]].astype(np.uint64)
        ctx.sgd_set_order(keep)
        ctx.sgd_epoch(0.004, 0.05, 0.03, mode=mfx.SGD_HOGWILD, order=mfx.ORDER_HOST)
        Uh, Vh = ctx.get_factors()
        ctx.sgd_set_ifw()                                            # off again: the plain update
        ctx.set_factors(U0, V0)
        ctx.sgd_epoch(0.002, 0.05, 0.03, mode=mfx.SGD_SERIAL, order=mfx.ORDER_NATURAL, arith=mfx.ARITH_REF64)
        Up, _ = ctx.get_factors()
    o0, w0 = orc.objective_ifw(U0, V0, nU, nI, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, oU, oI, 0.05, 0.03, pop, rho, orc.DOT_TREE)
    assert abs(e0.sse - w0) <= 1e-12 * w0 and e0.n == tr.nnz
    Uo, Vo = U0.copy(), V0.copy()
    orc.sgd_pass_ifw(Uo, Vo, tr.rowids(), tr.rowind, tr.rowval, order, 0.004, 0.05, 0.03, pop, rho, orc.DOT_TREE)
    assert np.isfinite(Uo).all() and np.array_equal(U, Uo) and np.array_equal(V, Vo)
    o1, w1 = orc.objective_ifw(Uo, Vo, nU, nI, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, oU, oI, 0.05, 0.03, pop, rho, orc.DOT_TREE)
    assert abs(e1.sse - w1) <= 1e-12 * w1
    assert abs((e1.sse + 0.05 * e1.unorm2 + 0.03 * e1.inorm2) - o1) <= 1e-6 * o1
    assert w1 < w0                                                    # the weighted loss went down
    Uc, Vc = U0.copy(), V0.copy()
    orc.sgd_pass_ifw(Uc, Vc, tr.rowids(), tr.rowind, tr.rowval, keep, 0.004, 0.05, 0.03, pop, rho, orc.DOT_TREE)
    assert np.array_equal(Uh, Uc) and np.array_equal(Vh, Vc)
    Uq, Vq = U0.copy(), V0.copy()
    orc.sgd_pass(Uq, Vq, tr.rowids(), tr.rowind, tr.rowval, None, 0.002, 0.05, 0.03, orc.ARITH_REF64, orc.DOT_TREE)
    assert np.array_equal(Up, Uq) and np.isfinite(Up).all()
