"""ModelDropoutSigmoid (--algo=TMF, modelDropoutSigmoid.cpp): rank truncated per rating by the frequency of its
rarer side.  Device visit and truncated evaluation against the oracle, bit-exact in list order."""
import numpy as np
import pytest

from matfac_amd import Ctx, mfx, synth
from oracle import binding as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K,rho,alpha", [(5, 1.0, 0.0), (16, 2.0, -0.5), (40, 0.5, 0.3), (64, 1.0, 0.0), (100, 3.0, 0.2)])
def test_truncated_rank_visit_and_evaluation(K, rho, alpha):
    d = synth.make(dict(nU=400, nI=150, nnz=12000, K=K), seed=K + 1)
    tr, te = d["train"], d["test"]
    nU, nI = d["nUsers"], max(d["nItems"], tr.ncols)
    rng = np.random.default_rng(K)
    U0 = rng.normal(0, 0.4, (nU, K)).astype(np.float32)
    V0 = rng.normal(0, 0.4, (nI, K)).astype(np.float32)
    oU, oI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
    uf = np.zeros(nU); uf[:tr.nrows] = np.diff(tr.rowptr)
    itf = np.zeros(nI); itf[:tr.ncols] = np.bincount(tr.rowind, minlength=tr.ncols)
    both = np.concatenate([uf[:tr.nrows], itf[:tr.ncols]])
    mean, std = both.mean(), np.sqrt(((both - both.mean()) ** 2).sum() / len(both))        # meanStdDev, util.cpp:278-294
    ru, ri = orc.tmf_ranks(uf, mean, std, rho, alpha, K), orc.tmf_ranks(itf, mean, std, rho, alpha, K)
    assert ru.min() >= 1 and ru.max() <= K and len(np.unique(np.concatenate([ru, ri]))) > 1           # ranks really vary
    order = np.arange(tr.nnz, dtype=np.uint64)
    orc.MT(4).shuffle_u64(order)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_csr(mfx.MAT_TEST, te.nrows, nI, te.rowptr, te.rowind, te.rowval)
        ctx.set_model(nU, nI, K)
        ctx.set_factors(U0, V0)
        ctx.compute_invalid()
        full = ctx.eval(mfx.MAT_TEST)
        ctx.set_tmf(uf.astype(np.float32), ru, itf.astype(np.float32), ri)
        e0 = ctx.eval(mfx.MAT_TEST)
        ctx.sgd_set_order(order)
        ctx.sgd_epoch(0.004, 0.05, 0.03, mode=mfx.SGD_LEVELS, order=mfx.ORDER_HOST)     # dataflow replay of the list ...
        Ul, Vl = ctx.get_factors()
        ctx.set_factors(U0, V0)
        ctx.sgd_epoch(0.004, 0.05, 0.03, mode=mfx.SGD_SERIAL, order=mfx.ORDER_HOST)     # ... and one group in list order: the same bits
        Us, Vs = ctx.get_factors()
        assert np.array_equal(Ul, Us) and np.array_equal(Vl, Vs)
        U, V = ctx.get_factors()
        e1 = ctx.eval(mfx.MAT_TRAIN)
        keep_i = (np.arange(nI) % 2).astype(np.uint8)
        ef = ctx.eval_filtered(mfx.MAT_TEST, None, keep_i)
        with pytest.raises(mfx.MfxError):
            ctx.sgd_epoch(0.004, 0.05, 0.03, mode=mfx.SGD_USERS, order=mfx.ORDER_NATURAL)
        ctx.set_factors(U0, V0)
        first = np.unique(tr.rowids(), return_index=True)[1]
        keep = first[np.unique(tr.rowind[first], return_index=True)[1]].astype(np.uint64)
        ctx.sgd_set_order(keep)
        ctx.sgd_epoch(0.004, 0.05, 0.03, mode=mfx.SGD_HOGWILD, order=mfx.ORDER_HOST)
        Uh, Vh = ctx.get_factors()
        ctx.set_tmf()
        ctx.set_factors(U0, V0)
        again = ctx.eval(mfx.MAT_TEST)
    _, s0, n0 = orc.rmse_tmf(U0, V0, nU, nI, te.nrows, te.rowptr, te.rowind, te.rowval, oU, oI, uf, itf, ru, ri, orc.DOT_TREE)
    assert e0.n == n0 and abs(e0.sse - s0) <= 1e-12 * s0 and e0.sse != full.sse
    assert (again.n, again.sse) == (full.n, full.sse)
    Uo, Vo = U0.copy(), V0.copy()
    orc.sgd_pass_tmf(Uo, Vo, tr.rowids(), tr.rowind, tr.rowval, order, 0.004, 0.05, 0.03, uf, itf, ru, ri, orc.DOT_TREE)
    assert np.isfinite(Uo).all() and np.array_equal(U, Uo) and np.array_equal(V, Vo)
    _, s1, n1 = orc.rmse_tmf(Uo, Vo, nU, nI, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, oU, oI, uf, itf, ru, ri, orc.DOT_TREE)
    assert e1.n == n1 and abs(e1.sse - s1) <= 1e-12 * s1
    _, sf, nf = orc.rmse_tmf(Uo, Vo, nU, nI, te.nrows, te.rowptr, te.rowind, te.rowval, oU, (oI | (1 - keep_i)).astype(np.uint8), uf, itf,
                             ru, ri, orc.DOT_TREE)
    assert ef.n == nf and abs(ef.sse - sf) <= 1e-12 * sf
    Uc, Vc = U0.copy(), V0.copy()
    orc.sgd_pass_tmf(Uc, Vc, tr.rowids(), tr.rowind, tr.rowval, keep, 0.004, 0.05, 0.03, uf, itf, ru, ri, orc.DOT_TREE)
    assert np.array_equal(Uh, Uc) and np.array_equal(Vh, Vc)
    # dimensions beyond a row's largest rank are never touched
    top_u = np.zeros(nU, int)
    np.maximum.at(top_u, tr.rowids(), np.where(uf[tr.rowids()] < itf[tr.rowind], ru[tr.rowids()], ri[tr.rowind]))
    for u in range(0, nU, 37):
        assert np.array_equal(U[u, top_u[u]:], U0[u, top_u[u]:])


def _perm_matrix(n, seed):
    """n x n with one rating per user and per item: every visit of an epoch is conflict-free in any order."""
    rng = np.random.default_rng(seed)
    cols = rng.permutation(n).astype(np.int32)
    vals = (rng.integers(1, 11, n) * 0.5).astype(np.float32)
    return synth.CSR(n, n, np.arange(n + 1, dtype=np.int64), cols, vals)


@pytest.mark.parametrize("K", [10, 64, 128])
def test_tiled_kernel_with_truncated_ranks_and_with_weights(K):
    """The sibling models on MFX_SGD_TILED: on a conflict-free matrix the epoch equals the oracle's list-order pass
    up to the fixed-point representation of the owned item rows (2e-7), whatever order the slots are visited in."""
    n = 3000
    tr = _perm_matrix(n, K)
    rng = np.random.default_rng(K)
    U0 = rng.normal(0, 0.3, (n, K)).astype(np.float32)
    V0 = rng.normal(0, 0.3, (n, K)).astype(np.float32)
    # artificial popularity tables (the matrix itself has one rating per row): what matters is that they vary
    uf = rng.integers(1, 400, n).astype(np.float64)
    itf = rng.integers(1, 400, n).astype(np.float64)
    both = np.concatenate([uf, itf])
    ru, ri = orc.tmf_ranks(uf, both.mean(), both.std(), 1.0, 0.0, K), orc.tmf_ranks(itf, both.mean(), both.std(), 1.0, 0.0, K)
    pu, pi = uf / uf.sum(), itf / itf.sum()
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, n, n, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_model(n, n, K)
        ctx.set_factors(U0, V0)
        ctx.set_tmf(uf.astype(np.float32), ru, itf.astype(np.float32), ri)
        ctx.sgd_epoch(0.01, 0.05, 0.02, mode=mfx.SGD_TILED, seed=3, epoch=1)
        Ut, Vt = ctx.get_factors()
        ctx.set_tmf()
        ctx.set_factors(U0, V0)
        ctx.sgd_set_ifw(uf.astype(np.float32), pu.astype(np.float32), itf.astype(np.float32), pi.astype(np.float32), 2000.0)
        ctx.sgd_epoch(0.01, 0.05, 0.02, mode=mfx.SGD_TILED, seed=3, epoch=1)
        Uw, Vw = ctx.get_factors()
        ctx.sgd_set_ifw()
        ctx.set_factors(U0, V0)
        ctx.sgd_epoch(0.01, 0.05, 0.02, mode=mfx.SGD_TILED, seed=3, epoch=1, arith=mfx.ARITH_REF64)    # plain again
        Up, Vp = ctx.get_factors()
    Uo, Vo = U0.copy(), V0.copy()
    orc.sgd_pass_tmf(Uo, Vo, tr.rowids(), tr.rowind, tr.rowval, None, 0.01, 0.05, 0.02, uf, itf, ru, ri, orc.DOT_TREE)
    assert np.abs(Ut - Uo).max() <= 2e-7 and np.abs(Vt - Vo).max() <= 2e-7
    rank = np.where(uf < itf[tr.rowind], ru, ri[tr.rowind])                      # user u's only rating decides its rank
    for u in range(0, n, 97):
        assert np.array_equal(Ut[u, rank[u]:], U0[u, rank[u]:])                  # dimensions beyond the rank: bit for bit
    Uo, Vo = U0.copy(), V0.copy()
    pop = (uf, itf, pu.astype(np.float32).astype(np.float64), pi.astype(np.float32).astype(np.float64))
    orc.sgd_pass_ifw(Uo, Vo, tr.rowids(), tr.rowind, tr.rowval, None, 0.01, 0.05, 0.02, pop, 2000.0, orc.DOT_TREE)
    assert np.abs(Uw - Uo).max() <= 2e-7 and np.abs(Vw - Vo).max() <= 2e-7
    assert np.abs(Uw - Ut).max() > 1e-4                                          # and the two models differ
    Uo, Vo = U0.copy(), V0.copy()
    orc.sgd_pass(Uo, Vo, tr.rowids(), tr.rowind, tr.rowval, None, 0.01, 0.05, 0.02, orc.ARITH_REF64, orc.DOT_TREE)
    assert np.abs(Up - Uo).max() <= 2e-7 and np.abs(Vp - Vo).max() <= 2e-7


def test_poisson_dropout_visits_match_oracle_and_draws_look_poisson():
    """ModelPoissonDropout (--algo=TMFDropout): the rank of a visit is a Poisson(lambda) draw -- here a pure function of
    (seed, epoch, user, item), the same on the flat kernels, on the tiled kernel and in the oracle."""
    K = 40
    # the sampler itself: mean and variance of Poisson(lambda) before clipping effects matter
    for lam in (3, 12, 30):
        d = np.array([orc.poisson_rank(lam, 7, e, u, 5 * u + 1, 400) for e in range(4) for u in range(2500)], float)
        assert abs(d.mean() - lam) < 0.15 * np.sqrt(lam) and abs(d.var() - lam) < 0.15 * lam, (lam, d.mean(), d.var())
    cdf = orc.cdf_ranks(K)
    assert np.all(np.diff(cdf) >= 0) and cdf[0] >= 1 and cdf[-1] == K - 1
    n = 3000
    tr = _perm_matrix(n, 77)
    rng = np.random.default_rng(5)
    U0 = rng.normal(0, 0.3, (n, K)).astype(np.float32)
    V0 = rng.normal(0, 0.3, (n, K)).astype(np.float32)
    uf = rng.integers(1, 400, n).astype(np.float64)
    itf = rng.integers(1, 400, n).astype(np.float64)
    both = np.concatenate([uf, itf])
    lu, li = orc.tmf_ranks(uf, both.mean(), both.std(), 1.0, 0.0, K), orc.tmf_ranks(itf, both.mean(), both.std(), 1.0, 0.0, K)
    eu, ei = np.minimum(cdf[lu - 1] + 1, K).astype(np.int32), np.minimum(cdf[li - 1] + 1, K).astype(np.int32)   # estRating's count
    outs = {}
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, n, n, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_model(n, n, K)
        ctx.compute_invalid()
        with pytest.raises(mfx.MfxError):
            ctx.set_tmf_dropout(lu, li, 9)                       # needs the evaluation ranks first
        ctx.set_tmf(uf.astype(np.float32), eu, itf.astype(np.float32), ei)
        ctx.set_tmf_dropout(lu, li, 9)
        for name, mode, order in (("serial", mfx.SGD_SERIAL, mfx.ORDER_NATURAL), ("flat", mfx.SGD_HOGWILD, mfx.ORDER_DEVICE),
                                  ("tiled", mfx.SGD_TILED, mfx.ORDER_DEVICE)):
            ctx.set_factors(U0, V0)
            for ep in (0, 1):
                ctx.sgd_epoch(0.01, 0.05, 0.02, mode=mode, order=order, seed=3, epoch=ep)
            outs[name] = ctx.get_factors()
        e = ctx.eval(mfx.MAT_TRAIN)
        ctx.set_tmf_dropout()
    Uo, Vo = U0.copy(), V0.copy()
    for ep in (0, 1):
        orc.sgd_pass_tmfd(Uo, Vo, tr.rowids(), tr.rowind, tr.rowval, None, 0.01, 0.05, 0.02, uf, itf, lu, li, 9, ep, orc.DOT_TREE)
    assert np.array_equal(outs["serial"][0], Uo) and np.array_equal(outs["serial"][1], Vo)
    assert np.array_equal(outs["flat"][0], Uo) and np.array_equal(outs["flat"][1], Vo)           # conflict-free matrix
    assert np.abs(outs["tiled"][0] - Uo).max() <= 4e-7 and np.abs(outs["tiled"][1] - Vo).max() <= 4e-7
    # the evaluation uses the cdf ranks, not a draw
    oU, oI = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    _, s, cnt = orc.rmse_tmf(outs["tiled"][0], outs["tiled"][1], n, n, n, tr.rowptr, tr.rowind, tr.rowval, oU, oI, uf, itf, eu, ei, orc.DOT_TREE)
    assert e.n == cnt and abs(e.sse - s) <= 1e-12 * s
