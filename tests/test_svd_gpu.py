"""The pieces of ModelMF::trainSGDParSVD (modelMF.cpp:353-557) on the device:
 * the truncated SVD that initialises the factors, against numpy's dense SVD (SVDLIBC, which the reference
   calls, is not in the tree: singular values and the spanned subspaces are what is comparable);
 * the SGD visit with the per-dimension regulariser, bit-exact against the oracle in list order;
 * Model::objectiveSing's weighted norms."""
import numpy as np
import pytest

from matfac_amd import Ctx, mfx, synth
from oracle import binding as orc

pytestmark = pytest.mark.gpu


def dense(tr):
    R = np.zeros((tr.nrows, tr.ncols))
    R[tr.rowids(), tr.rowind] = tr.rowval
    return R


@pytest.mark.parametrize("nU,nI,nnz,K", [(600, 250, 30000, 16), (2000, 300, 120000, 40), (300, 1200, 40000, 8), (120, 90, 3000, 70)])
def test_truncated_svd_matches_dense_svd(nU, nI, nnz, K):
    d = synth.make(dict(nU=nU, nI=nI, nnz=nnz, K=K), seed=K)
    tr = d["train"]
    nI = max(d["nItems"], tr.ncols)
    R = dense(tr)
    Ud, sd, Vtd = np.linalg.svd(R, full_matrices=False)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_model(nU, nI, K)
        V0 = np.full((nI, K), 7.0, np.float32)
        ctx.set_factors(np.zeros((nU, K), np.float32), V0)
        sig = ctx.svd_init(power_iters=12, oversample=12, seed=3)
        U, V = ctx.get_factors()
    assert np.all(np.diff(sig) <= 1e-4 * sig[0])                                   # descending
    assert np.allclose(sig, sd[:K], rtol=2e-3, atol=1e-3 * sd[0]), np.abs(sig / sd[:K] - 1).max()
    assert np.allclose(sig[:K // 2], sd[:K // 2], rtol=1e-4)
    V = V[:tr.ncols]
    # orthonormal columns, R v_k = sigma_k u_k
    assert np.abs(U.T @ U - np.eye(K)).max() < 2e-3 and np.abs(V.T @ V - np.eye(K)).max() < 2e-3
    assert np.abs(R @ V - U * sig).max() < 2e-3 * sd[0]
    # the rank-K approximation is as good as the optimal one
    best = np.linalg.norm(R - (Ud[:, :K] * sd[:K]) @ Vtd[:K])
    got = np.linalg.norm(R - (U * sig) @ V.T)
    assert got <= best * (1 + 1e-3), (got, best)
    # the leading subspace agrees with numpy's (principal angles)
    k2 = max(1, K // 2)
    if sd[k2 - 1] > 1.05 * sd[k2]:
        c = np.linalg.svd(Vtd[:k2] @ V[:, :k2], compute_uv=False)
        assert c.min() > 1 - 1e-3


def test_items_beyond_the_train_matrix_keep_their_rows():
    d = synth.make(dict(nU=200, nI=80, nnz=4000, K=6), seed=2)
    tr = d["train"]
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_model(200, tr.ncols + 5, 6)
        V0 = np.full((tr.ncols + 5, 6), 7.0, np.float32)
        ctx.set_factors(np.zeros((200, 6), np.float32), V0)
        ctx.svd_init()
        _, V = ctx.get_factors()
    assert np.all(V[tr.ncols:] == 7.0) and not np.any(V[:tr.ncols] == 7.0)


@pytest.mark.parametrize("K", [5, 16, 40, 64, 100])
def test_dimreg_sgd_serial_is_bit_exact_and_objective_sing_matches(K):
    d = synth.make(dict(nU=300, nI=120, nnz=9000, K=K), seed=5)
    tr = d["train"]
    nU, nI = d["nUsers"], max(d["nItems"], tr.ncols)
    rng = np.random.default_rng(K)
    U0 = rng.normal(0, 0.3, (nU, K)).astype(np.float32)
    V0 = rng.normal(0, 0.3, (nI, K)).astype(np.float32)
    sing = np.sort(rng.uniform(1, 300, K)).astype(np.float32)[::-1].copy()
    sing_a, sing_b = np.float32(0.01), np.float32(0.02)
    regk = ((sing_a + np.float32(1)) / (sing_b + sing)).astype(np.float32)       # modelMF.cpp:498
    oU, oI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_model(nU, nI, K)
        ctx.set_factors(U0, V0)
        ctx.compute_invalid()
        ctx.sgd_set_dim_reg(regk)
        ctx.sgd_epoch(0.01, 9.0, 9.0, mode=mfx.SGD_LEVELS, order=mfx.ORDER_NATURAL)       # dataflow replay ...
        Ul, Vl = ctx.get_factors()
        ctx.set_factors(U0, V0)
        ctx.sgd_epoch(0.01, 9.0, 9.0, mode=mfx.SGD_SERIAL, order=mfx.ORDER_NATURAL)       # ... == one group in list order; uReg/iReg are ignored
        Us, Vs = ctx.get_factors()
        assert np.array_equal(Ul, Us) and np.array_equal(Vl, Vs)
        U, V = ctx.get_factors()
        e = ctx.eval_weighted(mfx.MAT_TRAIN, sing)
        with pytest.raises(mfx.MfxError):
            ctx.sgd_epoch(0.01, 0, 0, mode=mfx.SGD_USERS, order=mfx.ORDER_NATURAL)
        # the parallel kernel: a conflict-free batch is exact too
        ctx.set_factors(U0, V0)
        n = min(nU, nI)
        first = np.unique(tr.rowids(), return_index=True)[1]
        keep = first[np.unique(tr.rowind[first], return_index=True)[1]]              # distinct users AND distinct items
        ctx.sgd_set_order(keep.astype(np.uint64))
        ctx.sgd_epoch(0.01, 0, 0, mode=mfx.SGD_HOGWILD, order=mfx.ORDER_HOST)
        Uh, Vh = ctx.get_factors()
        ctx.sgd_set_dim_reg(None)
        ctx.set_factors(U0, V0)
        ctx.sgd_epoch(0.01, 0.5, 0.5, mode=mfx.SGD_SERIAL, order=mfx.ORDER_NATURAL, arith=mfx.ARITH_REF64F)
        Up, _ = ctx.get_factors()
    Uo, Vo = U0.copy(), V0.copy()
    orc.sgd_pass_dimreg(Uo, Vo, tr.rowids(), tr.rowind, tr.rowval, None, 0.01, regk, orc.DOT_TREE)
    assert np.array_equal(U, Uo) and np.array_equal(V, Vo)
    obj, sse, ur, ir = orc.objective_sing(Uo, Vo, nU, nI, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, oU, oI, sing, orc.DOT_TREE)
    assert abs(e.sse - sse) <= 1e-12 * sse and abs(e.unorm2 - ur) <= 1e-10 * ur and abs(e.inorm2 - ir) <= 1e-10 * ir
    Uc, Vc = U0.copy(), V0.copy()
    orc.sgd_pass_dimreg(Uc, Vc, tr.rowids(), tr.rowind, tr.rowval, keep.astype(np.uint64), 0.01, regk, orc.DOT_TREE)
    assert np.array_equal(Uh, Uc) and np.array_equal(Vh, Vc)
    Uq, Vq = U0.copy(), V0.copy()                                       # clearing the regulariser restores the scalar path
    orc.sgd_pass(Uq, Vq, tr.rowids(), tr.rowind, tr.rowval, None, 0.01, 0.5, 0.5, orc.ARITH_REF64F, orc.DOT_TREE)
    assert np.array_equal(Up, Uq)


@pytest.mark.parametrize("K", [10, 64, 128])
def test_dimreg_on_the_tiled_kernel(K):
    """The per-dimension regulariser as a variant of MFX_SGD_TILED: on a conflict-free matrix the epoch equals the oracle's
    list-order pass up to the fixed-point item rows (2e-7)."""
    n = 3000
    rng = np.random.default_rng(K)
    tr = synth.CSR(n, n, np.arange(n + 1, dtype=np.int64), rng.permutation(n).astype(np.int32), (rng.integers(1, 11, n) * 0.5).astype(np.float32))
    U0 = rng.normal(0, 0.3, (n, K)).astype(np.float32)
    V0 = rng.normal(0, 0.3, (n, K)).astype(np.float32)
    regk = (np.float32(1.01) / (np.float32(0.02) + np.sort(rng.uniform(1, 300, K)).astype(np.float32)[::-1])).astype(np.float32)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, n, n, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_model(n, n, K)
        ctx.set_factors(U0, V0)
        ctx.sgd_set_dim_reg(regk)
        ctx.sgd_epoch(0.01, 9.0, 9.0, mode=mfx.SGD_TILED, seed=3, epoch=1)
        U, V = ctx.get_factors()
        ctx.sgd_set_dim_reg(None)
    Uo, Vo = U0.copy(), V0.copy()
    orc.sgd_pass_dimreg(Uo, Vo, tr.rowids(), tr.rowind, tr.rowval, None, 0.01, regk, orc.DOT_TREE)
    assert np.abs(U - Uo).max() <= 2e-7 and np.abs(V - Vo).max() <= 2e-7 and np.abs(U - U0).max() > 1e-4
