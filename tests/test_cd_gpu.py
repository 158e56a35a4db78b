"""ModelMF::trainCCD (modelMF.cpp:1528-1605) on the device, through the C ABI, against the oracle's sequential
restatement fed with the same per-row factor orders.  Double sums are associated differently (lanes, then a
tree), so the comparison is to a few float ulps, not bit-exact; the two residual views must be bit-identical
to each other."""
import numpy as np
import pytest

from matfac_amd import Ctx, mfx, synth
from oracle import binding as orc

pytestmark = pytest.mark.gpu


def setup(nU, nI, nnz, K, seed=4):
    d = synth.make(dict(nU=nU, nI=nI, nnz=nnz, K=K), seed=seed)
    tr = d["train"]
    nI = max(d["nItems"], tr.ncols)
    cp, ci, cv = orc.create_col_index(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
    U0, V0 = synth.init_factors(seed, nU, nI, K)
    V0 = (V0 * 30).astype(np.float32)              # the reference starts CCD from U = 0 and whatever V is
    invU, invI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
    return tr, nU, nI, (cp, ci, cv), U0, V0, invU, invI


# (users, items, ratings, K): wavefront rows only; workgroup rows in registers; 1024-thread rows; items beyond 16384 ratings (streaming)
@pytest.mark.parametrize("nU,nI,nnz,K", [(300, 70, 4000, 8), (3000, 400, 150000, 10), (9000, 30, 200000, 64), (60000, 24, 800000, 4),
                                         (500, 2000, 30000, 5), (2000, 300, 60000, 128)])
def test_ccd_iterations_match_oracle_with_the_same_factor_orders(nU, nI, nnz, K):
    tr, nU, nI, (cp, ci, cv), U0, V0, invU, invI = setup(nU, nI, nnz, K)
    uReg, iReg = 0.3, 0.2
    Uo = np.zeros_like(U0)
    Vo = V0.copy()
    rr, rc = tr.rowval.copy(), cv.copy()
    mt = orc.MT(7)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval, cp, ci, cv)
        ctx.set_model(nU, nI, K)
        ctx.set_factors(U0, V0)
        ctx.ccd_begin()
        for it in range(3):
            uo = np.zeros((nU, K), np.uint16)
            io = np.zeros((nI, K), np.uint16)
            orc.ccd_iter(Uo, Vo, nU, nI, tr.ncols, tr.rowptr, tr.rowind, rr, cp, ci, rc, invU, invI, uReg, iReg, mt, uo, io)
            ctx.ccd_sweep(mfx.SIDE_USERS, uReg, uo[:tr.nrows])
            ctx.ccd_sweep(mfx.SIDE_ITEMS, iReg, io[:tr.ncols])
            U, V = ctx.get_factors()
            grr, grc = ctx.debug_ccd_residuals(tr.nnz)
            scale = max(1.0, float(np.abs(Vo).max()), float(np.abs(Uo).max()))
            assert np.allclose(U, Uo, rtol=2e-5, atol=2e-6 * scale), (it, np.abs(U - Uo).max())
            assert np.allclose(V, Vo, rtol=2e-5, atol=2e-6 * scale), (it, np.abs(V - Vo).max())
            assert np.allclose(grr, rr, rtol=0, atol=2e-5 * scale)
            # the device's two views hold the same numbers
            e_of_d = np.lexsort((tr.rowids(), tr.rowind))            # CSR position of each column-view entry
            assert np.array_equal(grc, grr[e_of_d])
        ctx.ccd_end()
    # rows without ratings are never touched: U stays 0 there (uFac zeroed), V keeps its start
    assert np.all(U[invU.astype(bool)] == 0) and np.array_equal(V[invI.astype(bool)], V0[invI.astype(bool)])


def test_ccd_device_orders_descend_and_keep_the_residual_consistent():
    tr, nU, nI, (cp, ci, cv), U0, V0, invU, invI = setup(4000, 500, 200000, 32, seed=9)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)     # column view built on the device
        ctx.set_model(nU, nI, 32)
        ctx.set_factors(U0, V0)
        ctx.compute_invalid()
        ctx.ccd_begin()
        objs = []
        for it in range(4):
            ctx.ccd_sweep(mfx.SIDE_USERS, 0.5, None, seed=3, it=it)
            objs.append(ctx.objective(0.5, 0.5))
            ctx.ccd_sweep(mfx.SIDE_ITEMS, 0.5, None, seed=3, it=it)
            objs.append(ctx.objective(0.5, 0.5))
        U, V = ctx.get_factors()
        grr, _ = ctx.debug_ccd_residuals(tr.nnz)
        ctx.ccd_end()
        with pytest.raises(mfx.MfxError):
            ctx.ccd_sweep(mfx.SIDE_USERS, 0.5)
    # every coordinate step minimises the regularised objective along its coordinate
    assert all(b <= a * (1 + 1e-6) for a, b in zip(objs, objs[1:])), objs
    assert objs[-1] < 0.5 * objs[0]
    est = np.einsum("ek,ek->e", U[tr.rowids()].astype(np.float64), V[tr.rowind].astype(np.float64))
    assert np.allclose(grr, tr.rowval - est, atol=5e-4)


def test_ccd_rejects_a_column_view_that_is_not_the_stable_transpose():
    tr, nU, nI, (cp, ci, cv), U0, V0, invU, invI = setup(300, 70, 4000, 8)
    j = int(np.argmax(np.diff(cp)))                 # swap two entries inside the fullest column
    ci2, cv2 = ci.copy(), cv.copy()
    a = cp[j]
    ci2[[a, a + 1]] = ci2[[a + 1, a]]
    cv2[[a, a + 1]] = cv2[[a + 1, a]]
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval, cp, ci2, cv2)
        ctx.set_model(nU, nI, 8)
        with pytest.raises(mfx.MfxError) as e:
            ctx.ccd_begin()
        assert "stable transpose" in str(e.value)
