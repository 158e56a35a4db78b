"""GPU parity of the ALS half-sweep (MFMA Gramian + in-register LDL^T) against the oracle's
restatement of modelMF.cpp:805-880 (scalar Gramian + Eigen-style pivoted LDLT).

fp32 tolerance: both sides solve (Q^T Q + reg I) x = Q^T r in fp32 with different but backward
stable algorithms, so they agree to ~eps * cond(A); the tests bound the row-wise relative error by
that and check the normal-equation residual of the GPU result in float64."""
import numpy as np
import pytest

from matfac_amd import Ctx, mfx, synth
from oracle import binding as orc
from tests.util import load_ctx

pytestmark = pytest.mark.gpu


def _data(nU, nI, nnz, seed):
    d = synth.make(dict(nU=nU, nI=nI, nnz=nnz, K=0), seed=seed)
    tr = d["train"]
    cp, ci, cv = orc.create_col_index(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
    return d, tr, (cp, ci, cv)


# K > 64 runs the blocked kernels of als_wide.hip (64x64 block pairs + LDL^T in LDS)
@pytest.mark.parametrize("K,reg", [(64, 5.0), (64, 0.5), (64, 0.05), (32, 1.0), (10, 1.0), (48, 2.0), (128, 2.0), (100, 1.0),
                                   (192, 3.0), (256, 5.0), (65, 0.5)])
def test_half_sweeps_match_oracle(K, reg):
    d, tr, (cp, ci, cv) = _data(1500, 400, 60000, seed=K)
    nU, nI = d["nUsers"], d["nItems"]
    rng = np.random.default_rng(K)
    U0 = rng.normal(0, 0.3, (nU, K)).astype(np.float32)
    V0 = rng.normal(0, 0.3, (nI, K)).astype(np.float32)
    assert np.bincount(tr.rowind).max() > 1024        # exercises the split-row (slab) path
    with Ctx(0) as ctx:
        invU, invI = load_ctx(ctx, d, K, U0, V0)
        ctx.als_half_sweep(mfx.SIDE_USERS, reg)
        U1, V1 = ctx.get_factors()
        ctx.als_half_sweep(mfx.SIDE_ITEMS, reg)
        U2, V2 = ctx.get_factors()
    assert np.array_equal(V1, V0) and np.array_equal(U2, U1)
    Uo, Vo = U0.copy(), V0.copy()
    orc.als_half(0, Uo, Vo, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, reg, nthreads=4)
    # users: relative error per row
    scale = np.maximum(np.linalg.norm(Uo, axis=1), 1e-6)
    relu = np.linalg.norm(U1 - Uo, axis=1) / scale
    # cond(A) <= (sigma_max^2 + reg)/reg ; eps = 6e-8 ; generous constant
    tol = 2e-4 if reg >= 0.5 else 5e-3
    assert relu.max() < tol, relu.max()
    orc.als_half(1, Vo, U1, tr.ncols, cp, ci, cv, invI, reg, nthreads=4)   # items from the GPU's users
    scale = np.maximum(np.linalg.norm(Vo, axis=1), 1e-6)
    reli = np.linalg.norm(V2[:tr.ncols] - Vo[:tr.ncols], axis=1) / scale[:tr.ncols]
    assert reli.max() < tol, reli.max()
    # invalid rows keep their old factors (modelMF.cpp:809-811, 847-849)
    assert np.array_equal(U1[invU.astype(bool)], U0[invU.astype(bool)])
    assert np.array_equal(V2[invI.astype(bool)], V0[invI.astype(bool)])
    # float64 residual of the normal equations for the heaviest and a few random rows
    deg = np.diff(tr.rowptr)
    for u in list(np.argsort(deg)[-3:]) + list(rng.integers(0, nU, 5)):
        if invU[u]:
            continue
        sl = slice(tr.rowptr[u], tr.rowptr[u + 1])
        Q = V0[tr.rowind[sl]].astype(np.float64)
        r = tr.rowval[sl].astype(np.float64)
        A = Q.T @ Q + reg * np.eye(K)
        b = Q.T @ r
        res = np.linalg.norm(A @ U1[u].astype(np.float64) - b) / np.linalg.norm(b)
        assert res < 1e-4, (u, res)


@pytest.mark.parametrize("K", [10, 64])
def test_half_sweep_at_the_references_default_regulariser(K):
    """uReg = iReg = 0.01 (main.cpp:29-31), the reference's default and the worst conditioning it is run at: A = Q^T Q + 0.01 I
    with Q the rated rows.  The device solves without pivoting, Eigen's ldlt() (restated in the oracle, orc_ldlt_solve) pivots on
    the diagonal; both are backward stable on an SPD matrix, so what is compared is (1) each side's OWN normal-equation residual
    ||A x - b|| / ||b|| in float64 -- the device's must not be worse than a small multiple of the pivoted solve's -- and (2) the
    row-wise difference, which is bounded by eps x cond(A) and is reported.  Rows with fewer ratings than K (A singular up to the
    0.01 I) are the hard ones; the three heaviest rows exercise the split-row path."""
    reg = 0.01
    d, tr, (cp, ci, cv) = _data(1500, 400, 60000, seed=100 + K)
    nU, nI = d["nUsers"], d["nItems"]
    rng = np.random.default_rng(K)
    U0 = rng.normal(0, 0.3, (nU, K)).astype(np.float32)
    V0 = rng.normal(0, 0.3, (nI, K)).astype(np.float32)
    with Ctx(0) as ctx:
        invU, invI = load_ctx(ctx, d, K, U0, V0)
        ctx.als_half_sweep(mfx.SIDE_USERS, reg)
        U1, _ = ctx.get_factors()
    Uo, Vo = U0.copy(), V0.copy()
    orc.als_half(0, Uo, Vo, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, reg, nthreads=4)
    deg = np.diff(tr.rowptr)
    res_g, res_o, rel, conds = [], [], [], []
    for u in range(nU):
        if invU[u]:
            continue
        sl = slice(tr.rowptr[u], tr.rowptr[u + 1])
        Q = V0[tr.rowind[sl]].astype(np.float64)
        r = tr.rowval[sl].astype(np.float64)
        A = Q.T @ Q + reg * np.eye(K)
        b = Q.T @ r
        nb = np.linalg.norm(b)
        res_g.append(np.linalg.norm(A @ U1[u].astype(np.float64) - b) / nb)
        res_o.append(np.linalg.norm(A @ Uo[u].astype(np.float64) - b) / nb)
        rel.append(np.linalg.norm(U1[u] - Uo[u]) / max(np.linalg.norm(Uo[u]), 1e-6))
        if len(conds) < 200:
            conds.append(np.linalg.cond(A))
    res_g, res_o, rel = np.array(res_g), np.array(res_o), np.array(rel)
    print("K=%d reg=0.01: residual ||Ax-b||/||b||  device (unpivoted) max %.2e median %.2e | oracle (pivoted LDLT) max %.2e median %.2e | "
          "row-wise |x_dev - x_orc| / |x_orc| max %.2e median %.2e | cond(A) of the first 200 rows up to %.1e"
          % (K, res_g.max(), np.median(res_g), res_o.max(), np.median(res_o), rel.max(), np.median(rel), max(conds)))
    assert np.isfinite(U1).all()
    # Measured (MI355X, round 4): K = 10 residuals 9.3e-7 / 1.8e-7 (max / median) against the oracle's 1.0e-6 / 1.9e-7, rows apart by
    # at most 9.0e-6; K = 64 residuals 1.5e-6 / 1.8e-7 against 1.3e-6 / 2.0e-7, rows apart by at most 1.3e-4 (median 6.7e-5), cond(A)
    # up to 2e3.  The unpivoted solve is as accurate as the pivoted one at the reference's default: no pivoting is added.
    assert res_g.max() <= max(3.0 * res_o.max(), 5e-6)
    assert np.median(res_g) <= max(2.0 * np.median(res_o), 5e-7)
    assert rel.max() < 1e-3 and np.median(rel) < 3e-4


def test_als_iterations_track_oracle_trajectory():
    """ModelMF::trainALS for 5 iterations: objective and validation RMSE per iteration."""
    K, reg = 64, 3.0
    d, tr, (cp, ci, cv) = _data(1200, 500, 50000, seed=21)
    va = d["val"]
    nU, nI = d["nUsers"], d["nItems"]
    U0, V0 = orc.init_factors(1, nU, nI, K)
    Uo, Vo = U0.copy(), V0.copy()
    with Ctx(0) as ctx:
        invU, invI = load_ctx(ctx, d, K, U0, V0)
        for it in range(5):
            ctx.als_half_sweep(mfx.SIDE_USERS, reg)
            ctx.als_half_sweep(mfx.SIDE_ITEMS, reg)
            orc.als_half(0, Uo, Vo, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, reg, nthreads=4)
            orc.als_half(1, Vo, Uo, tr.ncols, cp, ci, cv, invI, reg, nthreads=4)
            g_obj = ctx.objective(reg, reg)
            g_val = ctx.rmse(mfx.MAT_VAL)
            o_obj, *_ = orc.objective(Uo, Vo, nU, nI, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, invI, reg, reg)
            o_val, _, _ = orc.rmse(Uo, Vo, nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, invU, invI)
            assert abs(g_obj - o_obj) <= 1e-4 * o_obj, (it, g_obj, o_obj)
            assert abs(g_val - o_val) <= 1e-5, (it, g_val, o_val)      # <= 1e-5 abs RMSE
        U, V = ctx.get_factors()
    rel = np.abs(U - Uo).max() / np.abs(Uo).max()
    assert rel < 1e-3


def test_nonpositive_ratings_are_skipped():
    """rating <= 0 contributes nothing (modelMF.cpp:819): flip some ratings to 0 / negative."""
    K, reg = 64, 1.0
    d, tr, (cp, ci, cv) = _data(300, 200, 8000, seed=33)
    tr.rowval[::7] = 0.0
    tr.rowval[3::11] = -1.5
    cp, ci, cv = orc.create_col_index(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
    nU, nI = d["nUsers"], d["nItems"]
    rng = np.random.default_rng(1)
    U0 = rng.normal(0, 0.3, (nU, K)).astype(np.float32)
    V0 = rng.normal(0, 0.3, (nI, K)).astype(np.float32)
    with Ctx(0) as ctx:
        invU, invI = load_ctx(ctx, d, K, U0, V0)
        ctx.als_half_sweep(mfx.SIDE_USERS, reg)
        U1, _ = ctx.get_factors()
    Uo, Vo = U0.copy(), V0.copy()
    orc.als_half(0, Uo, Vo, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, reg)
    rel = np.linalg.norm(U1 - Uo, axis=1) / np.maximum(np.linalg.norm(Uo, axis=1), 1e-6)
    assert rel.max() < 2e-4


@pytest.mark.parametrize("K", [64, 128])
def test_zero_row_behind_the_tables_survives_the_other_trainers(K):
    """The ALS accumulation gathers an all-zero row stored behind each factor table for every rating it skips.  Nothing
    else may ever write there: run the other device paths that write U and V, then repeat the skipped-ratings check."""
    reg = 1.0
    d, tr, _ = _data(300, 200, 8000, seed=34)
    tr.rowval[::5] = 0.0
    tr.rowval[2::9] = -2.0
    nU, nI = d["nUsers"], d["nItems"]
    rng = np.random.default_rng(3)
    U0 = rng.normal(0, 0.3, (nU, K)).astype(np.float32)
    V0 = rng.normal(0, 0.3, (nI, K)).astype(np.float32)
    with Ctx(0) as ctx:
        invU, invI = load_ctx(ctx, d, K, U0, V0)
        for mode, order in ((mfx.SGD_TILED, mfx.ORDER_DEVICE), (mfx.SGD_HOGWILD, mfx.ORDER_DEVICE), (mfx.SGD_USERS, mfx.ORDER_NATURAL)):
            ctx.sgd_epoch(0.002, 0.02, 0.02, mode=mode, order=order, seed=1, epoch=0)
        ctx.snapshot_best()
        ctx.ccdpp_begin()
        ctx.ccdpp_rank1(0, reg, reg, add_back=False)
        ctx.ccdpp_end()
        ctx.svd_init(1, 8, 1)
        ctx.restore_best()
        ctx.als_half_sweep(mfx.SIDE_ITEMS, reg)
        ctx.set_factors(U0, V0)                       # known input again; the padding row is not part of it
        ctx.als_half_sweep(mfx.SIDE_USERS, reg)
        U1, _ = ctx.get_factors()
    Uo, Vo = U0.copy(), V0.copy()
    orc.als_half(0, Uo, Vo, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, reg)
    rel = np.linalg.norm(U1 - Uo, axis=1) / np.maximum(np.linalg.norm(Uo, axis=1), 1e-6)
    assert rel.max() < 2e-4


@pytest.mark.parametrize("K", [100, 128, 192, 256])
@pytest.mark.parametrize("solver", ["blocked", "unblocked"])
def test_wide_als_both_solvers_match_oracle(K, solver, monkeypatch):
    """K > 64 has two phase-B kernels (als_wide.hip): the workgroup-per-row LDL^T in registers and the blocked one-wave-per-row
    LDL^T with its block products on the MFMA.  The library picks by block count; both are held to the oracle at every K here."""
    monkeypatch.setenv("MFX_ALS_SOLVER", solver)
    reg = 2.0
    d, tr, (cp, ci, cv) = _data(1500, 300, 70000, seed=K + 3)
    nU, nI = d["nUsers"], d["nItems"]
    rng = np.random.default_rng(K)
    U0 = rng.normal(0, 0.3, (nU, K)).astype(np.float32)
    V0 = rng.normal(0, 0.3, (nI, K)).astype(np.float32)
    assert np.bincount(tr.rowind).max() > 1024            # several segment partials per row
    with Ctx(0) as ctx:
        invU, invI = load_ctx(ctx, d, K, U0, V0)
        ctx.als_half_sweep(mfx.SIDE_USERS, reg)
        U1, _ = ctx.get_factors()
        ctx.als_half_sweep(mfx.SIDE_ITEMS, reg)
        _, V2 = ctx.get_factors()
    Uo, Vo = U0.copy(), V0.copy()
    orc.als_half(0, Uo, Vo, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, reg, nthreads=4)
    orc.als_half(1, Vo, U1, tr.ncols, cp, ci, cv, invI, reg, nthreads=4)
    relu = np.linalg.norm(U1 - Uo, axis=1) / np.maximum(np.linalg.norm(Uo, axis=1), 1e-6)
    reli = np.linalg.norm(V2[:tr.ncols] - Vo[:tr.ncols], axis=1) / np.maximum(np.linalg.norm(Vo[:tr.ncols], axis=1), 1e-6)
    # unpivoted LDL^T vs Eigen's pivoted one: both backward stable, they agree to eps * cond(A); the item systems here sum up to
    # 2000 outer products against reg = 2 (cond ~ 1e3..1e4): 5e-4 relative per row
    assert relu.max() < 5e-4 and reli.max() < 5e-4, (relu.max(), reli.max())


def test_wide_als_in_several_batches_equals_one_batch(monkeypatch):
    """als_wide.hip caps the segment partials (default 8 GB) and sweeps the rows in batches; a tiny cap forces many."""
    K, reg = 128, 1.5
    d, tr, _ = _data(900, 300, 40000, seed=21)
    nU, nI = d["nUsers"], d["nItems"]
    rng = np.random.default_rng(2)
    U0 = rng.normal(0, 0.3, (nU, K)).astype(np.float32)
    V0 = rng.normal(0, 0.3, (nI, K)).astype(np.float32)
    res = []
    for cap in (None, "0.002"):                   # 2 MB: about 40 segments of 49 KB per batch
        if cap:
            monkeypatch.setenv("MFX_ALS_SLAB_GB", cap)
        with Ctx(0) as ctx:
            load_ctx(ctx, d, K, U0, V0)
            ctx.als_half_sweep(mfx.SIDE_USERS, reg)
            ctx.als_half_sweep(mfx.SIDE_ITEMS, reg)
            res.append(ctx.get_factors())
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
