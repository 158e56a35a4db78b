"""GPU parity of the CCD++ rank-one sweeps against the oracle's restatement of
modelMF.cpp:1013-1121 (and the FreqAdap variant :1258-1360).

Every term is the reference's (float products, double accumulation, one rounding to float); only
the association of the double sums differs (16-lane groups / segments vs the sequential CSR walk),
which changes a float result by at most 1 ulp and almost never at all.  Tolerance: <= 2 ulp on the
updated factor columns, 1e-6 absolute on the residuals, 1e-6 on RMSE."""
import numpy as np
import pytest

from matfac_amd import Ctx, mfx, synth
from oracle import binding as orc
from tests.util import load_ctx

pytestmark = pytest.mark.gpu


def ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai)
    bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    return np.abs(ai - bi)


def _setup(nU, nI, nnz, K, seed):
    d = synth.make(dict(nU=nU, nI=nI, nnz=nnz, K=K), seed=seed)
    tr = d["train"]
    cp, ci, cv = orc.create_col_index(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
    U0, V0 = orc.init_factors(1, d["nUsers"], d["nItems"], K)
    return d, tr, (cp, ci, cv), U0, V0


@pytest.mark.parametrize("K,freq", [(8, -1.0), (64, -1.0), (16, 75.0), (128, -1.0)])
def test_rank1_steps_match_oracle(K, freq):
    d, tr, (cp, ci, cv), U0, V0 = _setup(1500, 400, 60000, K, seed=K)
    nU, nI = d["nUsers"], d["nItems"]
    uReg, iReg = 0.3, 0.2
    Uo, Vo = U0.copy(), V0.copy()
    Uo[:] = 0
    rr, rc = tr.rowval.copy(), cv.copy()
    assert np.bincount(tr.rowind).max() > 1024           # a split column
    with Ctx(0) as ctx:
        invU, invI = load_ctx(ctx, d, K, U0, V0)
        ctx.ccdpp_begin()
        for it in range(2):
            for k in range(min(K, 6)):
                ctx.ccdpp_rank1(k, uReg, iReg, add_back=it > 0, inner=5, freq_thresh=freq)
                orc.ccdpp_rank1(k, Uo, Vo, nU, nI, tr.ncols, tr.rowptr, tr.rowind, rr, cp, ci, rc, invU, invI,
                                uReg, iReg, it > 0, 5, freq, nthreads=4)
                U, V = ctx.get_factors()
                assert ulp_diff(U[:, k], Uo[:, k]).max() <= 2, (it, k)
                assert ulp_diff(V[:, k], Vo[:, k]).max() <= 2, (it, k)
        grr, grc = ctx.debug_residuals(tr.nnz)
        U, V = ctx.get_factors()
        ctx.ccdpp_end()
    assert np.abs(grr - rr).max() < 1e-5 and np.abs(grc - rc).max() < 1e-5
    # the two residual views stay in lock-step (same multiset of values per rating)
    order = np.argsort(tr.rowind, kind="stable")
    assert np.array_equal(grr[order], grc)
    # columns never touched keep: U = 0 (uFac.fill(0)), V = init
    assert np.all(U[:, min(K, 6):] == 0) and np.array_equal(V[:, min(K, 6):], V0[:, min(K, 6):])
    if freq >= 0:
        cold = np.diff(cp) < freq
        assert np.all(V[:tr.ncols][cold, 1:min(K, 6)] == 0) and np.any(V[:tr.ncols][cold, 0] != 0)


@pytest.mark.parametrize("fuse", ["", "0"])
def test_ccdpp_outer_iterations_track_oracle_and_objective_decreases(fuse, monkeypatch):
    """Four outer iterations with the factors in shuffled order.  From the second factor on the deferred subtract and the add-back
    ride on the first sweep of each factor (the fused pass kernels, default) or run as their own sweep (MFX_CCD_FUSE=0): the same
    models either way -- both keep every rounding of modelMF.cpp:1032-1056 and :1095-1116."""
    if fuse:
        monkeypatch.setenv("MFX_CCD_FUSE", fuse)
    K, reg = 16, 0.5
    d, tr, (cp, ci, cv), U0, V0 = _setup(1000, 600, 40000, K, seed=7)
    va = d["val"]
    nU, nI = d["nUsers"], d["nItems"]
    Uo, Vo = U0.copy(), V0.copy()
    Uo[:] = 0
    rr, rc = tr.rowval.copy(), cv.copy()
    mt = orc.MT(1)
    dims = np.arange(K, dtype=np.int32)
    objs = []
    with Ctx(0) as ctx:
        invU, invI = load_ctx(ctx, d, K, U0, V0)
        ctx.ccdpp_begin()
        for it in range(4):
            mt.shuffle_i32(dims)                       # modelMF.cpp:1026
            for k in dims:
                ctx.ccdpp_rank1(int(k), reg, reg, add_back=it > 0)
                orc.ccdpp_rank1(int(k), Uo, Vo, nU, nI, tr.ncols, tr.rowptr, tr.rowind, rr, cp, ci, rc, invU, invI,
                                reg, reg, it > 0, 5, -1.0, nthreads=4)
            g_obj, g_val = ctx.objective(reg, reg), ctx.rmse(mfx.MAT_VAL)
            o_obj, *_ = orc.objective(Uo, Vo, nU, nI, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, invI, reg, reg)
            o_val, _, _ = orc.rmse(Uo, Vo, nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, invU, invI)
            assert abs(g_obj - o_obj) <= 1e-5 * o_obj and abs(g_val - o_val) <= 1e-6
            objs.append(g_obj)
        U, V = ctx.get_factors()
        grr, grc = ctx.debug_residuals(tr.nnz)
        ctx.ccdpp_end()
    assert ulp_diff(U, Uo).max() <= 2 and ulp_diff(V, Vo).max() <= 2
    assert np.abs(grr - rr).max() < 1e-5 and np.abs(grc - rc).max() < 1e-5
    assert all(b <= a * (1 + 1e-6) for a, b in zip(objs, objs[1:]))   # CCD++ never increases the objective


@pytest.mark.parametrize("light", ["1024", "0", "1000000"])
def test_column_view_strips_and_light_columns_agree(light, monkeypatch):
    """The column view keeps columns of at most MFX_CCD_LIGHT (default 1024) entries whole behind the user strips and
    cuts the others into 8192-user strips: 20 000 users = 3 strips, 300 items of which about half are light.  All three
    layouts (mixed, strips only, whole columns only) give the oracle's columns within 2 ulp and bit-identical views."""
    monkeypatch.setenv("MFX_CCD_LIGHT", light)
    K = 8
    d, tr, (cp, ci, cv), U0, V0 = _setup(20000, 300, 500000, K, seed=31)
    nU, nI = d["nUsers"], d["nItems"]
    deg = np.bincount(tr.rowind, minlength=tr.ncols)
    assert (deg > 1024).sum() > 20 and ((deg > 0) & (deg <= 1024)).sum() > 20
    uReg, iReg = 0.3, 0.2
    Uo, Vo = U0.copy(), V0.copy()
    Uo[:] = 0
    rr, rc = tr.rowval.copy(), cv.copy()
    with Ctx(0) as ctx:
        invU, invI = load_ctx(ctx, d, K, U0, V0)
        ctx.ccdpp_begin()
        for it in range(2):
            for k in range(3):
                ctx.ccdpp_rank1(k, uReg, iReg, add_back=it > 0, inner=5)
                orc.ccdpp_rank1(k, Uo, Vo, nU, nI, tr.ncols, tr.rowptr, tr.rowind, rr, cp, ci, rc, invU, invI, uReg, iReg, it > 0, 5,
                                -1.0, nthreads=4)
                U, V = ctx.get_factors()
                assert ulp_diff(U[:, k], Uo[:, k]).max() <= 2 and ulp_diff(V[:, k], Vo[:, k]).max() <= 2, (it, k)
        grr, grc = ctx.debug_residuals(tr.nnz)
        ctx.ccdpp_end()
    assert np.abs(grr - rr).max() < 1e-5 and np.abs(grc - rc).max() < 1e-5
    order = np.argsort(tr.rowind, kind="stable")
    assert np.array_equal(grr[order], grc)



@pytest.mark.parametrize("contig", ["", "1"])
@pytest.mark.parametrize("nI", [500, 70000])
def test_wide_item_axis_and_both_window_layouts(nI, contig, monkeypatch):
    """The pass kernels have two index widths and the block plan two ways of dealing the trips.  70 000 items: v_k does not fit in
    LDS (gathered from L2), item and column ids are 32 bits; 500 items: 16-bit ids, v_k in LDS.  MFX_CCD_CONTIG=1 gives every
    workgroup one contiguous window, what regions of 2^30 entries and more use (the loop addresses a region with 32-bit byte
    offsets), instead of chunks dealt round-robin.  Same tolerances as above on all four."""
    if contig:
        monkeypatch.setenv("MFX_CCD_CONTIG", contig)
    K = 8
    d, tr, (cp, ci, cv), U0, V0 = _setup(12000, nI, 200000, K, seed=5 + nI)
    nU, nIt = d["nUsers"], d["nItems"]
    uReg, iReg = 0.3, 0.2
    Uo, Vo = U0.copy(), V0.copy()
    Uo[:] = 0
    rr, rc = tr.rowval.copy(), cv.copy()
    with Ctx(0) as ctx:
        invU, invI = load_ctx(ctx, d, K, U0, V0)
        ctx.ccdpp_begin()
        for it in range(2):
            for k in range(3):
                ctx.ccdpp_rank1(k, uReg, iReg, add_back=it > 0, inner=5)
                orc.ccdpp_rank1(k, Uo, Vo, nU, nIt, tr.ncols, tr.rowptr, tr.rowind, rr, cp, ci, rc, invU, invI, uReg, iReg, it > 0, 5,
                                -1.0, nthreads=4)
                U, V = ctx.get_factors()
                assert ulp_diff(U[:, k], Uo[:, k]).max() <= 2 and ulp_diff(V[:, k], Vo[:, k]).max() <= 2, (it, k)
        grr, grc = ctx.debug_residuals(tr.nnz)
        ctx.ccdpp_end()
    assert np.abs(grr - rr).max() < 1e-5 and np.abs(grc - rc).max() < 1e-5
    order = np.argsort(tr.rowind, kind="stable")
    assert np.array_equal(grr[order], grc)


def test_random_small_shapes_against_the_oracle(monkeypatch):
    """The block plan, the padded views, the record batches and the fused first sweeps on shapes that stress their edges: fewer
    ratings than one trip (most groups of the one workgroup have no trip at all), rows and columns without ratings, one-entry
    pieces next to 1024-entry ones.  Twelve random shapes; three rank-one steps against the oracle at its tolerances, then three
    more that add back -- from their second factor on the residual update rides on the first sweep -- against the oracle (errors of
    1 ulp propagate through the later factors of a 70-rating matrix: relative 2e-6) and BIT FOR BIT against the same steps with the
    update as a sweep of its own (MFX_CCD_FUSE=0)."""
    rng = np.random.default_rng(2024)
    shapes = [(3, 2, 4), (70, 9, 65), (200, 300, 700), (5000, 40, 30000), (40, 5000, 30000), (9000, 33, 9000)]
    shapes += [(int(rng.integers(50, 3000)), int(rng.integers(5, 900)), int(rng.integers(100, 40000))) for _ in range(6)]
    for nU, nI, nnz in shapes:
        nnz = min(nnz, nU * nI // 2 + 1)
        K = 4
        d, tr, (cp, ci, cv), U0, V0 = _setup(nU, nI, nnz, K, seed=nU + nI)
        nUs, nIs = d["nUsers"], d["nItems"]
        shape = (nU, nI, tr.nnz)
        Uo, Vo = U0.copy(), V0.copy()
        Uo[:] = 0
        rr, rc = tr.rowval.copy(), cv.copy()
        got = {}
        for fuse in ("1", "0"):
            monkeypatch.setenv("MFX_CCD_FUSE", fuse)
            with Ctx(0) as ctx:
                invU, invI = load_ctx(ctx, d, K, U0, V0)
                ctx.ccdpp_begin()
                for it in range(2):
                    for k in range(3):
                        ctx.ccdpp_rank1(k, 0.3, 0.2, add_back=it > 0, inner=3)
                        if fuse == "1":
                            orc.ccdpp_rank1(k, Uo, Vo, nUs, nIs, tr.ncols, tr.rowptr, tr.rowind, rr, cp, ci, rc, invU, invI, 0.3, 0.2, it > 0, 3,
                                            -1.0, nthreads=2)
                    if it == 0 and fuse == "1":
                        U, V = ctx.get_factors()
                        assert ulp_diff(U[:, :3], Uo[:, :3]).max() <= 2 and ulp_diff(V[:, :3], Vo[:, :3]).max() <= 2, shape
                U, V = ctx.get_factors()
                grr, grc = ctx.debug_residuals(tr.nnz)
                ctx.ccdpp_end()
            got[fuse] = (U, V, grr, grc)
        U, V, grr, grc = got["1"]
        assert np.allclose(U[:, :3], Uo[:, :3], rtol=2e-6, atol=1e-7) and np.allclose(V[:, :3], Vo[:, :3], rtol=2e-6, atol=1e-7), shape
        assert np.abs(grr - rr).max() < 1e-5 and np.abs(grc - rc).max() < 1e-5, shape
        for x, y in zip(got["1"], got["0"]):
            assert np.array_equal(x, y), shape
