"""Edge cases through the C ABI: users without ratings, items that only occur in val/test (beyond the
train matrix' columns), a single rating, K not a multiple of 4, bad arguments.  Oracle = checker."""
import numpy as np
import pytest

from matfac_amd import Ctx, mfx, synth
from oracle import binding as orc

pytestmark = pytest.mark.gpu


def ragged():
    """6 users (users 1 and 4 have no train rating), train uses items 0..3, val mentions item 5, test item 6."""
    tr = synth.CSR(6, 4, np.array([0, 2, 2, 3, 5, 5, 7]), np.array([0, 3, 1, 0, 2, 1, 3], np.int32),
                   np.array([4, 2.5, 3, 5, 1, 2, 4.5], np.float32))
    va = synth.CSR(6, 6, np.array([0, 1, 2, 3, 3, 4, 5]), np.array([1, 0, 5, 2, 3], np.int32),
                   np.array([3, 4, 2, 1.5, 5], np.float32))
    te = synth.CSR(6, 7, np.array([0, 1, 1, 2, 3, 3, 4]), np.array([2, 6, 3, 0], np.int32),
                   np.array([2, 3, 4, 1], np.float32))
    return dict(train=tr, val=va, test=te, nUsers=6, nItems=7)


def test_invalid_sets_and_masked_evaluation():
    d = ragged()
    tr, va, te = d["train"], d["val"], d["test"]
    K = 5
    rng = np.random.default_rng(0)
    U = rng.normal(0, 0.5, (6, K)).astype(np.float32)
    V = rng.normal(0, 0.5, (7, K)).astype(np.float32)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, 6, 4, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_csr(mfx.MAT_VAL, 6, 6, va.rowptr, va.rowind, va.rowval)
        ctx.set_csr(mfx.MAT_TEST, 6, 7, te.rowptr, te.rowind, te.rowval)
        ctx.set_model(6, 7, K)
        ctx.set_factors(U, V)
        invU, invI = ctx.compute_invalid()
        ev, et = ctx.eval(mfx.MAT_VAL), ctx.eval(mfx.MAT_TEST)
        obj = ctx.objective(0.1, 0.2)
    oU, oI = orc.invalid(6, 4, tr.rowptr, tr.rowind, 6, 7)
    assert invU.tolist() == oU.tolist() == [0, 1, 0, 0, 1, 0]
    assert invI.tolist() == oI.tolist() == [0, 0, 0, 0, 1, 1, 1]          # items 4..6 never rated in train
    rv, sv, nv = orc.rmse(U, V, 6, 7, 6, va.rowptr, va.rowind, va.rowval, oU, oI, orc.DOT_TREE)
    rt, stt, nt = orc.rmse(U, V, 6, 7, 6, te.rowptr, te.rowind, te.rowval, oU, oI, orc.DOT_TREE)
    assert (ev.n, et.n) == (nv, nt) == (2, 3)        # val: users 1,4 invalid, item 5 invalid; test: item 6 invalid
    assert abs(ev.sse - sv) <= 1e-12 * sv and abs(et.sse - stt) <= 1e-12 * stt
    oobj, *_ = orc.objective(U, V, 6, 7, 6, tr.rowptr, tr.rowind, tr.rowval, oU, oI, 0.1, 0.2, orc.DOT_TREE)
    assert abs(obj - oobj) <= 1e-12 * oobj


@pytest.mark.parametrize("K", [1, 3, 5, 17, 64])
def test_all_trainers_leave_invalid_rows_alone_and_match_oracle(K):
    d = ragged()
    tr = d["train"]
    cp, ci, cv = orc.create_col_index(6, 4, tr.rowptr, tr.rowind, tr.rowval)
    rng = np.random.default_rng(K)
    U0 = rng.normal(0, 0.5, (6, K)).astype(np.float32)
    V0 = rng.normal(0, 0.5, (7, K)).astype(np.float32)
    oU, oI = orc.invalid(6, 4, tr.rowptr, tr.rowind, 6, 7)
    ru = tr.rowids()
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, 6, 4, tr.rowptr, tr.rowind, tr.rowval, cp, ci, cv)
        ctx.set_model(6, 7, K)
        # serial SGD in CSR order
        ctx.set_factors(U0, V0)
        ctx.sgd_epoch(0.05, 0.1, 0.1, mode=mfx.SGD_SERIAL, order=mfx.ORDER_NATURAL, arith=mfx.ARITH_REF64)
        U, V = ctx.get_factors()
        Uo, Vo = U0.copy(), V0.copy()
        orc.sgd_pass(Uo, Vo, ru, tr.rowind, tr.rowval, None, 0.05, 0.1, 0.1, orc.ARITH_REF64, orc.DOT_TREE)
        assert np.array_equal(U, Uo) and np.array_equal(V, Vo)
        # tiled SGD: 7 ratings, all of them visited once; invalid rows untouched
        ctx.set_factors(U0, V0)
        ctx.sgd_epoch(0.05, 0.1, 0.1, mode=mfx.SGD_TILED, seed=1, epoch=0)
        U, V = ctx.get_factors()
        u, i, r = ctx.debug_epoch_list()
        assert sorted(zip(u.tolist(), i.tolist())) == sorted(zip(ru.tolist(), tr.rowind.tolist()))
        assert np.array_equal(U[[1, 4]], U0[[1, 4]]) and np.array_equal(V[4:], V0[4:])
        assert not np.array_equal(U[0], U0[0])
        # ALS
        if K <= 256:
            ctx.set_factors(U0, V0)
            ctx.als_half_sweep(mfx.SIDE_USERS, 0.7)
            ctx.als_half_sweep(mfx.SIDE_ITEMS, 0.7)
            U, V = ctx.get_factors()
            Uo, Vo = U0.copy(), V0.copy()
            orc.als_half(0, Uo, Vo, 6, tr.rowptr, tr.rowind, tr.rowval, oU, 0.7)
            orc.als_half(1, Vo, Uo, 4, cp, ci, cv, oI, 0.7)
            assert np.allclose(U, Uo, rtol=2e-4, atol=2e-5) and np.allclose(V, Vo, rtol=2e-4, atol=2e-5)
            assert np.array_equal(U[[1, 4]], U0[[1, 4]]) and np.array_equal(V[4:], V0[4:])
        # CCD++
        ctx.set_factors(U0, V0)
        ctx.compute_invalid()
        ctx.ccdpp_begin()
        Uo, Vo = U0.copy(), V0.copy()
        Uo[:] = 0
        rr, rc = tr.rowval.copy(), cv.copy()
        for k in range(K):
            ctx.ccdpp_rank1(k, 0.3, 0.3, add_back=False)
            orc.ccdpp_rank1(k, Uo, Vo, 6, 7, 4, tr.rowptr, tr.rowind, rr, cp, ci, rc, oU, oI, 0.3, 0.3, False)
        grr, grc = ctx.debug_residuals(tr.nnz)
        U, V = ctx.get_factors()
        ctx.ccdpp_end()
        assert np.allclose(U, Uo, rtol=1e-6, atol=1e-7) and np.allclose(V, Vo, rtol=1e-6, atol=1e-7)
        assert np.allclose(grr, rr, atol=1e-6) and np.allclose(grc, rc, atol=1e-6)
        assert np.all(U[[1, 4]] == 0) and np.array_equal(V[4:], V0[4:])      # uFac.fill(0); invalid items keep iFac


def test_filtered_evaluation_matches_oracle_masks():
    """Model::RMSE(mat, filtItems, ...) / RMSEU (model.cpp:348-394, 446-486) = the masked evaluation with (invalid OR not kept)."""
    d = synth.make(dict(nU=500, nI=200, nnz=20000, K=8), seed=3)
    tr, te = d["train"], d["test"]
    nU, nI = d["nUsers"], max(d["nItems"], tr.ncols)
    rng = np.random.default_rng(1)
    U = rng.normal(0, 0.6, (nU, 8)).astype(np.float32)
    V = rng.normal(0, 0.6, (nI, 8)).astype(np.float32)
    ku = (rng.random(nU) < 0.3).astype(np.uint8)
    ki = (rng.random(nI) < 0.6).astype(np.uint8)
    oU, oI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_csr(mfx.MAT_TEST, te.nrows, nI, te.rowptr, te.rowind, te.rowval)
        ctx.set_model(nU, nI, 8)
        ctx.set_factors(U, V)
        ctx.compute_invalid()
        full = ctx.eval(mfx.MAT_TEST)
        for keep_u, keep_i in ((ku, None), (None, ki), (ku, ki)):
            e = ctx.eval_filtered(mfx.MAT_TEST, keep_u, keep_i)
            mu = oU | (1 - keep_u) if keep_u is not None else oU
            mi = oI | (1 - keep_i) if keep_i is not None else oI
            _, sse, n = orc.rmse(U, V, nU, nI, te.nrows, te.rowptr, te.rowind, te.rowval, mu.astype(np.uint8), mi.astype(np.uint8), orc.DOT_TREE)
            assert e.n == n and 0 < n < full.n and abs(e.sse - sse) <= 1e-12 * sse
        again = ctx.eval(mfx.MAT_TEST)                      # the filter does not stick
        assert (again.n, again.sse) == (full.n, full.sse)
        # the fused pair of isTerminateModel (objective on train + RMSE on another matrix) = the two single calls, bit for bit
        for K2 in (8, 64, 200):
            U2 = rng.normal(0, 0.3, (nU, K2)).astype(np.float32)
            V2 = rng.normal(0, 0.3, (nI, K2)).astype(np.float32)
            ctx.set_model(nU, nI, K2)
            ctx.set_factors(U2, V2)
            ctx.compute_invalid()
            a1, b1 = ctx.eval(mfx.MAT_TRAIN, with_norms=True), ctx.eval(mfx.MAT_TEST)
            a2, b2 = ctx.eval2(mfx.MAT_TRAIN, True, mfx.MAT_TEST, False)
            assert (a1.sse, a1.n, a1.unorm2, a1.inorm2) == (a2.sse, a2.n, a2.unorm2, a2.inorm2)
            assert (b1.sse, b1.n) == (b2.sse, b2.n)
            a3 = ctx.eval(mfx.MAT_TRAIN, with_norms=True)   # and the single call after the pair still reads its own region
            assert (a3.sse, a3.unorm2) == (a1.sse, a1.unorm2)


def test_single_rating_and_argument_errors():
    tr = synth.CSR(1, 1, np.array([0, 1]), np.array([0], np.int32), np.array([3.0], np.float32))
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, 1, 1, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_model(1, 1, 2)
        ctx.set_factors(np.array([[0.5, 0.5]], np.float32), np.array([[1.0, 1.0]], np.float32))
        ctx.compute_invalid()
        for mode in (mfx.SGD_HOGWILD, mfx.SGD_TILED, mfx.SGD_SERIAL):
            ctx.sgd_epoch(0.01, 0.0, 0.0, mode=mode, order=mfx.ORDER_DEVICE if mode != mfx.SGD_SERIAL else mfx.ORDER_NATURAL)
        assert ctx.eval(mfx.MAT_TRAIN).n == 1
        with pytest.raises(mfx.MfxError) as e:
            ctx.eval(mfx.MAT_VAL)                              # never uploaded
        assert e.value.code == -5
        with pytest.raises(mfx.MfxError):
            ctx.sgd_epoch(0.01, 0, 0, mode=7)
        with pytest.raises(mfx.MfxError):
            ctx.ccdpp_rank1(0, 0.1, 0.1, False)                # without ccdpp_begin
        with pytest.raises(mfx.MfxError):
            ctx.set_csr(mfx.MAT_VAL, 1, 1, np.array([0, 1]), np.array([3], np.int32), np.array([1.0], np.float32))  # column out of range
        with pytest.raises(mfx.MfxError):
            ctx.set_model(0, 5, 4)
    with Ctx(0) as ctx:
        ctx.set_model(4, 4, 300)
        ctx.set_csr(mfx.MAT_TRAIN, 4, 4, np.arange(5), np.arange(4, dtype=np.int32), np.ones(4, np.float32))
        with pytest.raises(mfx.MfxError) as e:
            ctx.als_half_sweep(mfx.SIDE_USERS, 1.0)            # ALS is built for K <= 256
        assert "K <= 256" in str(e.value)


def test_contexts_give_their_device_memory_back():
    """Every trainer run once in a context, then mfx_destroy: free device memory returns to where it was
    (hipMemGetInfo through the HIP runtime libmfx.so is linked against)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()

    def free_now():
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    d = synth.make(dict(nU=3000, nI=800, nnz=120000, K=16), seed=5)
    tr, va = d["train"], d["val"]
    nU, nI = d["nUsers"], max(d["nItems"], tr.ncols)
    U0, V0 = synth.init_factors(1, nU, nI, 16)

    def once(K):
        with Ctx(0) as ctx:
            ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
            ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
            ctx.set_model(nU, nI, K)
            ctx.compute_invalid()
            for mode in (mfx.SGD_TILED, mfx.SGD_HOGWILD):
                ctx.sgd_epoch(0.002, 0.01, 0.01, mode=mode, seed=1, epoch=0)
            ctx.als_half_sweep(mfx.SIDE_USERS, 1.0)
            ctx.als_half_sweep(mfx.SIDE_ITEMS, 1.0)
            ctx.ccdpp_begin()
            ctx.ccdpp_rank1(0, 0.5, 0.5, add_back=False)
            ctx.ccdpp_end()
            ctx.ccd_begin()
            ctx.ccd_sweep(mfx.SIDE_USERS, 0.5)
            ctx.ccd_end()
            sig = ctx.svd_init(2, 4, 1)
            ctx.sgd_set_dim_reg(1.0 / (1.0 + sig))
            ctx.sgd_epoch(0.002, 0, 0, mode=mfx.SGD_HOGWILD)
            ctx.sgd_set_dim_reg(None)
            f = np.ones(nU, np.float32)
            ctx.set_tmf(f, np.full(nU, K, np.int32), np.ones(nI, np.float32), np.full(nI, max(1, K // 2), np.int32))
            ctx.sgd_epoch(0.002, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=1)
            ctx.rmse(mfx.MAT_VAL)

    once(16)                       # warm up allocator pools, code objects
    once(128)
    series = [free_now()]
    for rep in range(4):
        for K in (16, 128, 16):
            once(K)
        series.append(free_now())
    print("free device memory after each batch of 3 contexts:", series)
    # the runtime grows its own pools in 16 MB steps now and then; a leak would take memory with EVERY batch
    drops = [a - b for a, b in zip(series, series[1:])]
    assert sum(1 for x in drops if x > 0) <= 2 and series[0] - series[-1] <= 64 << 20, series
