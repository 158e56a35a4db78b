"""Pins the CPU oracle (oracle/oracle.cpp): the known-answer tests of SURVEY.md 8a/8c -- the only
reference-derived vectors that exist (the reference ships no tests or fixtures) -- plus algebraic
invariants of each restated loop."""
import os

import numpy as np
import pytest

from matfac_amd import synth
from oracle import binding as orc


# ---- libstdc++ facts recorded from the reference's own RNG call sites ------------
def test_init_stream_matches_recorded_vector():
    # model.cpp:2331-2341 with seed=1: uFac(0,0..7)
    U, V = orc.init_factors(1, 4, 3, 8)
    exp = np.array([-0.00736924401, -0.000826997333, -0.00562081626, 0.00357729429, 0.00869385805,
                    0.000388327433, -0.00930855796, 0.000594003825], np.float32)
    assert np.array_equal(U[0], exp)
    assert np.all(np.abs(U) <= 0.01) and np.all(np.abs(V) <= 0.01)
    # fill order: uFac row by row, then iFac (one stream)
    U2, V2 = orc.init_factors(1, 4, 3, 4)
    allv = np.concatenate([U.ravel(), V.ravel()])
    assert np.array_equal(np.concatenate([U2.ravel(), V2.ravel()]), allv[:28])


def test_mt19937_and_shuffle_vectors():
    mt = orc.MT(1)
    assert (mt.next(), mt.next()) == (1791095845, 4282876139)
    a = np.arange(10, dtype=np.uint64)
    orc.MT(1).shuffle_u64(a)
    assert a.tolist() == [9, 0, 2, 5, 7, 4, 6, 3, 1, 8]          # modelMF.cpp:78 (vector<size_t>)
    b = np.arange(8, dtype=np.int32)
    orc.MT(1).shuffle_i32(b)
    assert b.tolist() == [1, 0, 2, 5, 7, 4, 6, 3]                # modelMF.cpp:1026 (vector<int>)
    # parBlockShuffle with one thread is a plain std::shuffle (util.cpp:1047-1064)
    c = np.arange(10, dtype=np.uint64)
    orc.MT(1).par_block_shuffle_u64(c, 1)
    assert c.tolist() == a.tolist()
    # with T threads every block stays inside its own range
    c = np.arange(103, dtype=np.uint64)
    orc.MT(5).par_block_shuffle_u64(c, 4)
    bs = 103 // 4
    for t in range(4):
        lo, hi = t * bs, (t + 1) * bs if t < 3 else 103
        assert sorted(c[lo:hi].tolist()) == list(range(lo, hi))


def test_block_sequence_is_a_perfect_matching():
    r, c = orc.MT(3).block_seq(8)                                  # util.cpp:1077-1107
    assert sorted(r.tolist()) == list(range(8)) and sorted(c.tolist()) == list(range(8))


# ---- SURVEY.md 8a known answers -------------------------------------------------------------------
def test_sgd_known_answer():
    # a4: K=2, p=(0.1,0.2), q=(0.3,0.4), r=1, lr=0.01, uReg=iReg=0.1
    for arith in (orc.ARITH_REF64, orc.ARITH_REF64F, orc.ARITH_F32):
        for dm in (orc.DOT_SEQ, orc.DOT_TREE):
            U = np.array([[0.1, 0.2]], np.float32)
            V = np.array([[0.3, 0.4]], np.float32)
            orc.sgd_pass(U, V, np.zeros(1, np.int32), np.zeros(1, np.int32), np.ones(1, np.float32), None,
                         0.01, 0.1, 0.1, arith, dm)
            assert np.allclose(U, [[0.10514, 0.20672]], rtol=0, atol=2e-8)
            assert np.allclose(V, [[0.301271492, 0.402879616]], rtol=0, atol=5e-8)


def test_sgd_item_update_sees_updated_user_row():
    U = np.array([[0.5, -0.25]], np.float32)
    V = np.array([[0.125, 0.75]], np.float32)
    p, q = U[0].astype(np.float64), V[0].astype(np.float64)
    lr, ur, ir, r = 0.01, 0.1, 0.2, 2.0
    diff = r - float(np.float32(np.float32(p[0] * q[0]) + np.float32(p[1] * q[1])))
    pn = (p - np.float32(lr).astype(np.float64) * (-2.0 * diff * q + 2.0 * np.float64(np.float32(ur)) * p)).astype(np.float32)
    qn = (q - np.float32(lr).astype(np.float64) * (-2.0 * diff * pn.astype(np.float64) + 2.0 * np.float64(np.float32(ir)) * q)).astype(np.float32)
    orc.sgd_pass(U, V, np.zeros(1, np.int32), np.zeros(1, np.int32), np.array([r], np.float32), None, lr, ur, ir)
    assert np.array_equal(U[0], pn) and np.array_equal(V[0], qn)


def test_als_known_answer_and_ldlt():
    # a9: rows q1=(1,0), q2=(1,1), r=(2,3), uReg=0.5 => A=[[2.5,1],[1,1.5]], b=(5,3)
    x = orc.ldlt_solve(np.array([[2.5, 1], [1, 1.5]]), np.array([5, 3.0]))
    assert np.allclose(x, [4.5 / 2.75, 2.5 / 2.75], rtol=2e-7)
    V = np.array([[1, 0], [1, 1]], np.float32)
    U = np.zeros((1, 2), np.float32)
    orc.als_half(0, U, V, 1, np.array([0, 2], np.int64), np.array([0, 1], np.int32), np.array([2, 3], np.float32),
                 np.zeros(1, np.uint8), 0.5)
    assert np.allclose(U[0], [1.6363636, 0.9090909], rtol=2e-7)
    # pivoted LDLT on random SPD systems vs float64 solve; pivoting must kick in (largest diagonal last)
    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 17, 64):
        Q = rng.normal(size=(3 * n, n))
        A = Q.T @ Q + 0.1 * np.eye(n)
        A[n - 1, n - 1] += 50.0
        b = rng.normal(size=n)
        x = orc.ldlt_solve(A, b)
        ref = np.linalg.solve(A, b)
        assert np.linalg.norm(x - ref) / np.linalg.norm(ref) < 2e-4


def test_ccdpp_known_answer():
    # a10: row residuals (1,2,3), v_k=(0.5,-1,2), uReg=0.1 => num=4.5, denom=5.35, u=0.8411215
    nU, nI, K = 1, 3, 1
    rowptr = np.array([0, 3], np.int64)
    rowind = np.array([0, 1, 2], np.int32)
    vals = np.array([1, 2, 3], np.float32)
    cp, ci, cv = orc.create_col_index(1, 3, rowptr, rowind, vals)
    U = np.zeros((1, 1), np.float32)
    V = np.array([[0.5], [-1], [2]], np.float32)
    rr, rc = vals.copy(), cv.copy()
    # inner=1 and a huge iReg freezes v (numerator/denominator ~ 0): check u after the row pass
    orc.ccdpp_rank1(0, U, V, nU, nI, 3, rowptr, rowind, rr, cp, ci, rc, np.zeros(1, np.uint8), np.zeros(3, np.uint8),
                    0.1, 1e30, False, inner=1)
    assert abs(U[0, 0] - 0.8411215) < 1e-6


# ---- invariants of the restated loops ----------------------------------------------------------------
def _small(seed=3, nU=120, nI=90, nnz=2500):
    d = synth.make(dict(nU=nU, nI=nI, nnz=nnz, K=0), seed=seed)
    tr = d["train"]
    cp, ci, cv = orc.create_col_index(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
    return d, tr, (cp, ci, cv)


def test_col_index_is_stable_transpose():
    d, tr, (cp, ci, cv) = _small()
    cp2, ci2, cv2 = tr.col_view()
    assert np.array_equal(cp, cp2) and np.array_equal(ci, ci2) and np.array_equal(cv, cv2)
    for c in range(tr.ncols):
        assert np.all(np.diff(ci[cp[c]:cp[c + 1]]) > 0)           # users ascending inside a column


def test_invalid_sets():
    rowptr = np.array([0, 2, 2, 3], np.int64)
    rowind = np.array([0, 2, 2], np.int32)
    invU, invI = orc.invalid(3, 3, rowptr, rowind, 4, 5)
    assert invU.tolist() == [0, 1, 0, 1]                           # user 1 empty, user 3 beyond the matrix
    assert invI.tolist() == [0, 1, 0, 1, 1]                        # item 1 unrated, items 3,4 beyond ncols


def test_objective_and_rmse_against_numpy():
    d, tr, _ = _small()
    va = d["val"]
    nU, nI, K = d["nUsers"], d["nItems"], 7
    rng = np.random.default_rng(1)
    U = rng.normal(0, 0.5, (nU, K)).astype(np.float32)
    V = rng.normal(0, 0.5, (nI, K)).astype(np.float32)
    invU, invI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
    obj, sse, un, inn = orc.objective(U, V, nU, nI, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, invI, 0.3, 0.7)
    ru = tr.rowids()
    est = np.einsum("ij,ij->i", U[ru].astype(np.float64), V[tr.rowind].astype(np.float64))
    sse_np = ((tr.rowval - est) ** 2).sum()
    vu, vi = ~invU.astype(bool), ~invI.astype(bool)
    un_np = (U[vu].astype(np.float64) ** 2).sum()
    in_np = (V[vi].astype(np.float64) ** 2).sum()
    assert abs(sse - sse_np) < 1e-5 * sse_np and abs(un - un_np) < 1e-5 * un_np and abs(inn - in_np) < 1e-5 * in_np
    assert abs(obj - (sse + float(np.float32(0.3)) * un + float(np.float32(0.7)) * inn)) < 1e-12 * obj
    r, vsse, cnt = orc.rmse(U, V, nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, invU, invI)
    vru = va.rowids()
    keep = vu[vru] & (va.rowind < nI) & vi[np.minimum(va.rowind, nI - 1)]
    est = np.einsum("ij,ij->i", U[vru[keep]].astype(np.float64), V[va.rowind[keep]].astype(np.float64))
    assert cnt == keep.sum() and abs(r - np.sqrt(((va.rowval[keep] - est) ** 2).mean())) < 1e-6
    # the two dot orders agree to fp32 round-off
    r2, _, _ = orc.rmse(U, V, nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, invU, invI, orc.DOT_TREE)
    assert abs(r - r2) < 1e-6


@pytest.mark.parametrize("K", [3, 16, 17, 33, 64, 65, 130])
def test_tree_dot_matches_sequential_dot_to_roundoff(K):
    rng = np.random.default_rng(K)
    a = rng.normal(size=K).astype(np.float32)
    b = rng.normal(size=K).astype(np.float32)
    ref = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
    bound = 4e-7 * float(np.sum(np.abs(a * b))) + 1e-12
    assert abs(orc.dot(a, b, orc.DOT_SEQ) - ref) < bound * K ** 0.5
    assert abs(orc.dot(a, b, orc.DOT_TREE) - ref) < bound * K ** 0.5
    L, C = orc.tree_shape(K)
    assert 4 * L * C >= K and (L, C) == ((4, 1) if K <= 16 else (8, 1) if K <= 32 else (16, (K + 63) // 64))


def test_als_rows_solve_their_normal_equations():
    d, tr, (cp, ci, cv) = _small()
    nU, nI, K, reg = d["nUsers"], d["nItems"], 12, 0.8
    rng = np.random.default_rng(5)
    U = rng.normal(0, 0.4, (nU, K)).astype(np.float32)
    V = rng.normal(0, 0.4, (nI, K)).astype(np.float32)
    invU, invI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
    U1 = U.copy()
    orc.als_half(0, U1, V, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, reg, nthreads=2)
    U2 = U.copy()
    orc.als_half(0, U2, V, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, reg, nthreads=1)
    assert np.array_equal(U1, U2)                                  # thread-count independent
    for u in range(nU):
        if invU[u]:
            assert np.array_equal(U1[u], U[u])
            continue
        sl = slice(tr.rowptr[u], tr.rowptr[u + 1])
        Q = V[tr.rowind[sl]].astype(np.float64)
        A = Q.T @ Q + reg * np.eye(K)
        b = Q.T @ tr.rowval[sl].astype(np.float64)
        assert np.linalg.norm(A @ U1[u] - b) / np.linalg.norm(b) < 1e-5   # (A+lambda I)x = b to 1e-5


def test_ccdpp_residual_views_stay_equal_and_objective_decreases():
    d, tr, (cp, ci, cv) = _small()
    nU, nI, K, reg = d["nUsers"], d["nItems"], 6, 0.2
    U, V = orc.init_factors(1, nU, nI, K)
    U[:] = 0
    invU, invI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
    rr, rc = tr.rowval.copy(), cv.copy()
    order = np.argsort(tr.rowind, kind="stable")
    prev = None
    for it in range(3):
        for k in range(K):
            orc.ccdpp_rank1(k, U, V, nU, nI, tr.ncols, tr.rowptr, tr.rowind, rr, cp, ci, rc, invU, invI, reg, reg, it > 0)
            assert np.array_equal(rr[order], rc)
        obj, *_ = orc.objective(U, V, nU, nI, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, invI, reg, reg)
        assert prev is None or obj <= prev * (1 + 1e-6)
        prev = obj
    # the residual IS r - p.q
    ru = tr.rowids()
    est = np.einsum("ij,ij->i", U[ru].astype(np.float64), V[tr.rowind].astype(np.float64))
    assert np.abs(rr - (tr.rowval - est)).max() < 1e-4


def test_text_csr_and_factor_files_roundtrip(tmp_path):
    d, tr, _ = _small()
    p = str(tmp_path / "m.csr")
    orc.write_csr_text(p, tr.nrows, tr.rowptr, tr.rowind, tr.rowval)
    nr, nc, rp, ri, rv = orc.read_csr_text(p)
    assert (nr, nc) == (tr.nrows, tr.ncols)
    assert np.array_equal(rp, tr.rowptr) and np.array_equal(ri, tr.rowind) and np.array_equal(rv, tr.rowval)
    # an empty line is a user without ratings; '%' lines are comments
    open(p, "w").write("% comment\n0 1.5 3 2\n\n2 4.5\n")
    nr, nc, rp, ri, rv = orc.read_csr_text(p)
    assert (nr, nc) == (3, 4) and rp.tolist() == [0, 2, 2, 3] and ri.tolist() == [0, 3, 2] and rv.tolist() == [1.5, 2, 4.5]
    M = np.random.default_rng(0).normal(size=(5, 3)).astype(np.float32)
    q = str(tmp_path / "f.mat")
    orc.write_mat(q, M)
    assert open(q).readline().endswith(" \n")                       # io.cpp:139-154: "v " per value
    M2 = orc.read_mat(q, 5, 3)
    assert np.allclose(M2, M, rtol=1e-5)                             # default ostream precision: 6 digits


def test_train_loop_termination_rules():
    """isTerminateModel (model.cpp:1471-1540): best-val snapshot, EPS convergence, NaN rollback."""
    d, tr, (cp, ci, cv) = _small(nU=200, nI=150, nnz=5000)
    va = d["val"]
    nU, nI, K = d["nUsers"], d["nItems"], 8
    U0, V0 = orc.init_factors(1, nU, nI, K)
    tcsr = (tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval, cp, ci, cv)
    vcsr = (va.nrows, va.rowptr, va.rowind, va.rowval)
    r = orc.train(orc.M_ALS, U0, V0, tcsr, vcsr, nU, nI, K, 30, 1, 0.005, 2.0, 2.0)
    assert r["iters"] <= 30 and np.all(np.diff(r["obj"]) <= 1e-6 * r["obj"][:-1])     # ALS is monotone
    assert r["bestIter"] == int(np.argmin(r["val"]))
    invU, invI = r["invU"], r["invI"]
    best_val, _, _ = orc.rmse(r["Ubest"], r["Vbest"], nU, nI, *vcsr, invU, invI)
    assert abs(best_val - r["val"].min()) < 1e-12
    # SGD with an absurd learning rate hits the NaN guard: learnRate is halved from the best model's
    r = orc.train(orc.M_SGD, U0 * 50, V0 * 50, tcsr, vcsr, nU, nI, K, 6, 1, 5.0, 0.01, 0.01)
    assert r["learnRate"] < 5.0 and np.isnan(r["obj"]).any()
    # sequential SGD (a4), Hogwild with one thread (a5), user-shuffle (a7), stratified (a6) all learn
    for m in (orc.M_SGD, orc.M_HOGSGD, orc.M_SGDU, orc.M_SGDPAR, orc.M_CCDPP, orc.M_CCDPP_FA, orc.M_CCD):
        r = orc.train(m, U0, V0, tcsr, vcsr, nU, nI, K, 8, 1, 0.01, 0.05, 0.05, nthreads=2 if m == orc.M_SGDPAR else 1)
        assert np.isfinite(r["obj"]).all() and r["obj"][-1] < r["obj"][0], m
