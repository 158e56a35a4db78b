"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg (see oracle/oracle.h).  The product package matfac_amd never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

DOT_SEQ, DOT_TREE = 0, 1
ARITH_REF64, ARITH_REF64F, ARITH_F32 = 0, 1, 2
M_SGD, M_HOGSGD, M_SGDPAR, M_SGDU, M_ALS, M_CCDPP, M_CCDPP_FA, M_CCD = range(8)


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    if not os.path.exists(_SO):
        build()
    return C.CDLL(_SO)


lib = _load()

_f = C.POINTER(C.c_float)
_d = C.POINTER(C.c_double)
_i32 = C.POINTER(C.c_int32)
_i64 = C.POINTER(C.c_int64)
_u64 = C.POINTER(C.c_uint64)
_u8 = C.POINTER(C.c_uint8)


def p(a, t):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be contiguous"
    return a.ctypes.data_as(t)


def F(a):
    assert a.dtype == np.float32
    return p(a, _f)


def I32(a):
    assert a.dtype == np.int32
    return p(a, _i32)


def I64(a):
    assert a.dtype == np.int64
    return p(a, _i64)


def U64(a):
    assert a.dtype == np.uint64
    return p(a, _u64)


def U8(a):
    assert a.dtype == np.uint8
    return p(a, _u8)


lib.orc_dot.restype = C.c_float
lib.orc_mt_create.restype = C.c_void_p
lib.orc_mt_next.restype = C.c_uint32
lib.orc_strat_create.restype = C.c_void_p
lib.orc_objective.restype = C.c_double
lib.orc_rmse.restype = C.c_double
lib.orc_time_hogwild.restype = C.c_double
lib.orc_time_strat.restype = C.c_double
lib.orc_rand_pairs.restype = C.c_int64
lib.orc_bias_eval.restype = C.c_double


class TrainCfg(C.Structure):
    _fields_ = [
        ("method", C.c_int32), ("K", C.c_int32), ("maxIter", C.c_int32), ("seed", C.c_int32),
        ("nthreads", C.c_int32), ("dot_mode", C.c_int32),
        ("uReg", C.c_float), ("iReg", C.c_float), ("learnRate", C.c_float),
        ("nUsers", C.c_int32), ("nItems", C.c_int32),
        ("tr_nrows", C.c_int32), ("tr_ncols", C.c_int32),
        ("tr_rowptr", _i64), ("tr_rowind", _i32), ("tr_rowval", _f),
        ("tr_colptr", _i64), ("tr_colind", _i32), ("tr_colval", _f),
        ("va_nrows", C.c_int32), ("va_rowptr", _i64), ("va_rowind", _i32), ("va_rowval", _f),
    ]


def tree_shape(K):
    L, Cc = C.c_int(), C.c_int()
    lib.orc_tree_shape(K, C.byref(L), C.byref(Cc))
    return L.value, Cc.value


def dot(a, b, mode=DOT_SEQ):
    return lib.orc_dot(F(a), F(b), len(a), mode)


def init_factors(seed, nU, nI, K):
    U = np.empty((nU, K), np.float32)
    V = np.empty((nI, K), np.float32)
    lib.orc_init_factors(seed, nU, nI, K, F(U), F(V))
    return U, V


class MT:
    """std::mt19937 + the reference's shuffles."""

    def __init__(self, seed):
        self.h = C.c_void_p(lib.orc_mt_create(C.c_uint32(seed)))

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_mt_free(self.h)
            self.h = None

    def next(self):
        return lib.orc_mt_next(self.h)

    def shuffle_u64(self, arr):
        lib.orc_mt_shuffle_u64(self.h, U64(arr), C.c_int64(len(arr)))

    def shuffle_i32(self, arr):
        lib.orc_mt_shuffle_i32(self.h, I32(arr), C.c_int64(len(arr)))

    def par_block_shuffle_u64(self, arr, nthreads):
        lib.orc_mt_par_block_shuffle_u64(self.h, U64(arr), C.c_int64(len(arr)), nthreads)

    def block_seq(self, dim):
        r = np.empty(dim, np.int32)
        c = np.empty(dim, np.int32)
        lib.orc_mt_block_seq(self.h, dim, I32(r), I32(c))
        return r, c


def create_col_index(nrows, ncols, rowptr, rowind, rowval):
    nnz = int(rowptr[nrows])
    colptr = np.empty(ncols + 1, np.int64)
    colind = np.empty(nnz, np.int32)
    colval = np.empty(nnz, np.float32)
    lib.orc_create_col_index(nrows, ncols, I64(rowptr), I32(rowind), F(rowval),
                             I64(colptr), I32(colind), F(colval))
    return colptr, colind, colval


def invalid(nrows, ncols, rowptr, rowind, nUsers, nItems):
    invU = np.empty(nUsers, np.uint8)
    invI = np.empty(nItems, np.uint8)
    lib.orc_invalid(nrows, ncols, I64(rowptr), I32(rowind), nUsers, nItems, U8(invU), U8(invI))
    return invU, invI


def read_csr_text(path):
    nr, nc, nz = C.c_int32(), C.c_int32(), C.c_int64()
    rc = lib.orc_read_csr_text(path.encode(), C.byref(nr), C.byref(nc), C.byref(nz), None, None, None)
    if rc:
        raise IOError("orc_read_csr_text rc=%d" % rc)
    rowptr = np.empty(nr.value + 1, np.int64)
    rowind = np.empty(nz.value, np.int32)
    rowval = np.empty(nz.value, np.float32)
    rc = lib.orc_read_csr_text(path.encode(), C.byref(nr), C.byref(nc), C.byref(nz),
                               I64(rowptr), I32(rowind), F(rowval))
    assert rc == 0
    return nr.value, nc.value, rowptr, rowind, rowval


def write_csr_text(path, nrows, rowptr, rowind, rowval):
    rc = lib.orc_write_csr_text(path.encode(), nrows, I64(rowptr), I32(rowind), F(rowval))
    assert rc == 0


def write_mat(path, M):
    assert lib.orc_write_mat(path.encode(), F(M), M.shape[0], M.shape[1]) == 0


def read_mat(path, nrows, ncols):
    M = np.empty((nrows, ncols), np.float32)
    rc = lib.orc_read_mat(path.encode(), F(M), nrows, ncols)
    if rc:
        raise IOError("orc_read_mat rc=%d" % rc)
    return M


def sgd_pass(U, V, u, i, r, order, lr, uReg, iReg, arith=ARITH_REF64, dot_mode=DOT_SEQ):
    K = U.shape[1]
    lib.orc_sgd_pass(K, F(U), F(V), I32(u), I32(i), F(r), U64(order) if order is not None else None,
                     C.c_int64(len(order) if order is not None else len(u)),
                     C.c_float(lr), C.c_float(uReg), C.c_float(iReg), arith, dot_mode)


def sgd_pass_dimreg(U, V, u, i, r, order, lr, regk, dot_mode=DOT_SEQ):
    K = U.shape[1]
    regk = np.ascontiguousarray(regk, np.float32)
    lib.orc_sgd_pass_dimreg(K, F(U), F(V), I32(u), I32(i), F(r), U64(order) if order is not None else None,
                            C.c_int64(len(order) if order is not None else len(u)), C.c_float(lr), F(regk), dot_mode)


def objective_sing(U, V, nUsers, nItems, nrows, rowptr, rowind, rowval, invU, invI, sing, dot_mode=DOT_SEQ):
    lib.orc_objective_sing.restype = C.c_double
    sse, ur, ir = C.c_double(), C.c_double(), C.c_double()
    sing = np.ascontiguousarray(sing, np.float32)
    o = lib.orc_objective_sing(U.shape[1], F(U), F(V), nUsers, nItems, nrows, I64(rowptr), I32(rowind), F(rowval),
                               U8(invU), U8(invI), F(sing), dot_mode, C.byref(sse), C.byref(ur), C.byref(ir))
    return o, sse.value, ur.value, ir.value


def _f64p(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(C.c_void_p)


def ifw_pop(nrows, ncols, rowptr, rowind, invU, invI):
    """(userFreq, itemFreq, invPopU, invPopI) as modelInvPopMF.cpp:84-113 builds them (float64)."""
    uf, itf = np.zeros(nrows), np.zeros(ncols)
    pu, pi = np.zeros(nrows), np.zeros(ncols)
    lib.orc_ifw_pop(nrows, ncols, I64(rowptr), I32(rowind), U8(invU), U8(invI), _f64p(uf), _f64p(itf), _f64p(pu), _f64p(pi))
    return uf, itf, pu, pi


def sgd_pass_ifw(U, V, u, i, r, order, lr, uReg, iReg, pop, rho, dot_mode=DOT_SEQ):
    uf, itf, pu, pi = pop
    lib.orc_sgd_pass_ifw(U.shape[1], F(U), F(V), I32(u), I32(i), F(r), U64(order) if order is not None else None,
                         C.c_int64(len(order) if order is not None else len(u)), C.c_float(lr), C.c_float(uReg),
                         C.c_float(iReg), _f64p(uf), _f64p(itf), _f64p(pu), _f64p(pi), C.c_float(rho), dot_mode)


def objective_ifw(U, V, nUsers, nItems, nrows, rowptr, rowind, rowval, invU, invI, uReg, iReg, pop, rho, dot_mode=DOT_SEQ):
    uf, itf, pu, pi = pop
    lib.orc_objective_ifw.restype = C.c_double
    w = C.c_double()
    o = lib.orc_objective_ifw(U.shape[1], F(U), F(V), nUsers, nItems, nrows, I64(rowptr), I32(rowind), F(rowval), U8(invU),
                              U8(invI), C.c_float(uReg), C.c_float(iReg), _f64p(uf), _f64p(itf), _f64p(pu), _f64p(pi), C.c_float(rho),
                              dot_mode, C.byref(w))
    return o, w.value


def tmf_ranks(freq, meanFreq, stdFreq, rho, alpha, K):
    freq = np.ascontiguousarray(freq, np.float64)
    out = np.zeros(len(freq), np.int32)
    lib.orc_tmf_ranks(len(freq), _f64p(freq), C.c_double(meanFreq), C.c_double(stdFreq), C.c_float(rho), C.c_float(alpha), K, I32(out))
    return out


def sgd_pass_tmf(U, V, u, i, r, order, lr, uReg, iReg, uf, itf, ru, ri, dot_mode=DOT_SEQ):
    lib.orc_sgd_pass_tmf(U.shape[1], F(U), F(V), I32(u), I32(i), F(r), U64(order) if order is not None else None,
                         C.c_int64(len(order) if order is not None else len(u)), C.c_float(lr), C.c_float(uReg), C.c_float(iReg),
                         _f64p(uf), _f64p(itf), I32(ru), I32(ri), dot_mode)


def rmse_tmf(U, V, nUsers, nItems, nrows, rowptr, rowind, rowval, invU, invI, uf, itf, ru, ri, dot_mode=DOT_SEQ):
    lib.orc_rmse_tmf.restype = C.c_double
    sse, cnt = C.c_double(), C.c_int64()
    r = lib.orc_rmse_tmf(U.shape[1], F(U), F(V), nUsers, nItems, nrows, I64(rowptr), I32(rowind), F(rowval), U8(invU), U8(invI),
                         _f64p(uf), _f64p(itf), I32(ru), I32(ri), dot_mode, C.byref(sse), C.byref(cnt))
    return r, sse.value, cnt.value


def cdf_ranks(K):
    out = np.zeros(K, np.int32)
    lib.orc_cdf_ranks(K, I32(out))
    return out


def poisson_rank(lam, seed, epoch, u, item, K):
    return lib.orc_poisson_rank(int(lam), C.c_uint32(seed), C.c_uint32(epoch), C.c_uint32(u), C.c_uint32(item), K)


def sgd_pass_tmfd(U, V, u, i, r, order, lr, uReg, iReg, uf, itf, lu, li, seed, epoch, dot_mode=DOT_SEQ):
    lib.orc_sgd_pass_tmfd(U.shape[1], F(U), F(V), I32(u), I32(i), F(r), U64(order) if order is not None else None,
                          C.c_int64(len(order) if order is not None else len(u)), C.c_float(lr), C.c_float(uReg), C.c_float(iReg),
                          _f64p(uf), _f64p(itf), I32(lu), I32(li), C.c_uint32(seed), C.c_uint32(epoch), dot_mode)


def sgd_hogwild(U, V, u, i, r, order, lr, uReg, iReg, arith=ARITH_F32, dot_mode=DOT_SEQ, nthreads=1):
    K = U.shape[1]
    lib.orc_sgd_hogwild(K, F(U), F(V), I32(u), I32(i), F(r), U64(order) if order is not None else None,
                        C.c_int64(len(order) if order is not None else len(u)),
                        C.c_float(lr), C.c_float(uReg), C.c_float(iReg), arith, dot_mode, nthreads)


def objective(U, V, nUsers, nItems, nrows, rowptr, rowind, rowval, invU, invI, uReg, iReg,
              dot_mode=DOT_SEQ):
    sse, un, inn = C.c_double(), C.c_double(), C.c_double()
    obj = lib.orc_objective(U.shape[1], F(U), F(V), nUsers, nItems, nrows, I64(rowptr), I32(rowind),
                            F(rowval), U8(invU), U8(invI), C.c_float(uReg), C.c_float(iReg), dot_mode,
                            C.byref(sse), C.byref(un), C.byref(inn))
    return obj, sse.value, un.value, inn.value


def rmse(U, V, nUsers, nItems, nrows, rowptr, rowind, rowval, invU, invI, dot_mode=DOT_SEQ):
    sse, cnt = C.c_double(), C.c_int64()
    v = lib.orc_rmse(U.shape[1], F(U), F(V), nUsers, nItems, nrows, I64(rowptr), I32(rowind), F(rowval),
                     U8(invU), U8(invI), dot_mode, C.byref(sse), C.byref(cnt))
    return v, sse.value, cnt.value


def ldlt_solve(A, b):
    K = len(b)
    A = np.ascontiguousarray(A, np.float32).copy()
    x = np.empty(K, np.float32)
    lib.orc_ldlt_solve(K, F(A), F(np.ascontiguousarray(b, np.float32)), F(x))
    return x


def als_half(side, X, Y, nX, ptr, ind, val, invX, reg, nthreads=1):
    lib.orc_als_half(side, X.shape[1], F(X), F(Y), nX, I64(ptr), I32(ind), F(val), U8(invX),
                     C.c_float(reg), nthreads)


def ccdpp_rank1(k, U, V, nUsers, nItems, ncols, rowptr, rowind, res_row, colptr, colind, res_col,
                invU, invI, uReg, iReg, add_back, inner=5, freq_thresh=-1.0, nthreads=1):
    lib.orc_ccdpp_rank1(U.shape[1], k, F(U), F(V), nUsers, nItems, ncols, I64(rowptr), I32(rowind),
                        F(res_row), I64(colptr), I32(colind), F(res_col), U8(invU), U8(invI),
                        C.c_float(uReg), C.c_float(iReg), int(add_back), inner,
                        C.c_float(freq_thresh), nthreads)


def ccd_iter(U, V, nUsers, nItems, ncols, rowptr, rowind, res_row, colptr, colind, res_col, invU, invI,
             uReg, iReg, mt=None, uorder=None, iorder=None, orders_given=False):
    """One trainCCD iteration.  uorder/iorder: uint16 [n][K] factor orders per row -- used when
    orders_given, else filled with what std::shuffle(udims, mt) produced."""
    for o in (uorder, iorder):
        assert o is None or (o.dtype == np.uint16 and o.flags.c_contiguous)
    P = lambda o: o.ctypes.data_as(C.c_void_p) if o is not None else None
    lib.orc_ccd_iter(U.shape[1], F(U), F(V), nUsers, nItems, ncols, I64(rowptr), I32(rowind), F(res_row),
                     I64(colptr), I32(colind), F(res_col), U8(invU), U8(invI), C.c_float(uReg),
                     C.c_float(iReg), mt.h if mt is not None else None, P(uorder), P(iorder), int(orders_given))


def train(method, U0, V0, train_csr, val_csr, nUsers, nItems, K, maxIter, seed, lr, uReg, iReg,
          nthreads=1, dot_mode=DOT_SEQ):
    """train_csr = (nrows, ncols, rowptr, rowind, rowval, colptr, colind, colval);
    val_csr = (nrows, rowptr, rowind, rowval).  Returns a dict."""
    U = np.ascontiguousarray(U0, np.float32).copy()
    V = np.ascontiguousarray(V0, np.float32).copy()
    Ub, Vb = np.empty_like(U), np.empty_like(V)
    obj = np.full(maxIter, np.nan)
    val = np.full(maxIter, np.nan)
    invU = np.empty(nUsers, np.uint8)
    invI = np.empty(nItems, np.uint8)
    cfg = TrainCfg()
    cfg.method, cfg.K, cfg.maxIter, cfg.seed = method, K, maxIter, seed
    cfg.nthreads, cfg.dot_mode = nthreads, dot_mode
    cfg.uReg, cfg.iReg, cfg.learnRate = uReg, iReg, lr
    cfg.nUsers, cfg.nItems = nUsers, nItems
    (cfg.tr_nrows, cfg.tr_ncols) = train_csr[0], train_csr[1]
    cfg.tr_rowptr, cfg.tr_rowind, cfg.tr_rowval = I64(train_csr[2]), I32(train_csr[3]), F(train_csr[4])
    cfg.tr_colptr, cfg.tr_colind, cfg.tr_colval = I64(train_csr[5]), I32(train_csr[6]), F(train_csr[7])
    cfg.va_nrows = val_csr[0]
    cfg.va_rowptr, cfg.va_rowind, cfg.va_rowval = I64(val_csr[1]), I32(val_csr[2]), F(val_csr[3])
    bestIter, flr = C.c_int32(), C.c_float()
    n = lib.orc_train(C.byref(cfg), F(U), F(V), F(Ub), F(Vb), p(obj, _d), p(val, _d),
                      C.byref(bestIter), C.byref(flr), U8(invU), U8(invI))
    return dict(iters=n, U=U, V=V, Ubest=Ub, Vbest=Vb, obj=obj[:n], val=val[:n],
                bestIter=bestIter.value, learnRate=flr.value, invU=invU, invI=invI)


def time_hogwild(U, V, u, i, r, nU, nI, K, lr, uReg, iReg, nthreads, colmajor, epochs=1):
    return lib.orc_time_hogwild(K, nU, nI, F(U), F(V), I32(u), I32(i), F(r), C.c_int64(len(u)),
                                C.c_float(lr), C.c_float(uReg), C.c_float(iReg), nthreads,
                                int(colmajor), epochs)


def time_strat(U, V, rowptr, rowind, rowval, nrows, ncols, invU, invI, T, lr, uReg, iReg, seed=1, epochs=1):
    """seconds for `epochs` stratified epochs (trainSGDPar) with T parts / T OpenMP threads"""
    mt = MT(seed)
    h = C.c_void_p(lib.orc_strat_create(mt.h, nrows, ncols, U8(invU), U8(invI), T))
    try:
        return lib.orc_time_strat(h, mt.h, U.shape[1], F(U), F(V), I64(rowptr), I32(rowind), F(rowval),
                                  C.c_float(lr), C.c_float(uReg), C.c_float(iReg), epochs)
    finally:
        lib.orc_strat_free(h)


def init_bias(seed, nU, nI, K):
    ub, ib = np.empty(nU, np.float32), np.empty(nI, np.float32)
    lib.orc_init_bias(int(seed), nU, nI, K, F(ub), F(ib))
    return ub, ib


def bias_pass(ub, ib, u, i, r, order, lr, uReg, iReg):
    n = len(order) if order is not None else len(u)
    lib.orc_bias_pass(F(ub), F(ib), I32(u), I32(i), F(r), p(order, _u64) if order is not None else None, C.c_int64(n),
                      C.c_float(lr), C.c_float(uReg), C.c_float(iReg))


def bias_eval(ub, ib, nUsers, nItems, nrows, rowptr, rowind, rowval, invU, invI, uReg=0.0, iReg=0.0):
    sse, cnt, a, b = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    obj = lib.orc_bias_eval(F(ub), F(ib), nUsers, nItems, nrows, I64(rowptr), I32(rowind), F(rowval), U8(invU), U8(invI),
                            C.c_float(uReg), C.c_float(iReg), C.byref(sse), C.byref(cnt), C.byref(a), C.byref(b))
    return obj, sse.value, cnt.value, a.value, b.value


def split_colors(nnz, testPc, valPc, seed):
    color = np.empty(nnz, np.int32)
    lib.orc_split_colors(C.c_int64(nnz), C.c_float(testPc), C.c_float(valPc), int(seed), I32(color))
    return color


def rand_pairs(nUsers, nItems, seed, nnz):
    n = lib.orc_rand_pairs(nUsers, nItems, int(seed), nnz, None)
    pairs = np.empty((n, 2), np.int32)
    lib.orc_rand_pairs(nUsers, nItems, int(seed), nnz, I32(pairs))
    return pairs


def max_threads():
    return lib.orc_max_threads()
