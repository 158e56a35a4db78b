"""ctypes view of include/mfx.h.  One method per C entry point; arrays are numpy."""
import ctypes as C

import numpy as np

from . import _lib

OK = 0
MAT_TRAIN, MAT_VAL, MAT_TEST = 0, 1, 2
ROWMAJOR, COLMAJOR = 0, 1
SNAP_CURRENT, SNAP_BEST = 0, 1
SIDE_USERS, SIDE_ITEMS = 0, 1
SGD_HOGWILD, SGD_SERIAL, SGD_USERS, SGD_TILED, SGD_LEVELS = 0, 1, 2, 3, 4
ORDER_DEVICE, ORDER_HOST, ORDER_NATURAL = 0, 1, 2
ARITH_REF64, ARITH_REF64F, ARITH_F32 = 0, 1, 2
SGD_F_ONE_GROUP, SGD_F_COUNT_VISITS, SGD_F_DRAIN_ONLY = 1, 2, 4
REDUCE_DELTA_SUM, REDUCE_AVERAGE = 0, 1
K_SGD, K_PERMUTE, K_EVAL, K_ALS_GRAM, K_ALS_SOLVE, K_CCD_ROW, K_CCD_COL, K_CCD_RESID, K_SGD_SWEEP, K_CD = range(10)
E_NODEVICE = -6


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)


class MfxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mfx error %d: %s" % (code, msg))
        self.code = code


class SgdOpts(C.Structure):
    _fields_ = [("mode", C.c_int32), ("order", C.c_int32), ("arith", C.c_int32),
                ("learnRate", C.c_float), ("uReg", C.c_float), ("iReg", C.c_float),
                ("seed", C.c_uint32), ("epoch", C.c_int32), ("blocks", C.c_int32), ("own", C.c_int32),
                ("first", C.c_int64), ("count", C.c_int64), ("flags", C.c_int32), ("item_part", C.c_int32)]


class EvalOut(C.Structure):
    _fields_ = [("sse", C.c_double), ("n", C.c_int64), ("unorm2", C.c_double), ("inorm2", C.c_double)]


def _p(a, dtype):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype)
    return a, a.ctypes.data_as(C.c_void_p)


def tree_shape(K):
    if K <= 16:
        return 4, 1
    if K <= 32:
        return 8, 1
    return 16, (K + 63) // 64


class Ctx:
    """One mfx_ctx (one device, one stream)."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        self.lib.mfx_last_error.restype = C.c_char_p
        self.h = C.c_void_p()
        rc = self.lib.mfx_create(int(device), C.byref(self.h))
        if rc != OK:
            msg = self.lib.mfx_last_error(None).decode()
            self.h = None
            raise MfxError(rc, msg)
        self.nU = self.nI = self.K = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.mfx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            raise MfxError(rc, self.lib.mfx_last_error(self.h).decode())

    # ---- data / model -------------------------------------------------------
    def set_csr(self, which, nrows, ncols, rowptr, rowind, rowval, colptr=None, colind=None, colval=None):
        keep = [_p(rowptr, np.int64), _p(rowind, np.int32), _p(rowval, np.float32),
                _p(colptr, np.int64), _p(colind, np.int32), _p(colval, np.float32)]
        ptrs = [k[1] if k else None for k in keep]
        self._chk(self.lib.mfx_set_csr(self.h, which, C.c_int32(nrows), C.c_int32(ncols), *ptrs))

    def set_model(self, nUsers, nItems, K):
        self._chk(self.lib.mfx_set_model(self.h, C.c_int32(nUsers), C.c_int32(nItems), C.c_int32(K)))
        self.nU, self.nI, self.K = nUsers, nItems, K

    def set_factors(self, U, V, layout=ROWMAJOR):
        ku, kv = _p(U, np.float32), _p(V, np.float32)
        if ku is not None:
            assert ku[0].size == self.nU * self.K
        if kv is not None:
            assert kv[0].size == self.nI * self.K
        self._chk(self.lib.mfx_set_factors(self.h, ku[1] if ku else None, kv[1] if kv else None, layout))

    def get_factors(self, snapshot=SNAP_CURRENT, layout=ROWMAJOR):
        shape_u = (self.nU, self.K) if layout == ROWMAJOR else (self.K, self.nU)
        shape_v = (self.nI, self.K) if layout == ROWMAJOR else (self.K, self.nI)
        U = np.empty(shape_u, np.float32)
        V = np.empty(shape_v, np.float32)
        self._chk(self.lib.mfx_get_factors(self.h, snapshot, U.ctypes.data_as(C.c_void_p),
                                           V.ctypes.data_as(C.c_void_p), layout))
        return U, V

    def compute_invalid(self):
        invU = np.empty(self.nU, np.uint8)
        invI = np.empty(self.nI, np.uint8)
        self._chk(self.lib.mfx_compute_invalid(self.h, invU.ctypes.data_as(C.c_void_p),
                                               invI.ctypes.data_as(C.c_void_p)))
        return invU, invI

    def snapshot_best(self):
        self._chk(self.lib.mfx_snapshot_best(self.h))

    def restore_best(self):
        self._chk(self.lib.mfx_restore_best(self.h))

    def synchronize(self):
        self._chk(self.lib.mfx_synchronize(self.h))

    # ---- SGD ----------------------------------------------------------------
    def sgd_set_order(self, perm):
        k = _p(perm, np.uint64)
        self._chk(self.lib.mfx_sgd_set_order(self.h, k[1], C.c_int64(k[0].size)))

    def sgd_set_order32(self, perm):
        k = _p(perm, np.uint32)
        self._chk(self.lib.mfx_sgd_set_order32(self.h, k[1], C.c_int64(k[0].size)))

    def sgd_apply_swaps32(self, pos):
        """std::shuffle's swaps on the device: for i = 1 .. n-1: swap(a[i], a[pos[i]]) on the list of sgd_set_order32"""
        k = _p(pos, np.uint32)
        self._chk(self.lib.mfx_sgd_apply_swaps32(self.h, k[1], C.c_int64(k[0].size)))

    def debug_order32(self):
        n = C.c_int64(0)
        self._chk(self.lib.mfx_debug_order32(self.h, None, C.c_int64(0), C.byref(n)))
        out = np.empty(n.value, np.uint32)
        self._chk(self.lib.mfx_debug_order32(self.h, out.ctypes.data_as(C.c_void_p), C.c_int64(out.size), C.byref(n)))
        return out

    def sgd_epoch(self, lr, uReg, iReg, mode=SGD_HOGWILD, order=ORDER_DEVICE, arith=ARITH_F32, seed=1,
                  epoch=0, first=0, count=0, blocks=0, own=0, flags=0, item_part=0):
        o = SgdOpts(mode, order, arith, lr, uReg, iReg, seed, epoch, blocks, own, first, count, flags, item_part)
        self._chk(self.lib.mfx_sgd_epoch(self.h, C.byref(o)))

    def debug_epoch_list(self):
        n = C.c_int64()
        self._chk(self.lib.mfx_debug_epoch_list(self.h, None, None, None, C.c_int64(0), C.byref(n)))
        u = np.empty(n.value, np.int32)
        i = np.empty(n.value, np.int32)
        r = np.empty(n.value, np.float32)
        self._chk(self.lib.mfx_debug_epoch_list(self.h, u.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p),
                                                r.ctypes.data_as(C.c_void_p), C.c_int64(n.value), C.byref(n)))
        return u, i, r

    def debug_levels_info(self):
        info = (C.c_int64 * 4)()
        ms = C.c_double()
        self._chk(self.lib.mfx_debug_levels_info(self.h, info, C.byref(ms)))
        return list(info), ms.value

    def debug_flow_queues(self):
        n, g = C.c_int64(), C.c_int64()
        self._chk(self.lib.mfx_debug_flow_queues(self.h, None, C.c_int64(0), None, C.byref(n), C.byref(g)))
        rec = np.empty((n.value, 4), np.int32)
        off = np.empty(g.value + 1, np.int64)
        self._chk(self.lib.mfx_debug_flow_queues(self.h, rec.ctypes.data_as(C.c_void_p), C.c_int64(n.value), off.ctypes.data_as(C.c_void_p),
                                                 C.byref(n), C.byref(g)))
        return rec, off

    def debug_visit_counts(self):
        n = C.c_int64()
        self._chk(self.lib.mfx_debug_visit_counts(self.h, None, C.c_int64(0), C.byref(n)))
        c = np.empty(n.value, np.uint32)
        self._chk(self.lib.mfx_debug_visit_counts(self.h, c.ctypes.data_as(C.c_void_p), C.c_int64(n.value), C.byref(n)))
        return c

    def debug_raise_drain_abort(self):
        self._chk(self.lib.mfx_debug_raise_drain_abort(self.h))

    def debug_tile_blocks(self, n_users, n_items):
        ub = np.empty(n_users, np.uint8)
        ib = np.empty(n_items, np.uint8)
        self._chk(self.lib.mfx_debug_tile_blocks(self.h, ub.ctypes.data_as(C.c_void_p), C.c_int64(n_users),
                                                 ib.ctypes.data_as(C.c_void_p), C.c_int64(n_items)))
        return ub, ib

    def debug_slots_digest(self):
        counts = (C.c_int64 * 4)()
        sums = (C.c_uint64 * 5)()
        self._chk(self.lib.mfx_debug_slots_digest(self.h, counts, sums))
        return list(counts), list(sums)

    def debug_col_view(self, ncols, nnz):
        cp = np.empty(ncols + 1, np.int64)
        ci = np.empty(nnz, np.int32)
        cv = np.empty(nnz, np.float32)
        self._chk(self.lib.mfx_debug_col_view(self.h, cp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                              cv.ctypes.data_as(C.c_void_p)))
        return cp, ci, cv

    # ---- evaluation -----------------------------------------------------------
    def eval(self, which, snapshot=SNAP_CURRENT, with_norms=False):
        out = EvalOut()
        self._chk(self.lib.mfx_eval(self.h, which, snapshot, int(with_norms), C.byref(out)))
        return out

    def eval2(self, whichA, normsA, whichB, normsB=False, snapshot=SNAP_CURRENT):
        a, b = EvalOut(), EvalOut()
        self._chk(self.lib.mfx_eval2(self.h, whichA, int(normsA), whichB, int(normsB), snapshot, C.byref(a), C.byref(b)))
        return a, b

    def eval_filtered(self, which, keep_users=None, keep_items=None, snapshot=SNAP_CURRENT):
        out = EvalOut()
        ku, ki = _p(keep_users, np.uint8), _p(keep_items, np.uint8)
        self._chk(self.lib.mfx_eval_filtered(self.h, which, snapshot, ku[1] if ku else None, ki[1] if ki else None, C.byref(out)))
        return out

    def objective(self, uReg, iReg, snapshot=SNAP_CURRENT):
        """Model::objective (model.cpp:1770-1815) from the device sums."""
        o = self.eval(MAT_TRAIN, snapshot, True)
        return o.sse + float(np.float32(uReg)) * o.unorm2 + float(np.float32(iReg)) * o.inorm2

    def rmse(self, which, snapshot=SNAP_CURRENT):
        o = self.eval(which, snapshot, False)
        return float(np.sqrt(o.sse / o.n)) if o.n else float("nan")

    # ---- ALS / CCD++ ----------------------------------------------------------
    def als_half_sweep(self, side, reg):
        self._chk(self.lib.mfx_als_half_sweep(self.h, side, C.c_float(reg)))

    def ccdpp_begin(self):
        self._chk(self.lib.mfx_ccdpp_begin(self.h))

    def ccdpp_rank1(self, k, uReg, iReg, add_back, inner=5, freq_thresh=-1.0):
        self._chk(self.lib.mfx_ccdpp_rank1(self.h, C.c_int32(k), C.c_int32(inner), C.c_float(uReg),
                                           C.c_float(iReg), C.c_int32(int(add_back)), C.c_float(freq_thresh)))

    def ccdpp_end(self):
        self._chk(self.lib.mfx_ccdpp_end(self.h))

    def debug_residuals(self, nnz):
        a = np.empty(nnz, np.float32)
        b = np.empty(nnz, np.float32)
        self._chk(self.lib.mfx_debug_residuals(self.h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))
        return a, b

    # ---- trainSGDParSVD -------------------------------------------------------------
    def svd_init(self, power_iters=8, oversample=10, seed=1):
        s = np.empty(self.K, np.float32)
        self._chk(self.lib.mfx_svd_init(self.h, C.c_int32(power_iters), C.c_int32(oversample), C.c_uint32(seed),
                                        s.ctypes.data_as(C.c_void_p)))
        return s

    def sgd_set_dim_reg(self, reg):
        k = _p(reg, np.float32)
        self._chk(self.lib.mfx_sgd_set_dim_reg(self.h, k[1] if k else None))

    def eval_weighted(self, which, w, snapshot=SNAP_CURRENT):
        out = EvalOut()
        k = _p(w, np.float32)
        self._chk(self.lib.mfx_eval_weighted(self.h, which, snapshot, k[1], C.byref(out)))
        return out

    # ---- ModelInvPopMF (IFWMF) --------------------------------------------------------
    def sgd_set_ifw(self, userFreq=None, invPopU=None, itemFreq=None, invPopI=None, rho=0.0):
        ks = [_p(a, np.float32) for a in (userFreq, invPopU, itemFreq, invPopI)]
        self._chk(self.lib.mfx_sgd_set_ifw(self.h, *[k[1] if k else None for k in ks], C.c_float(rho)))

    def eval_ifw(self, snapshot=SNAP_CURRENT):
        out = EvalOut()
        self._chk(self.lib.mfx_eval_ifw(self.h, snapshot, C.byref(out)))
        return out

    # ---- ModelDropoutSigmoid (TMF) ------------------------------------------------------
    def set_tmf(self, userFreq=None, userRank=None, itemFreq=None, itemRank=None):
        ks = [_p(userFreq, np.float32), _p(userRank, np.int32), _p(itemFreq, np.float32), _p(itemRank, np.int32)]
        self._chk(self.lib.mfx_set_tmf(self.h, *[k[1] if k else None for k in ks]))

    def set_tmf_dropout(self, userLambda=None, itemLambda=None, seed=1):
        ks = [_p(userLambda, np.int32), _p(itemLambda, np.int32)]
        self._chk(self.lib.mfx_set_tmf_dropout(self.h, *[k[1] if k else None for k in ks], C.c_uint32(seed)))

    # ---- ModelMFBias -----------------------------------------------------------------------
    def bias_set(self, uBias, iBias):
        a, b = _p(uBias, np.float32), _p(iBias, np.float32)
        assert a[0].size == self.nU and b[0].size == self.nI
        self._chk(self.lib.mfx_bias_set(self.h, a[1], b[1]))

    def bias_get(self, snapshot=SNAP_CURRENT):
        ub, ib = np.empty(self.nU, np.float32), np.empty(self.nI, np.float32)
        self._chk(self.lib.mfx_bias_get(self.h, snapshot, ub.ctypes.data_as(C.c_void_p), ib.ctypes.data_as(C.c_void_p)))
        return ub, ib

    def bias_epoch(self, lr, uReg, iReg, mode=SGD_LEVELS, order=ORDER_HOST, seed=1, epoch=0, first=0, count=0):
        o = SgdOpts(mode, order, ARITH_REF64, lr, uReg, iReg, seed, epoch, 0, 0, first, count, 0, 0)
        self._chk(self.lib.mfx_bias_epoch(self.h, C.byref(o)))

    def bias_eval(self, which, snapshot=SNAP_CURRENT):
        out = EvalOut()
        self._chk(self.lib.mfx_bias_eval(self.h, which, snapshot, C.byref(out)))
        return out

    # ---- cyclic coordinate descent (trainCCD) ---------------------------------------
    def ccd_begin(self):
        self._chk(self.lib.mfx_ccd_begin(self.h))

    def ccd_sweep(self, side, reg, order=None, seed=1, it=0):
        keep = _p(order, np.uint16)
        self._chk(self.lib.mfx_ccd_sweep(self.h, C.c_int32(side), C.c_float(reg), keep[1] if keep else None,
                                         C.c_uint32(seed), C.c_int32(it)))

    def ccd_end(self):
        self._chk(self.lib.mfx_ccd_end(self.h))

    def debug_ccd_residuals(self, nnz):
        a = np.empty(nnz, np.float32)
        b = np.empty(nnz, np.float32)
        self._chk(self.lib.mfx_debug_ccd_residuals(self.h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))
        return a, b

    # ---- multi-GPU ------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        rc = lib.mfx_comm_unique_id(buf)
        if rc != OK:
            lib.mfx_last_error.restype = C.c_char_p
            raise MfxError(rc, lib.mfx_last_error(None).decode())
        return bytes(buf.raw)

    def comm_init(self, nranks, rank, uid):
        self._chk(self.lib.mfx_comm_init(self.h, nranks, rank, C.c_char_p(uid)))

    def comm_init_external(self, nranks, rank, reduce):
        """reduce(array) must sum the numpy array (float32 or float64, a view of the library's host buffer) over
        all ranks IN PLACE -- e.g. torch.distributed.all_reduce(torch.from_numpy(a)) over gloo, or MPI."""
        def thunk(user, ptr, count, dtype):
            try:
                ct = C.c_double if dtype else C.c_float
                reduce(np.ctypeslib.as_array((ct * count).from_address(ptr)))
                return 0
            except Exception:                       # must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._reduce_cb = REDUCE_FN(thunk)          # keep the callback alive as long as the ctx
        self._chk(self.lib.mfx_comm_init_external(self.h, nranks, rank, self._reduce_cb, None))

    def comm_mark_synced(self):
        self._chk(self.lib.mfx_comm_mark_synced(self.h))

    def comm_destroy(self):
        self._chk(self.lib.mfx_comm_destroy(self.h))

    def allreduce_item_factors(self, op=REDUCE_DELTA_SUM):
        self._chk(self.lib.mfx_allreduce_item_factors(self.h, op))

    # item-part rotation (include/mfx.h): parts = item % nparts, one part per rank and sub-epoch
    def set_item_parts(self, nparts):
        self._chk(self.lib.mfx_sgd_set_item_parts(self.h, int(nparts)))

    def rotate_item_part(self, send_part, recv_part):
        self._chk(self.lib.mfx_rotate_item_part(self.h, int(send_part), int(recv_part)))

    def allgather_item_parts(self, my_part):
        self._chk(self.lib.mfx_allgather_item_parts(self.h, int(my_part)))

    def allreduce_f64(self, vals):
        a = np.ascontiguousarray(vals, np.float64).copy()
        self._chk(self.lib.mfx_allreduce_f64(self.h, a.ctypes.data_as(C.c_void_p), a.size))
        return a

    # ---- measurement ----------------------------------------------------------
    def prof_enable(self, on=True):
        """on: False / True / N > 1 = events around the launches of every N-th sgd_epoch only"""
        self._chk(self.lib.mfx_prof_enable(self.h, int(on)))

    def prof_reset(self):
        self._chk(self.lib.mfx_prof_reset(self.h))

    def prof_get(self, kernel):
        ms, n = C.c_double(), C.c_int64()
        self._chk(self.lib.mfx_prof_get(self.h, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value
