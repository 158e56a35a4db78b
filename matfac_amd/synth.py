"""Synthetic rating matrices of the BASELINE.md shapes (no MovieLens/Netflix files offline).

Mirrors the reference's own synthetic workflow -- ground-truth low-rank factors
(python/genLatFacs.py:17-37), a CSR sampled from them (io.cpp:726-787) and a per-rating
train/test/val split (io.cpp:410-459) -- with the power-law degree/popularity skew that
SURVEY.md 8(d) asks for.  Host-side data preparation only; nothing here is on the hot path.
"""
import ctypes as C
import os

import numpy as np

SHAPES = {
    "C1": dict(nU=943, nI=1682, nnz=100_000, K=10),             # MovieLens-100K shape
    "C2": dict(nU=138_493, nI=26_744, nnz=20_000_263, K=64),    # MovieLens-20M shape
    "C4": dict(nU=480_189, nI=17_770, nnz=100_000_000, K=128),  # Netflix shape
    # BASELINE.json config 5: 10 M x 1 M, 1 B ratings, rank 256 -- ONE matrix defined as 8 fixed user shards of 1.25 M users
    # (make_c5_shards): the union is the same whatever the number of GPUs it is cut over
    "C5": dict(nU=10_000_000, nI=1_000_000, nnz=1_000_000_000, K=256),
}
C5_SHARDS = 8


class CSR:
    __slots__ = ("nrows", "ncols", "rowptr", "rowind", "rowval")

    def __init__(self, nrows, ncols, rowptr, rowind, rowval):
        self.nrows, self.ncols = int(nrows), int(ncols)
        self.rowptr = np.ascontiguousarray(rowptr, np.int64)
        self.rowind = np.ascontiguousarray(rowind, np.int32)
        self.rowval = np.ascontiguousarray(rowval, np.float32)

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    def rowids(self):
        return np.repeat(np.arange(self.nrows, dtype=np.int32), np.diff(self.rowptr)).astype(np.int32)

    def col_view(self):
        """gk_csr_CreateIndex(COL): stable, users ascending inside a column."""
        order = np.argsort(self.rowind, kind="stable")
        colptr = np.zeros(self.ncols + 1, np.int64)
        np.cumsum(np.bincount(self.rowind, minlength=self.ncols), out=colptr[1:])
        return colptr, self.rowids()[order].astype(np.int32), self.rowval[order]


_lib = None


def _host():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmfhost.so")
        if not os.path.exists(path):
            raise ImportError("matfac_amd: %s not found; build it with `make -C matfac_amd/host`" % path)
        _lib = C.CDLL(path)
        _lib.mfh_synth_create.restype = C.c_void_p
        _lib.mfh_synth_create.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.c_uint32, C.c_double, C.c_double,
                                          C.c_double, C.c_int32, C.c_double, C.c_double, C.c_uint32, C.c_double, C.c_double]
        _lib.mfh_synth_free.argtypes = [C.c_void_p]
        _lib.mfh_synth_shape.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mfh_synth_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mfh_synth_nitems.argtypes = [C.c_void_p]
        _lib.mfh_init_factors.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    return _lib


def make(shape="C2", seed=1, scale=1.0, alpha_u=1.38, alpha_i=2.3, noise=0.5, K0=16, frac=(0.8, 0.1), shard=0,
         r0_u=0.1385, r0_i=0.01):
    """Generate + split with the C++ generator (matfac_amd/host/synth.cpp).  Returns a dict
    with full/train/val/test CSR (all with nUsers rows), nUsers, nItems (datastruct.cpp:91), K.
    User degrees are floor + log-normal (sigma alpha_u, floor r0_u * mean), item popularity is
    Zipf-Mandelbrot (rank + r0_i*nItems)^-alpha_i, both fitted to the MovieLens-20M marginals (see matfac_amd/host/synth.cpp)."""
    s = SHAPES[shape] if isinstance(shape, str) else shape
    lib = _host()
    nnz = max(int(s["nnz"] * scale), s["nU"])
    h = lib.mfh_synth_create(s["nU"], s["nI"], nnz, seed, alpha_u, alpha_i, noise, K0, frac[0], frac[1], shard,
                             r0_u, r0_i * s["nI"])
    if not h:
        raise ValueError("mfh_synth_create rejected the shape %r" % (s,))
    try:
        out = {}
        for which, name in ((0, "full"), (1, "train"), (2, "val"), (3, "test")):
            nr, nc, nz = C.c_int32(), C.c_int32(), C.c_int64()
            lib.mfh_synth_shape(h, which, C.byref(nr), C.byref(nc), C.byref(nz))
            rp = np.empty(nr.value + 1, np.int64)
            ri = np.empty(nz.value, np.int32)
            rv = np.empty(nz.value, np.float32)
            lib.mfh_synth_copy(h, which, rp.ctypes.data_as(C.c_void_p), ri.ctypes.data_as(C.c_void_p),
                               rv.ctypes.data_as(C.c_void_p))
            out[name] = CSR(nr.value, nc.value, rp, ri, rv)
        out["nUsers"] = s["nU"]
        out["nItems"] = int(lib.mfh_synth_nitems(h))
        out["K"] = s.get("K", 0)
        return out
    finally:
        lib.mfh_synth_free(h)


def concat_rows(mats):
    """CSR matrices over the same columns stacked on top of each other (user blocks of one matrix)."""
    ptr = [np.zeros(1, np.int64)]
    off = 0
    for m in mats:
        ptr.append(m.rowptr[1:] + off)
        off += m.nnz
    return CSR(sum(m.nrows for m in mats), mats[0].ncols, np.concatenate(ptr), np.concatenate([m.rowind for m in mats]),
               np.concatenate([m.rowval for m in mats]))


def make_c5_shards(first, count, seed=1, scale=1.0):
    """User shards [first, first + count) of the C5 matrix (8 shards of 1.25 M users x 1 M items, 156 M ratings each before the
    80/10/10 split): what one of N = 8 / count GPUs holds under strong scaling.  Returns train / val CSR and the user count."""
    s = SHAPES["C5"]
    per = dict(nU=s["nU"] // C5_SHARDS, nI=s["nI"], nnz=int(s["nnz"] / 0.8 * scale) // C5_SHARDS, K=s["K"])
    tr, va = [], []
    for j in range(first, first + count):
        d = make(per, seed=seed, shard=j, r0_i=0.002)
        tr.append(d["train"]); va.append(d["val"])
        del d
    return concat_rows(tr), concat_rows(va), per["nU"] * count


def init_factors(seed, nUsers, nItems, K, want_u=True, want_v=True):
    """Model::Model(const Params&) initialisation (model.cpp:2331-2341), host library."""
    lib = _host()
    U = np.empty((nUsers, K), np.float32) if want_u else None
    V = np.empty((nItems, K), np.float32) if want_v else None
    lib.mfh_init_factors(seed, nUsers, nItems, K, U.ctypes.data_as(C.c_void_p) if want_u else None,
                         V.ctypes.data_as(C.c_void_p) if want_v else None)
    return U, V
