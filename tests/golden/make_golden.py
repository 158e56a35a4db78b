"""Generates the golden fixtures in this directory with the CPU oracle (oracle/oracle.cpp).

The reference ships no golden vectors and cannot be built here, so these
are *restatement-generated* goldens (SURVEY.md 8c, item iv): trajectories of objective and
validation RMSE + final factors for a few iterations of each restated trainer, single-threaded,
on (1) a hand-written 6x5 / 20-rating matrix and (2) the C1-shaped synthetic matrix.  They guard
the oracle against regressions and give the GPU tests fixed targets.

    python tests/golden/make_golden.py        # rewrites tiny_6x5.npz and c1_shape.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from matfac_amd import synth  # noqa: E402
from oracle import binding as orc  # noqa: E402

METHODS = {"sgd": orc.M_SGD, "sgdu": orc.M_SGDU, "als": orc.M_ALS, "ccdpp": orc.M_CCDPP, "ccdpp_fa": orc.M_CCDPP_FA,
           "ccd": orc.M_CCD}


def tiny():
    # 6 users x 5 items, 20 train ratings; every user and item rated at least once
    rows = [[(0, 4.0), (1, 3.5), (3, 1.0)], [(0, 5.0), (2, 2.0), (4, 4.5)], [(1, 3.0), (2, 2.5), (3, 1.5), (4, 4.0)],
            [(0, 4.5), (3, 0.5), (4, 5.0)], [(1, 2.0), (2, 3.0), (3, 2.0)], [(0, 3.5), (1, 4.0), (2, 1.0), (4, 3.0)]]
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)
    ri = np.array([c for r in rows for c, _ in r], np.int32)
    rv = np.array([v for r in rows for _, v in r], np.float32)
    train = synth.CSR(6, 5, rp, ri, rv)
    vrows = [[(2, 2.0)], [(1, 3.0)], [(0, 4.0)], [(1, 2.5)], [(4, 3.5)], [(3, 1.0)]]
    vp = np.cumsum([0] + [len(r) for r in vrows]).astype(np.int64)
    val = synth.CSR(6, 5, vp, np.array([c for r in vrows for c, _ in r], np.int32),
                    np.array([v for r in vrows for _, v in r], np.float32))
    return train, val


def run_all(train, val, nU, nI, K, iters, lr, reg, seed=1):
    cp, ci, cv = orc.create_col_index(train.nrows, train.ncols, train.rowptr, train.rowind, train.rowval)
    out = {}
    for name, m in METHODS.items():
        U0, V0 = orc.init_factors(seed, nU, nI, K)
        r = orc.train(m, U0, V0, (train.nrows, train.ncols, train.rowptr, train.rowind, train.rowval, cp, ci, cv),
                      (val.nrows, val.rowptr, val.rowind, val.rowval), nU, nI, K, iters, seed, lr,
                      reg if m in (orc.M_SGD, orc.M_SGDU) else max(reg, 0.5),
                      reg if m in (orc.M_SGD, orc.M_SGDU) else max(reg, 0.5), nthreads=1, dot_mode=orc.DOT_SEQ)
        out[name] = r
    return out


def main():
    train, val = tiny()
    res = run_all(train, val, 6, 5, 3, 5, 0.05, 0.05)
    blob = dict(tr_rowptr=train.rowptr, tr_rowind=train.rowind, tr_rowval=train.rowval, va_rowptr=val.rowptr,
                va_rowind=val.rowind, va_rowval=val.rowval)
    for name, r in res.items():
        blob[name + "_obj"], blob[name + "_val"] = r["obj"], r["val"]
        blob[name + "_U"], blob[name + "_V"] = r["U"], r["V"]
        blob[name + "_Ubest"], blob[name + "_Vbest"] = r["Ubest"], r["Vbest"]
    np.savez(os.path.join(HERE, "tiny_6x5.npz"), **blob)

    d = synth.make("C1", seed=1)
    res = run_all(d["train"], d["val"], d["nUsers"], d["nItems"], 10, 4, 0.005, 0.01)
    blob = dict(train_nnz=d["train"].nnz, val_nnz=d["val"].nnz, nItems=d["nItems"],
                train_checksum=float(np.dot(d["train"].rowval.astype(np.float64), np.arange(d["train"].nnz) % 97)))
    for name, r in res.items():
        blob[name + "_obj"], blob[name + "_val"] = r["obj"], r["val"]
        blob[name + "_Ubest_head"] = r["Ubest"][:8]                 # small: first rows only
        blob[name + "_Vbest_head"] = r["Vbest"][:8]
        blob[name + "_Unorm"] = float(np.sqrt((r["Ubest"].astype(np.float64) ** 2).sum()))
        blob[name + "_Vnorm"] = float(np.sqrt((r["Vbest"].astype(np.float64) ** 2).sum()))
    np.savez(os.path.join(HERE, "c1_shape.npz"), **blob)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
