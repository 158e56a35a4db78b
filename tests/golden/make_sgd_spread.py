"""Generates tests/golden/sgd_spread_c1.json: the reference's OWN run-to-run spread of SGD results.

The CPU oracle's full training loops (ModelMF::train, modelMF.cpp:4-151, and hogTrain, :1656-1808, both with
Model::isTerminateModel) on the C1-shaped synthetic matrix (943 x 1682, 100 k train ratings, rank 10, reference default
hyper-parameters lr 0.005, ureg = ireg 0.01), same initial factors (seed 1), different std::mt19937 shuffle seeds /
OpenMP thread counts.  What is recorded: best-validation model's test RMSE, its validation RMSE, iterations.  The
sequential rows are deterministic (tests/test_golden.py re-derives them); the Hogwild rows are samples of a race.

    python tests/golden/make_sgd_spread.py          (about a minute on 8 cores)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                       # noqa: E402
from matfac_amd import synth             # noqa: E402
from oracle import binding as orc        # noqa: E402

CFG = dict(shape="C1", K=10, lr=0.005, ureg=0.01, ireg=0.01, maxIter=2000, init_seed=1, data_seed=1)
# a matrix beyond MFX_EXACT_BELOW, where the host classes take the lock-free tiled schedule: 30 000 x 8 000, 2.4 M train
# ratings, rank 32 (python tests/golden/make_sgd_spread.py mid -> sgd_spread_mid.json, about 5 minutes on 8 cores)
CFG_MID = dict(shape=dict(nU=30000, nI=8000, nnz=2_400_000, K=32), K=32, lr=0.005, ureg=0.01, ireg=0.01, maxIter=150,
               init_seed=1, data_seed=1)


def problem(CFG=CFG):
    shape = dict(synth.SHAPES[CFG["shape"]]) if isinstance(CFG["shape"], str) else dict(CFG["shape"])
    shape["nnz"] = int(shape["nnz"] / 0.8)
    return synth.make(shape, seed=CFG["data_seed"])


def run(d, method, train_seed, nthreads, CFG=CFG):
    tr, va, te = d["train"], d["val"], d["test"]
    nU, nI, K = d["nUsers"], d["nItems"], CFG["K"]
    cp, ci, cv = orc.create_col_index(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
    U0, V0 = orc.init_factors(CFG["init_seed"], nU, nI, K)
    r = orc.train(method, U0, V0, (tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval, cp, ci, cv),
                  (va.nrows, va.rowptr, va.rowind, va.rowval), nU, nI, K, CFG["maxIter"], train_seed, CFG["lr"],
                  CFG["ureg"], CFG["ireg"], nthreads=nthreads, dot_mode=orc.DOT_SEQ)
    test, _, _ = orc.rmse(r["Ubest"], r["Vbest"], nU, nI, te.nrows, te.rowptr, te.rowind, te.rowval, r["invU"], r["invI"],
                          orc.DOT_SEQ)
    val, _, _ = orc.rmse(r["Ubest"], r["Vbest"], nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, r["invU"], r["invI"],
                         orc.DOT_SEQ)
    return dict(train_seed=train_seed, threads=nthreads, test_rmse=test, val_rmse=val, iters=int(r["iters"]),
                best_iter=int(r["bestIter"]))


def _sgdpar_row(args):
    mid, seed = args
    cfg = CFG_MID if mid else CFG
    return run(problem(cfg), orc.M_SGDPAR, seed, SGDPAR_PARTS, cfg)


SGDPAR_PARTS = 8      # T of trainSGDPar (= omp_get_max_threads() in the reference): users and items dealt into 8 parts


if __name__ == "__main__":
    mid = len(sys.argv) > 1 and sys.argv[1] == "mid"
    cfg = CFG_MID if mid else CFG
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sgd_spread_%s.json" % ("mid" if mid else "c1"))
    if "sgdpar" in sys.argv[1:]:
        # add the rows of the reference's OWN stratified trainer (ModelMF::trainSGDPar, modelMF.cpp:154-350, via the oracle's
        # orc_strat_create / orc_strat_epoch: T = 8 parts, a fresh random matching per round) to the existing file: the
        # schedule the lock-free tiled kernel is modelled on.  Deterministic per seed (blocks of a round share no rows).
        #     python tests/golden/make_sgd_spread.py [mid] sgdpar
        import multiprocessing as mp
        out = json.load(open(path))
        with mp.Pool(5) as pool:
            out["sgdpar"] = pool.map(_sgdpar_row, [(mid, seed) for seed in range(1, 6)])
        t = np.array([x["test_rmse"] for x in out["sgdpar"]])
        out["sgdpar_parts"] = SGDPAR_PARTS
        out["sgdpar_test_rmse_mean"] = float(t.mean())
        out["sgdpar_test_rmse_std"] = float(t.std(ddof=1))
        json.dump(out, open(path, "w"), indent=1)
        print("sgdpar (T = %d) %.5f +- %.5f   sequential %.5f +- %.5f" % (SGDPAR_PARTS, t.mean(), t.std(ddof=1), out["sequential_test_rmse_mean"],
                                                                    out["sequential_test_rmse_std"]))
        sys.exit(0)
    d = problem(cfg)
    out = dict(config=cfg, train_nnz=d["train"].nnz, sequential=[], hogwild=[])
    for seed in range(1, 6 if mid else 9):
        out["sequential"].append(run(d, orc.M_SGD, seed, 1, cfg))
        print(out["sequential"][-1], flush=True)
    for threads in ((8,) if mid else (8, 64)):
        for seed in ((1, 2) if mid else (1, 2, 3)):
            out["hogwild"].append(run(d, orc.M_HOGSGD, seed, threads, cfg))
            print(out["hogwild"][-1], flush=True)
    t = np.array([x["test_rmse"] for x in out["sequential"]])
    out["sequential_test_rmse_mean"] = float(t.mean())
    out["sequential_test_rmse_std"] = float(t.std(ddof=1))
    h = np.array([x["test_rmse"] for x in out["hogwild"]])
    out["hogwild_test_rmse_mean"] = float(h.mean())
    out["hogwild_test_rmse_std"] = float(h.std(ddof=1))
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sgd_spread_%s.json" % ("mid" if mid else "c1")), "w"),
              indent=1)
    print("sequential %.5f +- %.5f   hogwild %.5f +- %.5f" % (t.mean(), t.std(ddof=1), h.mean(), h.std(ddof=1)))
