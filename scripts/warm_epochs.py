"""What the first epochs of the lock-free tiled schedule are run as, and where the converged model lands: the problems of
tests/golden/sgd_spread_{c1,mid}.json through the host class at the reference's default rate, MFX_EXACT=0, with
MFX_TILED_WARM = 0 (tiles from the first epoch: round 3) / hog:n / flow:n.  Counts the "Found nan" lines of the run and reports
the distance to the fixture's hogwild rows.   WHICH=mid,c1 WARM=0,hog:1,flow:1 METHODS=hogsgd python scripts/warm_epochs.py"""
import ctypes as C
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matfac_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def host_train(method, d, K, maxIter, seed, lr, ureg, ireg, env):
    lib = synth._host()
    tr, va, te = d["train"], d["val"], d["test"]
    nU, nI = d["nUsers"], d["nItems"]
    bufs = [np.empty((nU, K), np.float32), np.empty((nI, K), np.float32), np.empty((nU, K), np.float32), np.empty((nI, K), np.float32)]
    stats = np.zeros(8)
    invU, invI = np.empty(nU, np.uint8), np.empty(nI, np.uint8)
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    os.environ["MFX_NO_SAVE"] = "1"
    sys.stdout.flush()
    tmp = tempfile.TemporaryFile(mode="w+b")
    saved = os.dup(1)
    os.dup2(tmp.fileno(), 1)
    try:
        rc = lib.mfh_train(method.encode(), C.c_int32(tr.nrows), P(tr.rowptr), P(tr.rowind), P(tr.rowval), C.c_int32(tr.ncols),
                           P(va.rowptr), P(va.rowind), P(va.rowval), C.c_int32(va.ncols), P(te.rowptr), P(te.rowind), P(te.rowval),
                           C.c_int32(te.ncols), C.c_int32(K), C.c_int32(maxIter), C.c_int32(seed), C.c_float(lr), C.c_float(ureg),
                           C.c_float(ireg), None, P(bufs[0]), P(bufs[1]), P(bufs[2]), P(bufs[3]), P(stats), P(invU), P(invI))
    finally:
        C.CDLL(None).fflush(None)
        os.dup2(saved, 1)
        os.close(saved)
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    tmp.seek(0)
    log = tmp.read().decode(errors="replace")
    assert rc == 0
    return dict(test=stats[1], val=stats[2], lr=stats[3], iters=int(stats[7]), loop_s=stats[6], nans=log.count("Found nan"))


for which in os.environ.get("WHICH", "mid,c1").split(","):
    f = json.load(open(os.path.join(ROOT, "tests", "golden", "sgd_spread_%s.json" % which)))
    cfg = f["config"]
    shape = dict(synth.SHAPES[cfg["shape"]]) if isinstance(cfg["shape"], str) else dict(cfg["shape"])
    shape["nnz"] = int(shape["nnz"] / 0.8)
    d = synth.make(shape, seed=cfg["data_seed"])
    seq = np.array([x["test_rmse"] for x in f["sequential"]]); hog = np.array([x["test_rmse"] for x in f["hogwild"]])
    print("%s reference: sequential %.5f +- %.5f | hogwild %.5f [%.5f, %.5f] best_iter %s"
          % (which, seq.mean(), seq.std(ddof=1), hog.mean(), hog.min(), hog.max(), [x["best_iter"] for x in f["hogwild"]]), flush=True)
    lr = float(os.environ.get("LR", cfg["lr"]))
    for method in os.environ.get("METHODS", "hogsgd").split(","):
        for warm in os.environ.get("WARM", "0,hog:1,flow:1").split(","):
            for seed in [int(x) for x in os.environ.get("SEEDS", "1").split(",")]:
                env = {"MFX_EXACT": "0", "MFX_TILED_WARM": warm}
                if os.environ.get("WAVES"): env["MFX_SGD_WAVES"] = os.environ["WAVES"]
                h = host_train(method, d, cfg["K"], cfg["maxIter"], seed, lr, cfg["ureg"], cfg["ireg"], env)
                print("%s %s warm=%-7s seed %d: test RMSE %.5f (val %.5f, %d iterations, final lr %.5g, %d x 'Found nan', %.2f s): %+.2e from the hogwild mean"
                      % (which, method, warm, seed, h["test"], h["val"], h["iters"], h["lr"], h["nans"], h["loop_s"], h["test"] - hog.mean()), flush=True)
