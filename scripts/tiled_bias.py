"""How far the lock-free tiled schedule lands from the reference's own trainers, and what moves it: the problems of
tests/golden/sgd_spread_{c1,mid}.json trained through the host class (libmfhost.so) with the number of waves per workgroup that
take part in a slot (MFX_SGD_WAVES: ratings in flight on one owned row).  Reads the committed fixture only; nothing of the oracle
runs here.   WHICH=mid|c1 METHODS=sgd,sgdpar WAVES=16,8,4,2,1 python scripts/tiled_bias.py"""
import ctypes as C
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matfac_amd import synth


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
    try:
        rc = lib.mfh_train(method.encode(), C.c_int32(tr.nrows), P(tr.rowptr), P(tr.rowind), P(tr.rowval), C.c_int32(tr.ncols),
                           P(va.rowptr), P(va.rowind), P(va.rowval), C.c_int32(va.ncols), P(te.rowptr), P(te.rowind), P(te.rowval),
                           C.c_int32(te.ncols), C.c_int32(K), C.c_int32(maxIter), C.c_int32(seed), C.c_float(lr), C.c_float(ureg),
                           C.c_float(ireg), None, P(bufs[0]), P(bufs[1]), P(bufs[2]), P(bufs[3]), P(stats), P(invU), P(invI))
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    assert rc == 0
    return dict(test=stats[1], val=stats[2], iters=int(stats[7]), loop_s=stats[6])


which = os.environ.get("WHICH", "mid")
f = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sgd_spread_%s.json" % which)))
cfg = f["config"]
shape = dict(synth.SHAPES[cfg["shape"]]) if isinstance(cfg["shape"], str) else dict(cfg["shape"])
shape["nnz"] = int(shape["nnz"] / 0.8)
d = synth.make(shape, seed=cfg["data_seed"])
seq = np.array([x["test_rmse"] for x in f["sequential"]]); par = np.array([x["test_rmse"] for x in f["sgdpar"]])
print("%s reference: sequential %.5f +- %.5f | trainSGDPar (T=8) %.5f +- %.5f | hogwild %s"
      % (which, seq.mean(), seq.std(ddof=1), par.mean(), par.std(ddof=1), [round(x["test_rmse"], 5) for x in f["hogwild"]]), flush=True)
args = (cfg["K"], cfg["maxIter"], 1, cfg["lr"], cfg["ureg"], cfg["ireg"])
for method in os.environ.get("METHODS", "sgd").split(","):
    for waves in os.environ.get("WAVES", "16,8,4,2,1").split(","):
        h = host_train(method, d, *args, env={"MFX_EXACT": "0", "MFX_SGD_WAVES": waves})
        print("%s tiled, %2s waves per slot: test RMSE %.5f (val %.5f, %d iterations, %.2f s) = %+.1f sigma of sequential, %+.1f sigma of trainSGDPar"
              % (method, waves, h["test"], h["val"], h["iters"], h["loop_s"], (h["test"] - seq.mean()) / seq.std(ddof=1),
                 (h["test"] - par.mean()) / par.std(ddof=1)), flush=True)
for method, rows in (("sgd", f["sequential"]), ("sgdpar", f["sgdpar"])):
    h = host_train(method, d, *args, env={"MFX_SGDPAR_PARTS": str(f.get("sgdpar_parts", 8))})
    print("%s default path (order replay): test RMSE %.7f vs fixture seed 1 %.7f; %d iterations (fixture %d); %.2f s"
          % (method, h["test"], rows[0]["test_rmse"], h["iters"], rows[0]["iters"], h["loop_s"]), flush=True)
