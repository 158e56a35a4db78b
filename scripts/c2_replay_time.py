"""Exact replay (MFX_SGD_LEVELS) at the C2 shape: kernel and whole-call time per epoch with hybrid ownership (default) and with the item
rows owned throughout (MFX_FLOW_HYBRID=0), rank K (env K, default 64).  Diagnostic companion of bench.py's exact_replay record."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import Ctx, mfx, synth

K = int(os.environ.get("K", 64))
shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
d = synth.make(shape, seed=1); tr = d["train"]; nU, nI = d["nUsers"], shape["nI"]
U0, V0 = synth.init_factors(1, nU, nI, K)
rng = np.random.default_rng(1)
for mode in ("", "0", "host"):
    if mode: os.environ["MFX_FLOW_HYBRID"] = mode
    else: os.environ.pop("MFX_FLOW_HYBRID", None)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0)
        ctx.prof_enable(True)
        ks, ws = [], []
        for ep in range(4):
            ctx.sgd_set_order(rng.permutation(tr.nnz).astype(np.uint64)); ctx.synchronize(); ctx.prof_reset()
            t0 = time.perf_counter()
            ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_LEVELS, order=mfx.ORDER_HOST, arith=mfx.ARITH_REF64)
            ctx.synchronize(); w = time.perf_counter() - t0
            if ep: ks.append(ctx.prof_get(mfx.K_SGD)[0]); ws.append(w * 1e3)
        info, prep = ctx.debug_levels_info()
        print("K %d MFX_FLOW_HYBRID=%-5s kernel %.2f ms  call %.2f ms  longest queue %d  queues %d  prep %.2f ms" % (K, mode or "(def)", min(ks), min(ws), info[1], info[2], prep), flush=True)
