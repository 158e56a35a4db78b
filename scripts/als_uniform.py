"""ALS user half sweep on rows of equal length (diagnostic): how much of the accumulation time is per-row overhead.
ROWLEN=1000 (default) ratings per row, ~20 M ratings in total, 26 744 items, K=64."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import Ctx, mfx, synth

K, nI = 64, 26744
out = []
for rowlen in [int(x) for x in os.environ.get("ROWLEN", "1000,144,32").split(",")]:
    nU = 20_000_000 // rowlen
    rng = np.random.default_rng(1)
    rowptr = np.arange(nU + 1, dtype=np.int64) * rowlen
    rowind = np.sort(rng.integers(0, nI, size=(nU, rowlen), dtype=np.int32), axis=1).reshape(-1)
    rowval = rng.integers(1, 6, size=nU * rowlen).astype(np.float32)
    U0, V0 = synth.init_factors(1, nU, nI, K)
    ctx = Ctx(0)
    ctx.set_csr(mfx.MAT_TRAIN, nU, nI, rowptr, rowind, rowval)
    ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0); ctx.compute_invalid()
    ctx.als_half_sweep(mfx.SIDE_USERS, 5.0); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.als_half_sweep(mfx.SIDE_USERS, 5.0)
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    out.append(dict(rowlen=rowlen, rows=nU, nosolve=os.environ.get("MFX_ALS_NOSOLVE") is not None, users_ms=round(ms, 3),
                    mfma_tflops=round(nU * rowlen * 3 * 4096 / 2 / ms / 1e9, 1)))
    ctx.close()
print(json.dumps(out))
