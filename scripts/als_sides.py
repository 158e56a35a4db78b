"""Per-side timing of the ALS half sweeps on the C2 matrix (diagnostic): users / items, with MFX_ALS_NOSOLVE
set in the environment the factorisation is skipped (accumulation only)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matfac_amd import Ctx, mfx, synth

K = int(os.environ.get("ALS_K", 64))
shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
d = synth.make(shape, seed=1); tr = d["train"]; nU, nI = d["nUsers"], shape["nI"]
U0, V0 = synth.init_factors(1, nU, nI, K)
ctx = Ctx(0)
ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0); ctx.compute_invalid()
reg = 5.0
out = {"K": K, "nosolve": os.environ.get("MFX_ALS_NOSOLVE") is not None}
for side, name in ((mfx.SIDE_USERS, "users"), (mfx.SIDE_ITEMS, "items")):
    ctx.als_half_sweep(side, reg); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.als_half_sweep(side, reg)
    ctx.synchronize()
    out[name + "_ms"] = (time.perf_counter() - t0) / 5 * 1e3
print(json.dumps(out))
