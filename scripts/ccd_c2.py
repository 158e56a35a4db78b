"""CCD++ on the C2 matrix (20 M ratings, K=64): time per rank-one step and how much of it is launch latency."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import Ctx, mfx, synth
K = 64
shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
d = synth.make(shape, seed=1); tr = d["train"]; nU, nI = d["nUsers"], shape["nI"]
U0, V0 = synth.init_factors(1, nU, nI, K)
ctx = Ctx(0)
ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0); ctx.compute_invalid()
ctx.ccdpp_begin()
for k in range(4): ctx.ccdpp_rank1(k, 2.0, 2.0, add_back=False)
ctx.synchronize(); ctx.prof_enable(True); ctx.prof_reset()
t0 = time.perf_counter(); n = 32
for k in range(n): ctx.ccdpp_rank1(k, 2.0, 2.0, add_back=True)
ctx.synchronize(); wall = (time.perf_counter() - t0) / n
r = ctx.prof_get(mfx.K_CCD_ROW); c = ctx.prof_get(mfx.K_CCD_COL); x = ctx.prof_get(mfx.K_CCD_RESID)
kern = (r[0] + c[0] + x[0]) / n
print(json.dumps(dict(nnz=int(tr.nnz), K=K, ms_per_factor_wall=round(wall * 1e3, 3), ms_per_factor_profiled_kernels=round(kern, 3),
                      row_pass_ms=round(r[0] / max(r[1], 1), 4), col_pass_ms=round(c[0] / max(c[1], 1), 4), resid_ms=round(x[0] / max(x[1], 1), 4),
                      algorithmic_GBs=round(128 * tr.nnz / wall / 1e9, 1))))
ctx.ccdpp_end()
