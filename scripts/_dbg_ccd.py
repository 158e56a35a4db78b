import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import Ctx, mfx, synth
from oracle import binding as orc
from tests.util import load_ctx
from tests.test_ccd_gpu import ulp_diff, _setup
K = 8
d, tr, (cp, ci, cv), U0, V0 = _setup(1500, 400, 60000, K, seed=K)
nU, nI = d["nUsers"], d["nItems"]
for inner in (1, 2, 5):
    Uo, Vo = U0.copy(), V0.copy(); Uo[:] = 0
    rr, rc = tr.rowval.copy(), cv.copy()
    with Ctx(0) as ctx:
        invU, invI = load_ctx(ctx, d, K, U0, V0)
        ctx.ccdpp_begin()
        ctx.ccdpp_rank1(0, 0.3, 0.2, add_back=False, inner=inner, freq_thresh=-1.0)
        orc.ccdpp_rank1(0, Uo, Vo, nU, nI, tr.ncols, tr.rowptr, tr.rowind, rr, cp, ci, rc, invU, invI, 0.3, 0.2, False, inner, -1.0, nthreads=4)
        U, V = ctx.get_factors()
        ctx.ccdpp_end()
    du, dv = ulp_diff(U[:, 0], Uo[:, 0]), ulp_diff(V[:, 0], Vo[:, 0])
    rl = np.diff(tr.rowptr); cl = np.diff(cp)
    print("inner", inner, "U ulp: max", du.max(), "hist", np.bincount(np.minimum(du, 5)), "V ulp: max", dv.max(), "hist", np.bincount(np.minimum(dv, 5)))
    wu = np.argsort(-du)[:6]; wv = np.argsort(-dv)[:6]
    print("  worst rows", [(int(i), int(du[i]), int(rl[i]), float(U[i, 0]), float(Uo[i, 0])) for i in wu])
    print("  worst cols", [(int(i), int(dv[i]), int(cl[i]), float(V[i, 0]), float(Vo[i, 0])) for i in wv])
