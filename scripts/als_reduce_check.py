"""ALS item half-sweep on a matrix whose items have more than 1024 ratings (several segments per row: als_reduce_kernel) against the
oracle, row by row -- the case in which an unpadded v_permlane32_swap hazard showed."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matfac_amd import Ctx, mfx, synth
from oracle import binding as orc
from tests.util import small, load_ctx
K = 64
d = small(nU=6000, nI=300, nnz=400000, K=K, seed=3)
tr = d["train"]; nUu, nIi = d["nUsers"], d["nItems"]
U0, V0 = orc.init_factors(1, nUu, nIi, K); U0 *= 30; V0 *= 30
cp, ci, cv = orc.create_col_index(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
with Ctx(0) as ctx:
    invU, invI = load_ctx(ctx, d, K, U0, V0)
    ctx.als_half_sweep(mfx.SIDE_ITEMS, 5.0)
    U, V = ctx.get_factors()
Vo = V0.copy()
orc.als_half(1, Vo, U0, min(nIi, tr.ncols), cp, ci, cv, invI, 5.0)
err = np.abs(V - Vo).max(axis=1); deg = np.diff(cp)
bad = np.nonzero(err > 1e-3 * np.abs(Vo).max())[0]
print(os.environ.get("TAG", "default lib"), ": rows wrong %d (rows > 1024: %d), max err %.3g" % (bad.size, (deg > 1024).sum(), err.max()))
