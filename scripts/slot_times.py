"""Diagnostic (library built with -DMFX_EXP=8, scripts/slot_times.sh): when does every workgroup of a round launch of the tiled SGD
kernel finish?  Prints, per round and XCD, the workgroups' busy time (100 MHz wall clock) against the launch's span: a tile whose
longest slot (a popular item's pole) outlasts the others shows up as one late workgroup and an idle XCD."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matfac_amd import Ctx, mfx, synth

name = os.environ.get("SHAPE", "C2")
shape = dict(synth.SHAPES[name]); shape["nnz"] = int(shape["nnz"] / 0.8)
K = shape["K"]
d = synth.make(shape, seed=1)
tr = d["train"]
nU, nI = d["nUsers"], shape["nI"]
U0, V0 = synth.init_factors(1, nU, nI, K)
with Ctx(0) as ctx:
    ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
    ctx.set_model(nU, nI, K); ctx.compute_invalid()
    ctx.set_factors(U0, V0)
    for ep in range(3):
        ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=ep, flags=mfx.SGD_F_COUNT_VISITS)
    ctx.synchronize()
    c = ctx.debug_visit_counts()
blocks = int(os.environ.get("BLOCKS", "512"))
t = c[: 8 * blocks * 4].reshape(8, blocks, 4).astype(np.int64)
for r in range(8):
    t0, t1 = t[r, :, 0], t[r, :, 1]
    span = (t1.max() - t0.min()) / 100.0
    busy = (t1 - t0) / 100.0
    xcc = t[r, :, 2] & 255
    print("round %d: span %.1f us, workgroup busy mean %.1f us (%.0f %% of the span)" % (r, span, busy.mean(), 100 * busy.mean() / span))
    for x in range(8):
        m = xcc == x
        if not m.any(): continue
        e = (t1[m] - t0.min()) / 100.0
        sl = t[r, m, 2] >> 8
        print("   XCD %d: %3d workgroups, last finishes at %.1f us, median %.1f us, first %.1f us; slots per workgroup %d..%d, ratings %d..%d"
              % (x, m.sum(), e.max(), np.median(e), e.min(), sl.min(), sl.max(), t[r, m, 3].min(), t[r, m, 3].max()))
