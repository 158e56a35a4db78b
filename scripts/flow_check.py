"""host-built vs device-built dataflow queues at C2 (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matfac_amd import Ctx, mfx, synth
shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
d = synth.make(shape, seed=1); tr = d["train"]; nU, nI = d["nUsers"], shape["nI"]
K = 64
U0, V0 = synth.init_factors(1, nU, nI, K)
order = np.random.default_rng(1).permutation(tr.nnz).astype(np.uint64)
out = {}
for name, env in (("host", "1"), ("dev", None)):
    if env: os.environ["MFX_FLOW_HOST"] = env
    else: os.environ.pop("MFX_FLOW_HOST", None)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0); ctx.compute_invalid()
        ctx.sgd_set_order(order)
        ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_LEVELS, order=mfx.ORDER_HOST, arith=mfx.ARITH_REF64)
        rec, off = ctx.debug_flow_queues()
        U, V = ctx.get_factors()
        out[name] = (rec, off, U, V)
a, b = out["host"], out["dev"]
print("qoff equal", np.array_equal(a[1], b[1]), "records equal", np.array_equal(a[0], b[0]))
if not np.array_equal(a[0], b[0]):
    bad = np.nonzero((a[0] != b[0]).any(axis=1))[0]
    print("first bad", bad[:10], len(bad)); print(a[0][bad[:5]]); print(b[0][bad[:5]])
print("factors equal", np.array_equal(a[2], b[2]), np.array_equal(a[3], b[3]))
