"""MFX_SGD_LEVELS vs MFX_SGD_SERIAL: updates/s of the two bit-exact replay paths.
C1 (100 k ratings, K=10) whole; C2 (20 M, K=64): the level schedule on the whole list, the serial kernel on a prefix."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matfac_amd import Ctx, mfx, synth

for name, K, serial_n in (("C1", 10, None), ("C2", 64, 400_000)):
    shape = dict(synth.SHAPES[name]); shape["nnz"] = int(shape["nnz"] / 0.8)
    d = synth.make(shape, seed=1)
    tr = d["train"]
    nU, nI = d["nUsers"], shape["nI"]
    U0, V0 = synth.init_factors(1, nU, nI, K)
    rng = np.random.default_rng(1)
    order = rng.permutation(tr.nnz).astype(np.uint64)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0); ctx.compute_invalid()
        ctx.sgd_set_order(order)
        ctx.prof_enable(True)
        for rep in range(2):
            ctx.prof_reset()
            t0 = time.perf_counter()
            ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_LEVELS, order=mfx.ORDER_HOST, arith=mfx.ARITH_REF64)
            ctx.synchronize()
            wall = time.perf_counter() - t0
            ms, _ = ctx.prof_get(mfx.K_SGD)
            info, prep = ctx.debug_levels_info()
        print("%s exact replay [sched %d: %d / %d], kernels %.2f ms = %.1f M upd/s; host prep %.0f ms; call %.0f ms = %.1f M upd/s"
              % (name, info[0], info[1], info[2], ms, tr.nnz / ms / 1e3, prep, wall * 1e3, tr.nnz / wall / 1e6), flush=True)
        n = serial_n or tr.nnz
        ctx.prof_reset()
        ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_SERIAL, order=mfx.ORDER_HOST, arith=mfx.ARITH_REF64, first=0, count=n)
        ctx.synchronize()
        ms2, _ = ctx.prof_get(mfx.K_SGD)
        print("%s serial kernel: %d ratings in %.1f ms = %.2f M upd/s  -> level schedule %.0fx (kernels), %.0fx (whole call)"
              % (name, n, ms2, n / ms2 / 1e3, (tr.nnz / ms) / (n / ms2), (tr.nnz / (wall * 1e3)) / (n / ms2)), flush=True)
