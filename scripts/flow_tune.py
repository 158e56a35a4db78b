"""Exact replay at C2 (20 M ratings, K=64, double bracket): tagged dataflow kernel vs the version-counter kernel, bit for bit
against each other, kernel time per epoch; knobs MFX_FLOW_WGS (workgroups per CU).  DESIGN.md 3.1.2."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matfac_amd import Ctx, mfx, synth

name = os.environ.get("SHAPE", "C2")
K = int(os.environ.get("RANK", "64"))
arith = {"ref64": mfx.ARITH_REF64, "f32": mfx.ARITH_F32, "ref64f": mfx.ARITH_REF64F}[os.environ.get("ARITH", "ref64")]
shape = dict(synth.SHAPES[name]); shape["nnz"] = int(shape["nnz"] / 0.8)
d = synth.make(shape, seed=1)
tr = d["train"]
nU, nI = d["nUsers"], shape["nI"]
U0, V0 = synth.init_factors(1, nU, nI, K)
order = np.random.default_rng(1).permutation(tr.nnz).astype(np.uint64)
configs = [c for c in os.environ.get("CONFIGS", "ver:4 tag:4 tag:3 tag:2 tag:1").split()]
ref = None
with Ctx(0) as ctx:
    ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
    ctx.set_model(nU, nI, K); ctx.compute_invalid()
    ctx.sgd_set_order(order)
    ctx.prof_enable(True)
    for cfg in configs:
        kind, wgs = cfg.split(":")
        os.environ["MFX_FLOW_TAGGED"] = "0" if kind == "ver" else "1"
        os.environ["MFX_FLOW_WGS"] = wgs
        best = 1e30
        for rep in range(3):
            ctx.set_factors(U0, V0)
            ctx.prof_reset()
            t0 = time.perf_counter()
            ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_LEVELS, order=mfx.ORDER_HOST, arith=arith)
            ctx.synchronize()
            wall = time.perf_counter() - t0
            ms, _ = ctx.prof_get(mfx.K_SGD)
            best = min(best, ms)
        info, prep = ctx.debug_levels_info()
        U, V = ctx.get_factors()
        if ref is None: ref = (U, V)
        same = np.array_equal(U, ref[0]) and np.array_equal(V, ref[1])
        if os.environ.get("FLOW_STATS") and kind == "tag":
            import ctypes as C
            from matfac_amd import _lib
            ng = int(info[2])
            st = np.zeros((ng, 8), dtype=np.uint64)
            _lib.load().mfx_debug_flow_stats(st.ctypes.data_as(C.c_void_p), C.c_int64(ng))
            st = st.astype(np.float64)
            hot = int(np.argmax(st[:, 0] + (st[:, 2] if os.environ.get("FLOW_STATS") == "wide" else 0)))
            if os.environ.get("FLOW_STATS") == "wide":
                for nm, row in (("hottest queue", st[hot]), ("mean over queues", st.mean(axis=0))):
                    v = max(row[0] + row[2], 1)
                    print("   %s: pole visits %.0f in %.0f blocks, generic visits %.0f, polls %.0f; kernel cycles %.3g = %.0f per visit: in counted waits "
                          "%.0f, in polls %.0f, waiting for the block's records %.0f per visit" % (nm, row[0], row[3], row[2], row[1], row[5], row[5] / v, row[7] / v, row[6] / v, row[4] / v))
                print("   queues by kernel cycles: max %.3g, median %.3g, min %.3g" % (st[:, 5].max(), np.median(st[:, 5]), st[:, 5].min()))
                continue
            for nm, row in (("hottest queue", st[hot]), ("mean over queues", st.mean(axis=0))):
                v = max(row[0], 1)
                print("   %s: visits %.0f, waited for their row %.0f (probes %.0f), owned-row switches %.0f (table loads %.0f), "
                      "loop cycles %.3g = %.0f per visit, of which waiting for rows %.0f per visit"
                      % (nm, row[0], row[1], row[2], row[3], row[4], row[5], row[5] / v, row[6] / v))
        print("%s K=%d %s wgs=%s: kernels %.2f ms = %.1f M upd/s (longest queue %d, %d groups), prep %.0f ms, call %.0f ms; same bits as first: %s"
              % (name, K, kind, wgs, best, tr.nnz / best / 1e3, info[1], info[2], prep, wall * 1e3, same), flush=True)
