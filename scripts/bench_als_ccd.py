"""Timing of the ALS (C3) and CCD++ (C4) paths with HIP events + roofline figures.
Diagnostic companion of bench.py (which measures the headline SGD metric)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import Ctx, mfx, synth

what = os.environ.get("WHAT", "als,ccd").split(",")
if "als" in what:
    K = int(os.environ.get("ALS_K", 64))
    shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
    d = synth.make(shape, seed=1); tr, va = d["train"], d["val"]; nU, nI = d["nUsers"], shape["nI"]
    U0, V0 = synth.init_factors(1, nU, nI, K)
    ctx = Ctx(0)
    ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
    ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
    ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0); ctx.compute_invalid()
    reg = 5.0
    ctx.als_half_sweep(mfx.SIDE_USERS, reg); ctx.als_half_sweep(mfx.SIDE_ITEMS, reg); ctx.synchronize()
    ctx.prof_enable(True); ctx.prof_reset()
    iters = int(os.environ.get("ALS_ITERS", 5))
    t0 = time.perf_counter(); traj = []
    for it in range(iters):
        ctx.als_half_sweep(mfx.SIDE_USERS, reg); ctx.als_half_sweep(mfx.SIDE_ITEMS, reg)
        traj.append(round(ctx.rmse(mfx.MAT_VAL), 5))
    ctx.synchronize(); wall = (time.perf_counter() - t0) / iters
    g_ms, g_n = ctx.prof_get(mfx.K_ALS_GRAM); s_ms, s_n = ctx.prof_get(mfx.K_ALS_SOLVE)
    flops = 2 * tr.nnz * (2 * K * K + 2 * K) + (nU + nI) * (K ** 3 / 3 + 2 * K * K)
    print(json.dumps(dict(path="ALS C3" if K == 64 else "ALS C2 matrix", nnz=tr.nnz, K=K, ms_per_iter_events=(g_ms + s_ms) / iters, wall_ms_per_iter=wall * 1e3,
                          gram_ms=g_ms / iters, reduce_ms=s_ms / iters, tflops=flops / ((g_ms + s_ms) / iters * 1e-3) / 1e12,
                          mfma_peak_tflops=157.3, rating_iters_per_s=tr.nnz / ((g_ms + s_ms) / iters * 1e-3), val_rmse=traj)), flush=True)
    ctx.close()
if "ccd" in what:
    K = int(os.environ.get("CCD_K", 128))
    shape = dict(synth.SHAPES["C4"]); shape["nnz"] = int(shape["nnz"] * float(os.environ.get("CCD_SCALE", 1.0)) / 0.8)
    t0 = time.time(); d = synth.make(shape, seed=1); tr, va = d["train"], d["val"]; nU, nI = d["nUsers"], shape["nI"]
    gen = time.time() - t0
    U0, V0 = synth.init_factors(1, nU, nI, K)
    ctx = Ctx(0)
    ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
    ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
    ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0); ctx.compute_invalid()
    ctx.ccdpp_begin()
    reg = 2.0
    nk = int(os.environ.get("CCD_NK", 16))      # factors timed per outer iteration (scaled to K below)
    for k in range(2): ctx.ccdpp_rank1(k, reg, reg, add_back=False)
    ctx.synchronize(); ctx.prof_enable(True); ctx.prof_reset()
    t0 = time.perf_counter()
    for k in range(nk): ctx.ccdpp_rank1(k, reg, reg, add_back=True)
    ctx.synchronize(); wall = time.perf_counter() - t0
    r_ms, r_n = ctx.prof_get(mfx.K_CCD_ROW); c_ms, c_n = ctx.prof_get(mfx.K_CCD_COL); x_ms, x_n = ctx.prof_get(mfx.K_CCD_RESID)
    per_k = wall / nk
    bytes_per_k = 128 * tr.nnz
    print(json.dumps(dict(path="CCD++ C4", nnz=tr.nnz, K=K, datagen_s=gen, ms_per_factor=per_k * 1e3, s_per_outer_iter=per_k * K,
                          row_pass_ms=r_ms / max(r_n, 1), col_pass_ms=c_ms / max(c_n, 1), resid_ms=x_ms / max(x_n, 1),
                          algorithmic_GBs=bytes_per_k / per_k / 1e9, hbm_peak_GBs=8000,
                          row_pass_GBs=8 * tr.nnz / (r_ms / max(r_n, 1) * 1e-3) / 1e9,
                          resid_GBs=(2 * 12 * tr.nnz / (x_ms / max(x_n, 1) * 1e-3) / 1e9) if x_ms > 0 else None)), flush=True)   # (None: the update rides on the first sweep, MFX_CCD_FUSE)
    ctx.ccdpp_end(); ctx.close()
if "cd" in what:      # trainCCD (a12) on the C2 matrix
    K = int(os.environ.get("CD_K", 64))
    shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
    d = synth.make(shape, seed=1); tr, va = d["train"], d["val"]; nU, nI = d["nUsers"], shape["nI"]
    U0, V0 = synth.init_factors(1, nU, nI, K)
    ctx = Ctx(0)
    ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
    ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
    ctx.set_model(nU, nI, K); ctx.set_factors(U0, V0); ctx.compute_invalid()
    t0 = time.perf_counter(); ctx.ccd_begin(); ctx.synchronize(); begin = time.perf_counter() - t0
    reg = 5.0
    ctx.prof_enable(True)
    traj = []; su = si = 0.0
    iters = 4
    for it in range(iters):
        ctx.prof_reset(); ctx.ccd_sweep(mfx.SIDE_USERS, reg, None, 1, it); ctx.synchronize(); su += ctx.prof_get(mfx.K_CD)[0]
        ctx.prof_reset(); ctx.ccd_sweep(mfx.SIDE_ITEMS, reg, None, 1, it); ctx.synchronize(); si += ctx.prof_get(mfx.K_CD)[0]
        traj.append(round(ctx.rmse(mfx.MAT_VAL), 5))
    # per (rating, k): residual 4 B read + 4 B write, index 4 B, gathered element 4 B
    print(json.dumps(dict(path="CCD (trainCCD) C2", nnz=tr.nnz, K=K, begin_s=begin, user_sweep_ms=su / iters, item_sweep_ms=si / iters,
                          ms_per_iter=(su + si) / iters, rating_factor_updates_per_s=2 * tr.nnz * K / ((su + si) / iters * 1e-3),
                          val_rmse=traj)), flush=True)
    ctx.ccd_end(); ctx.close()
if "svd" in what:     # the SVD initialisation of trainSGDParSVD (a8) on the C2 matrix
    K = int(os.environ.get("SVD_K", 64))
    shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
    d = synth.make(shape, seed=1); tr = d["train"]; nU, nI = d["nUsers"], shape["nI"]
    ctx = Ctx(0)
    ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
    ctx.set_model(nU, nI, K); ctx.compute_invalid()
    ctx.svd_init(1, 10, 1); ctx.synchronize()
    t0 = time.perf_counter(); sig = ctx.svd_init(10, max(10, K // 8), 1); ctx.synchronize(); dt = time.perf_counter() - t0
    U, V = ctx.get_factors()
    # residual of the leading triplets: || R v_k - sigma_k u_k || through scipy
    import scipy.sparse as sp
    R = sp.csr_matrix((tr.rowval, tr.rowind, tr.rowptr), shape=(tr.nrows, nI))
    res = np.linalg.norm(R @ V[:, :8] - U[:, :8] * sig[:8], axis=0) / sig[:8]
    print(json.dumps(dict(path="SVD init C2", nnz=tr.nnz, K=K, seconds=dt, sigma_first=[float(x) for x in sig[:4]], sigma_last=float(sig[-1]),
                          rel_residual_first8=[float(x) for x in res])), flush=True)
    ctx.close()
