#!/usr/bin/env python3
"""bench.py -- rating-updates/sec of the rank-64 SGD hot path on synthetic ML-20M-shape CSR.

    python bench.py --gpus N --steps K --warmup W

One "step" = one SGD epoch (device-side reshuffle + Hogwild update kernel, modelMF.cpp:1739-1767)
over the rank's train ratings.  N > 1: one process per GPU (torch.distributed.run), the rating
matrix is sharded by user-row blocks (every rank owns a full ML-20M-shape block of users over the
SAME item catalogue: weak scaling) and the item-factor replicas are averaged with one RCCL all-reduce
after every local epoch.  Rank 0 prints ONE JSON line (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--rank-k", type=int, default=64, dest="K")
    ap.add_argument("--scale", type=float, default=1.0, help="scale nnz (debug only)")
    ap.add_argument("--arith", default="f32", choices=["f32", "ref64"])
    ap.add_argument("--mode", default="tiled", choices=["tiled", "hogwild"])
    ap.add_argument("--blocks", type=int, default=0, help="workgroups in flight (0 = library heuristic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=float, default=1.0, help="epochs of the CPU baseline sample")
    args = ap.parse_args()

    # everything but the final JSON line goes to stderr (RCCL prints its version banner on stdout)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    if N != world:
        if world == 1 and N > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % N)
        N = world

    import numpy as np

    dist = None
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1"   # rehearse the N > 1 code path with one rank
    # BENCH_COMM=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks -- the ranks share the
    # visible GPUs and the exchange goes through mfx_comm_init_external + gloo instead of RCCL (never the driver's mode)
    use_gloo = os.environ.get("BENCH_COMM") == "gloo"
    if N > 1 or force_dist:
        import torch
        import torch.distributed as dist
        if use_gloo:
            local_rank = local_rank % max(1, torch.cuda.device_count())
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from matfac_amd import Ctx, mfx, synth

    K = args.K
    shape = dict(synth.SHAPES[args.workload])
    # the NAMED shape is the training matrix: generate train/val/test = 80/10/10 around it
    shape["nnz"] = int(shape["nnz"] * args.scale / 0.8)
    t0 = time.time()
    d = synth.make(shape, seed=1, shard=rank)
    tr, va = d["train"], d["val"]
    nU, nI = d["nUsers"], shape["nI"]
    gen_s = time.time() - t0

    ctx = Ctx(local_rank)
    cp, ci, cv = tr.col_view() if False else (None, None, None)
    ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval, cp, ci, cv)
    ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
    ctx.set_model(nU, nI, K)
    U0, _ = synth.init_factors(1 + rank, nU, nI, K, want_v=False)   # U shard: per-rank stream
    _, V0 = synth.init_factors(1, 1, nI, K, want_u=False)            # V replica: identical everywhere
    ctx.set_factors(U0, V0)
    ctx.compute_invalid()
    exchange = "RCCL item-factor all-reduce"
    if N > 1 or force_dist:
        import torch
        if use_gloo:
            exchange = "gloo all-reduce staged through the host (rehearsal)"
            ctx.comm_init_external(N, rank, lambda a: dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM))
        else:
            # the library's own RCCL communicator (device buffers, on its stream).  Every rank reports whether it came
            # up; if any did not, all of them fall back to the host-staged external reducer over torch's communicator,
            # so that the run still completes (slower, and named as such in config.exchange).
            ok = 1
            uid = [None]
            if rank == 0:
                try:
                    uid = [Ctx.comm_unique_id()]
                except Exception as e:                               # noqa: BLE001
                    print("rank 0: mfx_comm_unique_id failed: %s" % e, file=sys.stderr)
            dist.broadcast_object_list(uid, src=0)
            try:
                if uid[0] is None:
                    raise RuntimeError("no unique id")
                if os.environ.get("BENCH_BREAK_RCCL") == "1":        # rehearsal of the fallback below
                    raise RuntimeError("BENCH_BREAK_RCCL=1")
                ctx.comm_init(N, rank, uid[0])
            except Exception as e:                                   # noqa: BLE001 -- reported, then agreed on below
                print("rank %d: mfx_comm_init failed: %s" % (rank, e), file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 0:
                exchange = "torch.distributed all-reduce staged through the host (library RCCL init failed)"
                if ok:
                    ctx.comm_destroy()

                def staged(a):
                    t = torch.from_numpy(a).cuda()
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    a[...] = t.cpu().numpy()
                ctx.comm_init_external(N, rank, staged)
        ctx.comm_mark_synced()

    # main.cpp:29-31 defaults are learnrate 0.005, ureg = ireg = 0.01.  On ML-20M-skewed data the
    # reference's sequential loop diverges at 0.005 in its first epoch (NaN) and its own guard
    # (model.cpp:1486-1498) halves the rate: 0.0025 is where the CPU reference actually trains.
    lr, ureg, ireg = 0.0025, 0.01, 0.01
    arith = mfx.ARITH_F32 if args.arith == "f32" else mfx.ARITH_REF64
    nnz = tr.nnz
    mode = mfx.SGD_TILED if args.mode == "tiled" else mfx.SGD_HOGWILD

    def step(ep):
        ctx.sgd_epoch(lr, ureg, ireg, mode=mode, order=mfx.ORDER_DEVICE, arith=arith, seed=1, epoch=ep,
                      blocks=args.blocks)
        if N > 1 or force_dist:
            ctx.allreduce_item_factors(mfx.REDUCE_AVERAGE)   # replicas of V averaged (summed deltas overshoot: DESIGN.md 3.1)

    def barrier():
        ctx.synchronize()
        if N > 1 or force_dist:
            import torch
            dist.barrier()
            if not use_gloo:
                torch.cuda.synchronize()

    for ep in range(args.warmup):
        step(ep)
    ctx.prof_enable(True)
    ctx.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for ep in range(args.warmup, args.warmup + args.steps):
        step(ep)
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    sgd_ms, sgd_launches = ctx.prof_get(mfx.K_SGD)
    perm_ms, _ = ctx.prof_get(mfx.K_PERMUTE)
    val_rmse = ctx.rmse(mfx.MAT_VAL)
    tr_rmse = ctx.rmse(mfx.MAT_TRAIN)

    total_nnz = nnz
    if N > 1 or force_dist:
        import torch
        t = torch.tensor([elapsed, float(nnz)], dtype=torch.float64, device="cpu" if use_gloo else "cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_nnz = int(t[1])

    out = None
    if rank == 0:
        value = total_nnz * args.steps / elapsed
        avg_ms = sgd_ms / max(1, sgd_launches)
        launches_per_step = max(1, sgd_launches // args.steps)   # tiled: 8 round launches per epoch
        # B_sgd(K) = 16K+12 bytes per update x updates one launch processes
        alg_bytes = (16 * K + 12) * nnz / launches_per_step
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        out = {
            "metric": "rating-updates/sec @ rank=%d" % K, "value": value, "unit": "updates/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s synthetic CSR %dx%d per GPU, train nnz=%d per GPU, rank=%d, "
                                   "%s Hogwild SGD epoch (device reshuffle + update kernel%s)"
                                   % (args.workload, {"C1": "ML-100K-shape", "C2": "ML-20M-shape", "C4": "Netflix-shape"}.get(args.workload, ""),
                                      nU, nI, nnz, K, "XCD-tiled" if mode == mfx.SGD_TILED else "flat",
                                      ", " + exchange if N > 1 or force_dist else ""),
                       "learnrate": lr, "ureg": ureg, "ireg": ireg, "arith": args.arith,
                       "parallelism": "user-block x%d" % N},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(mode),
                         "kernel": "sgd_slots_kernel" if mode == mfx.SGD_TILED else "sgd_hogwild_kernel",
                         "avg_launch_ms": avg_ms, "launches": sgd_launches, "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_launch": alg_bytes},
            "permute_ms_per_step": perm_ms / max(1, args.steps),
            "val_rmse_after": val_rmse, "train_rmse_after": tr_rmse,
            "datagen_s": gen_s,
        }
        if N == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(np, tr, nU, nI, K, lr, ureg, ireg, args.cpu_sample)
    if N > 1 or force_dist:
        ctx.comm_destroy()
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


def pmc_traffic(mode):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/r01_pmc_summary.json: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, KB -> bytes).
    rocprofv3 cannot run inside this process; None when no summary is committed for the kernel."""
    from matfac_amd import mfx
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
        key = "sgd_slots_kernel" if mode == mfx.SGD_TILED else "sgd_hogwild_kernel"
        return d[key]["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(np, tr, nU, nI, K, lr, ureg, ireg, epochs):
    """The oracle's OpenMP Hogwild loop (restatement of modelMF.cpp:1746-1767) on the host cores:
    reported baseline only.  Sample = `epochs` epoch(s) over the first part of the same train list."""
    from oracle import binding as orc
    from matfac_amd import synth
    threads = orc.max_threads()
    n = int(tr.nnz * min(1.0, epochs))
    rng = np.random.default_rng(1)
    sel = rng.permutation(tr.nnz)[:n]
    u = tr.rowids()[sel].astype(np.int32)
    i = tr.rowind[sel].astype(np.int32)
    r = tr.rowval[sel].astype(np.float32)
    U, V = synth.init_factors(1, nU, nI, K)
    res = {}
    for name, colmajor in (("rowmajor", 0), ("colmajor", 1)):
        Uc = np.ascontiguousarray(U.T if colmajor else U).copy()
        Vc = np.ascontiguousarray(V.T if colmajor else V).copy()
        orc.time_hogwild(Uc, Vc, u[: n // 20], i[: n // 20], r[: n // 20], nU, nI, K, lr, ureg, ireg, threads, colmajor)
        s = orc.time_hogwild(Uc, Vc, u, i, r, nU, nI, K, lr, ureg, ireg, threads, colmajor)
        res[name] = n / s
    return {"value": res["rowmajor"], "unit": "updates/s", "cores": threads, "kind": "port",
            "sample": "%d shuffled train ratings (%.2f epoch) of the same workload, OpenMP Hogwild "
                      "(modelMF.cpp:1746-1767 restated), row-major factors" % (n, n / tr.nnz),
            "value_colmajor_reference_layout": res["colmajor"]}


if __name__ == "__main__":
    main()
