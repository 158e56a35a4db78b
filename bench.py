#!/usr/bin/env python3
"""bench.py -- rating-updates/sec of the rank-64 SGD hot path on synthetic ML-20M-shape CSR (BASELINE.json C2).

    python bench.py --gpus N --steps K --warmup W

One "step" = one SGD epoch (device-side reshuffle + the tiled lock-free update kernel, the bracket of
modelMF.cpp:1739-1767) over the rank's train ratings.  N > 1: one process per GPU (started by torch.distributed.run, or
by this script itself when it finds no WORLD_SIZE), the rating matrix sharded by user-row blocks, the item-factor
replicas averaged with one RCCL all-reduce after every local epoch.
  --scaling weak    (default) every rank owns a full ML-20M-shape block of users over the SAME item catalogue
  --scaling strong  ONE ML-20M-shape matrix cut into N nnz-balanced user blocks (matfac_amd.dist.user_blocks)
Rank 0 prints ONE JSON line.  At N = 1 the line also carries `cpu_baseline` (the oracle's OpenMP loops on the host
cores), `secondary` (ALS at C3, CCD++ at C4, SGD at the C5-shard shape, each with its own roofline and CPU baseline)
and `rmse_parity` (the fast path against the reference's own seed-to-seed spread).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); f32 MFMA 157.3 TFLOP/s; 256 CUs x 4 SIMD-32, a wave64 vector instruction
# issues over 2 cycles, max clock 2.4 GHz
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TF = 157.3
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0      # G wave-instructions/s
L2_GATHER_PEAK_GBS = 17800.0               # rows gathered out of the XCDs' L2s: 16.8 - 18.8 TB/s chip-wide (MI355X_MICROARCH.md, 'Indexed rows')
PMC_SUMMARY = os.path.join("profiles", "r04_pmc_summary.json")
PMC_C4 = os.path.join("profiles", "r04_c4_pmc.json")
PMC_C5 = os.path.join("profiles", "r04_c5_pmc.json")


def _first_existing(path):
    """the committed counter summary of this round, or the newest earlier one (named in traffic_source either way)"""
    if os.path.exists(os.path.join(ROOT, path)):
        return path
    for r in ("r03", "r02", "r01"):
        q = path.replace("r04", r)
        if os.path.exists(os.path.join(ROOT, q)):
            return q
    return path


def host_cores():
    """(threads the OpenMP baselines run on, physical cores of the box): SURVEY 8(d) wants the core count stated"""
    from oracle import binding as orc
    threads = orc.max_threads()
    phys = set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
        if pid is not None and cid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    return threads, (len(phys) if phys else None)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (before anything
    here has touched the GPU), forward rank 0's JSON line, return their exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=8)   # (two turns through the four tilings of the tiled schedule; their lists are built in the first)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--rank-k", type=int, default=64, dest="K")
    ap.add_argument("--scale", type=float, default=1.0, help="scale nnz (debug only)")
    ap.add_argument("--arith", default="f32", choices=["f32", "ref64"])
    ap.add_argument("--mode", default="tiled", choices=["tiled", "hogwild"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--exchange", default="rotate", choices=["rotate", "allreduce"],
                    help="N > 1: rotate = item parts handed round a ring, every update applied once with its full step (default); "
                         "allreduce = replicas of V averaged with one all-reduce per epoch (what north_star names; every item step / N)")
    ap.add_argument("--blocks", type=int, default=0, help="workgroups in flight (0 = library heuristic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the ALS / CCD++ / C5-shard records (N = 1 only)")
    ap.add_argument("--secondary", default="als,ccd,c5", help="which secondary records to take")
    ap.add_argument("--no-parity", action="store_true", help="skip the rmse_parity record")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact_replay record")
    ap.add_argument("--cpu-sample", type=float, default=1.0, help="epochs of the CPU baseline sample")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0 and args.gpus > 1:
        sys.exit(self_launch(args))

    # everything but the final JSON line goes to stderr (RCCL prints its version banner on stdout)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = max(world, 1)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = world

    import numpy as np

    dist = None
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1"   # rehearse the N > 1 code path with one rank
    # BENCH_COMM=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks -- the ranks share the
    # visible GPUs and the exchange goes through mfx_comm_init_external + gloo instead of RCCL (never the driver's mode)
    use_gloo = os.environ.get("BENCH_COMM") == "gloo"
    multi = N > 1 or force_dist
    if multi:
        import torch
        import torch.distributed as dist
        if use_gloo:
            local_rank = local_rank % max(1, torch.cuda.device_count())
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from matfac_amd import Ctx, mfx, synth
    from matfac_amd import dist as mdist

    K = args.K
    shape = dict(synth.SHAPES[args.workload])
    # the NAMED shape is the training matrix: generate train/val/test = 80/10/10 around it
    shape["nnz"] = int(shape["nnz"] * args.scale / 0.8)
    t0 = time.time()
    if args.workload == "C5":
        # BASELINE.json config 5: ONE 10 M x 1 M matrix (8 fixed shards of 1.25 M users), strong scaling by construction
        if synth.C5_SHARDS % N != 0:
            raise SystemExit("--workload C5 is defined as %d user shards: --gpus must divide it" % synth.C5_SHARDS)
        per = synth.C5_SHARDS // N
        if (shape["nU"] // N) * 4 * ((K + 63) // 64 * 64) >= (1 << 32):
            raise SystemExit("--workload C5 at rank %d needs the user shard's factor table below 4 GiB (buffer addressing of the tiled "
                             "kernel): use --gpus >= %d" % (K, 4 if K > 128 else 2))
        args.scaling = "strong"
        tr, va, nU = synth.make_c5_shards(rank * per, per, seed=1, scale=args.scale)
    elif args.scaling == "strong" and N > 1:
        full = synth.make(shape, seed=1, shard=0)                 # the same matrix on every rank ...
        b = mdist.user_blocks(full["train"].rowptr, N)            # ... cut by train ratings
        tr = mdist.take_rows(full["train"], b[rank], b[rank + 1])
        va = mdist.take_rows(full["val"], b[rank], b[rank + 1])
        nU = int(b[rank + 1] - b[rank])
        del full
    else:
        d = synth.make(shape, seed=1, shard=rank)
        tr, va = d["train"], d["val"]
        nU = d["nUsers"]
    nI = shape["nI"]
    gen_s = time.time() - t0

    ctx = Ctx(local_rank)
    ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
    ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
    ctx.set_model(nU, nI, K)
    U0, _ = synth.init_factors(1 + rank, nU, nI, K, want_v=False)   # U shard: per-rank stream
    _, V0 = synth.init_factors(1, 1, nI, K, want_u=False)            # V replica: identical everywhere
    ctx.set_factors(U0, V0)
    ctx.compute_invalid()
    exchange = "RCCL over xGMI"
    ext_reducer = None           # the host-staged all-reduce the contexts use when the library's RCCL communicator is not (rehearsal / fallback)
    if multi:
        import torch
        if use_gloo:
            exchange = "gloo, staged through the host (rehearsal)"
            ext_reducer = lambda a: dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM)   # noqa: E731
            ctx.comm_init_external(N, rank, ext_reducer)
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
                exchange = "torch.distributed, staged through the host (library RCCL init failed)"
                if ok:
                    ctx.comm_destroy()

                def staged(a):
                    t = torch.from_numpy(a).cuda()
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    a[...] = t.cpu().numpy()
                ext_reducer = staged
                ctx.comm_init_external(N, rank, staged)
        ctx.comm_mark_synced()

    # main.cpp:29-31 defaults are learnrate 0.005, ureg = ireg = 0.01.  On ML-20M-skewed data the
    # reference's sequential loop diverges at 0.005 in its first epoch (NaN) and its own guard
    # (model.cpp:1486-1498) halves the rate: 0.0025 is where the CPU reference actually trains
    # (tests/test_fullsize_gpu.py::test_reference_loop_needs_the_halved_rate_at_c2 replays that first epoch).
    lr, ureg, ireg = 0.0025, 0.01, 0.01
    arith = mfx.ARITH_F32 if args.arith == "f32" else mfx.ARITH_REF64
    nnz = tr.nnz
    mode = mfx.SGD_TILED if args.mode == "tiled" else mfx.SGD_HOGWILD

    rotate = multi and args.exchange == "rotate" and mode == mfx.SGD_TILED
    if multi:
        exchange = ("item parts (item %% %d) handed round a ring -- one part per rank and sub-epoch, send/recv of 1/%d of V, all-gather after the "
                    "last sub-epoch: every update applied once with its full step" % (N, N) if rotate
                    else "item-factor all-reduce: replicas of V averaged once per epoch") + " [" + exchange + "]"
    if rotate:
        ctx.set_item_parts(N)

    def step(ep, exch=True):
        if rotate and exch:
            mdist.rotating_epoch(ctx, rank, N, lr, ureg, ireg, mode=mode, order=mfx.ORDER_DEVICE, arith=arith, seed=1, epoch=ep,
                                 blocks=args.blocks)
            return
        ctx.sgd_epoch(lr, ureg, ireg, mode=mode, order=mfx.ORDER_DEVICE, arith=arith, seed=1, epoch=ep,
                      blocks=args.blocks)
        if multi and exch:
            ctx.allreduce_item_factors(mfx.REDUCE_AVERAGE)   # replicas of V averaged (summed deltas overshoot: DESIGN.md 3.1)

    def barrier():
        ctx.synchronize()
        if multi:
            import torch
            dist.barrier()
            if not use_gloo:
                torch.cuda.synchronize()

    def global_rmse(which):
        o = ctx.eval(which)
        v = np.array([o.sse, float(o.n)])
        if multi:
            v = ctx.allreduce_f64(v)
        return float(np.sqrt(v[0] / v[1])) if v[1] else float("nan")

    for ep in range(args.warmup):
        step(ep)
    # HIP events inside the timed region, on the library's stream, around the launches of every 8th epoch: an event pair
    # around each of an epoch's 9 launches costs 6 % of its millisecond (scripts/event_overhead.py)
    prof_every = 8 if args.steps >= 16 else 1
    ctx.prof_enable(prof_every)
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
    val_rmse = global_rmse(mfx.MAT_VAL)
    tr_rmse = global_rmse(mfx.MAT_TRAIN)
    val_rmse_solo = None
    if multi:
        # the same epochs WITHOUT the exchange (every rank on its own replica of V): what averaging the replicas costs
        # or buys in convergence is the difference between the two validation RMSEs
        ctx.set_factors(U0, V0)
        ctx.comm_mark_synced()
        for ep in range(args.warmup + args.steps):
            step(ep, exch=False)
        val_rmse_solo = global_rmse(mfx.MAT_VAL)

    # ---- N > 1 sub-records (round 4): the OTHER exchange on the same data, and the strong-scaling split of the ONE named matrix next to
    # the weak-scaling default -- so that a scaling run measures what north_star names (one matrix, an item-factor all-reduce) as well as
    # the headline (rotation, one block per GPU).  Every rank takes the same path through this block (collectives inside).
    sub = {}

    def run_epochs(c, stepfn, warm, n_ep):
        """max over ranks of the wall time of n_ep epochs of stepfn on context c (after `warm` untimed ones)"""
        import torch
        for ep in range(warm):
            stepfn(ep)
        c.synchronize(); dist.barrier()
        t0_ = time.perf_counter()
        for ep in range(warm, warm + n_ep):
            stepfn(ep)
        c.synchronize(); dist.barrier()
        tt = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device="cpu" if use_gloo else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt[0])

    if multi and mode == mfx.SGD_TILED and os.environ.get("BENCH_NO_SUBRECORDS") != "1":
        n_ep = max(2, min(args.steps, 8))
        other = "allreduce" if rotate else "rotate"
        ctx.set_factors(U0, V0)
        ctx.comm_mark_synced()
        if other == "rotate":
            ctx.set_item_parts(N)

        def step_other(ep):
            if other == "rotate":
                mdist.rotating_epoch(ctx, rank, N, lr, ureg, ireg, mode=mode, order=mfx.ORDER_DEVICE, arith=arith, seed=1, epoch=ep, blocks=args.blocks)
            else:
                ctx.sgd_epoch(lr, ureg, ireg, mode=mode, order=mfx.ORDER_DEVICE, arith=arith, seed=1, epoch=ep, blocks=args.blocks)
                ctx.allreduce_item_factors(mfx.REDUCE_AVERAGE)
        el = run_epochs(ctx, step_other, 4, n_ep)
        # what one exchange costs on its own: the all-reduce of V / the ring shifts and the all-gather, timed without the epochs
        def exch_only(ep):
            if other == "rotate":
                steps_, held_ = mdist.rotation_schedule(rank, N)
                for part_, send_, recv_ in steps_:
                    if send_ is not None:
                        ctx.rotate_item_part(send_, recv_)
                ctx.allgather_item_parts(held_)
            else:
                ctx.allreduce_item_factors(mfx.REDUCE_AVERAGE)
        ex = run_epochs(ctx, exch_only, 1, 4)
        sub["exchange_" + other] = {"exchange": other, "epochs": n_ep, "ms_per_step": el / n_ep * 1e3,
                                    "ms_per_exchange_alone": ex / 4 * 1e3, "val_rmse_after_%d_epochs" % (4 + n_ep + 5): global_rmse(mfx.MAT_VAL),
                                    "note": "same data and schedule as the headline, the other exchange (rotate = item parts round a ring, every update "
                                            "applied once; allreduce = north_star's item-factor all-reduce, replicas averaged); the RMSE is after the 4 warm-up, "
                                            "the timed and the 5 exchange-only epochs of THIS sub-record"}
        if args.scaling == "weak" and args.workload != "C5":
            # strong scaling: ONE matrix of the named shape cut into N nnz-balanced user blocks, its own context and communicator
            full = synth.make(shape, seed=1, shard=0)
            bb = mdist.user_blocks(full["train"].rowptr, N)
            tr2 = mdist.take_rows(full["train"], bb[rank], bb[rank + 1])
            nU2 = int(bb[rank + 1] - bb[rank])
            full_nnz = int(full["train"].nnz)
            del full
            ctx2 = Ctx(local_rank)
            ctx2.set_csr(mfx.MAT_TRAIN, tr2.nrows, nI, tr2.rowptr, tr2.rowind, tr2.rowval)
            ctx2.set_model(nU2, nI, K)
            U2, _ = synth.init_factors(1 + rank, nU2, nI, K, want_v=False)
            ctx2.set_factors(U2, V0)
            ctx2.compute_invalid()
            import torch
            if ext_reducer is not None:
                ctx2.comm_init_external(N, rank, ext_reducer)
            else:
                uid2 = [Ctx.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid2, src=0)
                ctx2.comm_init(N, rank, uid2[0])
            ctx2.comm_mark_synced()
            ctx2.set_item_parts(N)

            def step_strong(ep):
                mdist.rotating_epoch(ctx2, rank, N, lr, ureg, ireg, mode=mode, order=mfx.ORDER_DEVICE, arith=arith, seed=1, epoch=ep, blocks=args.blocks)
            el2 = run_epochs(ctx2, step_strong, 4, n_ep)
            sub["strong_scaling"] = {"workload": "ONE %dx%d matrix, %d train ratings in total, cut into %d nnz-balanced user blocks" % (shape["nU"], nI, full_nnz, N),
                                     "exchange": "rotate", "epochs": n_ep, "ms_per_step": el2 / n_ep * 1e3, "value": full_nnz * n_ep / el2,
                                     "unit": "updates/s", "train_nnz_this_rank": int(tr2.nnz)}
            ctx2.comm_destroy()
            ctx2.close()

    total_nnz = nnz
    if multi:
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
        profiled_steps = (args.steps + prof_every - 1) // prof_every
        launches_per_step = max(1, sgd_launches // profiled_steps)   # tiled: 8 round launches per epoch
        kernel = "sgd_slots_kernel" if mode == mfx.SGD_TILED else "sgd_hogwild_kernel"
        out = {
            "metric": "rating-updates/sec @ rank=%d" % K, "value": value, "unit": "updates/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s synthetic CSR, %s, train nnz=%d %s, rank=%d, %s LOCK-FREE SGD epoch (hogTrain analogue: device "
                                   "reshuffle + update kernel%s; epochs take four tilings in turn, rounds in a fresh order); a lock-free trainer: whole loops land "
                                   "within 2e-3 of the reference hogTrain's mean test RMSE (rmse_parity.*.gpu_lock_free_tiled_hogtrain_*); "
                                   "the path that reproduces ModelMF::train to 1e-6 is exact_replay"
                                   % (args.workload, {"C1": "ML-100K-shape", "C2": "ML-20M-shape", "C4": "Netflix-shape", "C5": "10Mx1M (BASELINE config 5)"}.get(args.workload, ""),
                                      ("%dx%d per GPU (weak scaling: one such user block per GPU over the same items)" % (nU, nI)) if args.scaling == "weak" or N == 1
                                      else ("ONE %dx%d matrix cut into %d nnz-balanced user blocks (strong scaling)" % (shape["nU"], nI, N)),
                                      nnz if N == 1 or args.scaling == "weak" else total_nnz, "per GPU" if args.scaling == "weak" or N == 1 else "in total",
                                      K, "XCD-tiled" if mode == mfx.SGD_TILED else "flat", ", " + exchange if multi else ""),
                       "learnrate": lr, "ureg": ureg, "ireg": ireg, "arith": args.arith,
                       "parallelism": "user-block x%d" % N, "exchange": (args.exchange if multi else None)},
            "roofline": sgd_roofline(K, nnz, launches_per_step, avg_ms, sgd_launches, kernel),
            "permute_ms_per_step": perm_ms / max(1, profiled_steps),
            "val_rmse_after": val_rmse, "train_rmse_after": tr_rmse,
            "datagen_s": gen_s,
        }
        if multi:
            out["val_rmse_after_same_epochs_without_exchange"] = val_rmse_solo
            out["sub_records"] = sub
            out["config"]["sub_records"] = ("exchange_%s: the other exchange on the same blocks; strong_scaling: one %s matrix cut into %d user blocks (rotate)"
                                            % ("allreduce" if rotate else "rotate", args.workload, N)) if sub else None
    solo = N == 1 and not force_dist
    exact = None
    if rank == 0 and solo and not args.no_exact:
        try:
            exact = exact_replay(np, ctx, mfx, tr, U0, V0, lr, ureg, ireg, hybrid_too=True)
        except Exception as e:                  # noqa: BLE001 -- reported, the bench line does not depend on it
            exact = {"error": str(e)}
    if multi:
        ctx.comm_destroy()
        dist.barrier()
    ctx.close()
    if rank == 0 and solo:
        if exact is not None:
            if "error" not in exact:
                try:        # (after ctx.close(): the host class opens its own context on the device)
                    exact["host_class_loop"] = host_loop_ms(np, d, K, lr, ureg, ireg)
                except Exception as e:              # noqa: BLE001
                    exact["host_class_loop"] = {"error": str(e)}
            out["exact_replay"] = exact
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(np, tr, nU, nI, K, lr, ureg, ireg, args.cpu_sample)
        if not args.no_secondary and args.workload == "C2" and args.scale == 1.0:
            out["secondary"] = secondary(np, args, d, K, with_cpu=not args.no_cpu_baseline)
        if not args.no_parity:
            out["rmse_parity"] = rmse_parity(np)
        if not args.no_secondary and args.workload == "C2" and args.scale == 1.0:
            try:
                out["secondary"].append(default_rate_record(np, d, K, ureg, ireg))
            except Exception as e:              # noqa: BLE001
                out["secondary"].append({"config": "C2 at the default learning rate", "error": str(e)})
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    try:
        os.write(2, b"\n")          # (the host classes end their last line without a newline: keep the JSON at a line start for 2>&1 readers)
    except OSError:
        pass
    os.dup2(saved_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


def exact_replay(np, ctx, mfx, tr, U0, V0, lr, ureg, ireg, epochs=3, hybrid_too=False):
    """The path that IS the reference's sequential loop (ModelMF::train, modelMF.cpp:83-105): the same workload, one full-list
    permutation per epoch (what std::shuffle hands the loop), the reference's double bracket, replayed bit for bit by the
    tagged dataflow schedule (sgd_flow.hip; np.array_equal with the oracle at this size in tests/test_fullsize_gpu.py).
    updates/s over the whole mfx_sgd_epoch call (queue construction on the device + kernels) and over the kernels alone."""
    rng = np.random.default_rng(1)
    ctx.set_factors(U0, V0)
    ctx.prof_enable(True)
    calls, kernels = [], []
    info = None
    for ep in range(epochs + 1):
        order = rng.permutation(tr.nnz).astype(np.uint64)
        ctx.sgd_set_order(order)
        ctx.synchronize()
        ctx.prof_reset()
        t0 = time.perf_counter()
        ctx.sgd_epoch(lr, ureg, ireg, mode=mfx.SGD_LEVELS, order=mfx.ORDER_HOST, arith=mfx.ARITH_REF64)
        ctx.synchronize()
        wall = time.perf_counter() - t0
        ms, _ = ctx.prof_get(mfx.K_SGD)
        if ep > 0:                              # epoch 0: allocations of the schedule's buffers
            calls.append(wall)
            kernels.append(ms * 1e-3)
        info, _ = ctx.debug_levels_info()
    ctx.prof_enable(False)
    call_s, kern_s = float(np.median(calls)), float(np.median(kernels))
    hybrid = None
    if hybrid_too:
        # the same epochs with the item rows owned throughout (MFX_FLOW_HYBRID=0: what rounds 2 and 3 ran; the default since round 4 gives
        # the busiest users queues of their own -- DESIGN.md section 3.1.2)
        try:
            os.environ["MFX_FLOW_HYBRID"] = "0"
            ctx.set_factors(U0, V0)
            ctx.prof_enable(True)
            hk = []
            for ep in range(2):
                ctx.sgd_set_order(rng.permutation(tr.nnz).astype(np.uint64))
                ctx.synchronize()
                ctx.prof_reset()
                ctx.sgd_epoch(lr, ureg, ireg, mode=mfx.SGD_LEVELS, order=mfx.ORDER_HOST, arith=mfx.ARITH_REF64)
                ctx.synchronize()
                hk.append(ctx.prof_get(mfx.K_SGD)[0] * 1e-3)
            ctx.prof_enable(False)
            hinfo, _ = ctx.debug_levels_info()
            hybrid = {"kernel_ms_per_epoch": float(min(hk)) * 1e3, "kernel_updates_per_s": tr.nnz / float(min(hk)), "longest_queue": int(hinfo[1]),
                      "note": "MFX_FLOW_HYBRID=0: item rows owned throughout, every visit of a busy user a hand-off between item queues (the schedule of rounds 2 and 3)"}
        except Exception as e:                  # noqa: BLE001
            hybrid = {"error": str(e)}
        finally:
            os.environ.pop("MFX_FLOW_HYBRID", None)
    return {"item_rows_owned_throughout": hybrid,
            "path": "MFX_SGD_LEVELS, tagged dataflow schedule (order replay of ModelMF::train, double bracket): bit-identical to the "
                    "sequential loop", "updates_per_s": tr.nnz / call_s, "ms_per_epoch": call_s * 1e3,
            "kernel_updates_per_s": tr.nnz / kern_s, "kernel_ms_per_epoch": kern_s * 1e3, "epochs": epochs,
            "longest_queue": int(info[1]), "queues": int(info[2]),
            "frac_of_hbm_roofline_model": (16 * U0.shape[1] + 12) * tr.nnz / call_s / 1e9 / HBM_PEAK_GBS,
            "bound": "the most popular item's queue: longest_queue visits x the step of one visit (the busiest users have queues of their own: hybrid "
                     "ownership, DESIGN.md 3.1.2; with the item rows owned throughout the busiest user's 10 717 ratings are 10 717 hand-offs between "
                     "item queues at ~ 1.3 us, scripts/flow_model.py) -- latency, not bandwidth",
            "test_rmse_vs_reference": "rmse_parity.*.gpu_default_path_* (whole training loops against the fixture's seed-1 row)"}


def pmc(kernel, src=None):
    """Counters of `kernel` from the committed rocprofv3 PMC passes of this same command (scripts/profile_round.sh ->
    profiles/r04_pmc_summary.json).  rocprofv3 cannot run inside this process: these are per-launch means of THAT run,
    named as such in the record; None when the summary has no entry for the kernel."""
    path = os.path.join(ROOT, src or _first_existing(PMC_SUMMARY))
    try:
        return json.load(open(path))[kernel]
    except Exception:
        return None


def sgd_roofline(K, nnz, launches_per_step, avg_ms, launches, kernel):
    """SURVEY 8(d)'s fraction for the dominant kernel, counter-backed.

    `frac` = bytes that crossed the L2s' memory side per launch (FETCH_SIZE x 2 + WRITE_SIZE from the committed separate --pmc
    passes of this same command, profiles/r04_pmc_summary.json) / this run's average launch time (HIP events) / the 8 TB/s HBM peak:
    one division on two committed numbers.  At C2 those bytes are fabric requests, Infinity-Cache hits included (the 42 MB of
    factors never leave the MALL), so it is an upper bound on real HBM use -- `bound` says "l2-memory-side", not "hbm".
    Named secondaries, none of them `frac`:
      algorithmic  SURVEY 8(d)'s model line, (16K+12) bytes per update as if every row came from and went back to HBM; above 1 at
                   this working set (item rows owned in LDS, user rows served by L2) -- not a bound here, the work IS done
                   (device visit counter all ones at C2, tests/test_fullsize_gpu.py);
      l2_rows      the lock-free rows through the XCDs' L2s (one row read + one written per update) against the 16.8 - 18.8 TB/s the
                   guide measured for L2-served row gathers;
      valu_issue   SQ_INSTS_VALU per update at the guide's 2 cycles per wave64 instruction (no kernel-specific derating)."""
    avg_s = avg_ms * 1e-3
    per_launch = nnz / launches_per_step
    alg_bytes = (16 * K + 12) * per_launch
    r = {"kernel": kernel, "avg_launch_ms": avg_ms, "launches": launches, "launches_per_step": launches_per_step,
         "updates_per_launch": per_launch,
         "algorithmic": {"bytes_per_launch": alg_bytes, "achieved": alg_bytes / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS,
                         "note": "SURVEY 8(d) model: every row read and written in HBM; exceeds 1 when the rows are "
                                 "served from LDS / L2 / Infinity Cache (C2: 42 MB of factors) -- not a bound at this working set"}}
    src = _first_existing(PMC_SUMMARY)
    c = pmc(kernel, src)
    ld = (K + 63) // 64 * 64 if K > 32 else (32 if K > 16 else 16)
    row_bytes = 2 * 4 * ld * per_launch            # the lock-free row of every update: read once, written once, through one L2
    l2_ach = row_bytes / avg_s / 1e9
    r["l2_rows"] = {"bytes_per_update": 2 * 4 * ld, "achieved": l2_ach, "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s", "frac": l2_ach / L2_GATHER_PEAK_GBS,
                    "note": "user rows through the XCD's L2 (sc1 loads, plain stores); peak = middle of the 16.8 - 18.8 TB/s measured for "
                            "L2-served row gathers (MI355X_MICROARCH.md)"}
    if c and "SQ_INSTS_VALU" in c.get("counters_mean_per_launch", {}):
        valu = c["counters_mean_per_launch"]["SQ_INSTS_VALU"]
        per_update = valu / c.get("updates_per_launch", per_launch)
        ach = per_update * per_launch / avg_s / 1e9
        r["valu_issue"] = {"wave_instructions_per_update": per_update, "achieved": ach, "peak": VALU_PEAK_GINST,
                           "unit": "G wave-instr/s", "frac": ach / VALU_PEAK_GINST,
                           "note": "SQ_INSTS_VALU per launch from %s / updates of that launch; peak = 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per "
                                   "wave64 instruction (MI355X_MICROARCH.md)" % src}
    if c and "hbm_bytes_per_launch" in c:
        traffic = c["hbm_bytes_per_launch"] * (per_launch / c.get("updates_per_launch", per_launch))
        ach = traffic / avg_s / 1e9
        r.update({"bound": "l2-memory-side", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                  "traffic": traffic, "l2_hit_rate": c.get("l2_hit_rate"),
                  "traffic_over_algorithmic": traffic / alg_bytes,
                  "traffic_source": "%s: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE per launch (fabric requests of the L2s, Infinity-Cache hits "
                                    "included: an upper bound on HBM bytes), separate --pmc passes of this command; time from this run's HIP "
                                    "events; frac = traffic / avg_launch_ms / 8 TB/s" % src})
    else:
        a = r["algorithmic"]
        r.update({"bound": "hbm (model line: no committed counters for this kernel)", "achieved": a["achieved"], "peak": a["peak"],
                  "unit": "GB/s", "frac": a["frac"], "traffic": None})
    return r


def cpu_baseline(np, tr, nU, nI, K, lr, ureg, ireg, epochs):
    """The oracle's OpenMP loops on the host cores: Hogwild (restatement of the modelMF.cpp:1746-1767 bracket) and the
    stratified epoch of trainSGDPar (:271-309).  Reported baseline only.  Sample = `epochs` epoch(s) over the first part
    of the same train list (Hogwild) / the first users of the same matrix (stratified)."""
    from oracle import binding as orc
    from matfac_amd import synth
    threads, phys = host_cores()
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
    out = {"value": res["rowmajor"], "unit": "updates/s", "cores": threads, "physical_cores": phys, "kind": "port",
           "sample": "%d shuffled train ratings (%.2f epoch) of the same workload, OpenMP Hogwild "
                     "(modelMF.cpp:1746-1767 restated), row-major factors" % (n, n / tr.nnz),
           "value_colmajor_reference_layout": res["colmajor"]}
    # trainSGDPar: T = threads parts; every rating is scanned T times per epoch and updated about once, so the sample is
    # the first users holding ~1/8 of the ratings (bounded CPU time), timed for one epoch
    try:
        m = int(np.searchsorted(tr.rowptr, tr.nnz // 8))
        sub_ptr = tr.rowptr[: m + 1]
        sub_n = int(sub_ptr[-1])
        invU = np.zeros(m, np.uint8)
        invI = np.zeros(nI, np.uint8)
        Us, Vs = U[:m].copy(), V.copy()
        s = orc.time_strat(Us, Vs, sub_ptr, tr.rowind[:sub_n], tr.rowval[:sub_n], m, nI, invU, invI, threads, lr, ureg, ireg)
        out["stratified"] = {"value": sub_n / s, "unit": "updates/s (ratings of the matrix per epoch time)", "cores": threads, "physical_cores": phys,
                             "sample": "first %d users, %d ratings, one epoch of trainSGDPar (modelMF.cpp:271-309 restated), "
                                       "T = %d parts" % (m, sub_n, threads)}
    except Exception as e:                      # noqa: BLE001 -- the baseline is optional, the bench line is not
        out["stratified"] = {"error": str(e)}
    return out


def secondary(np, args, d2, K, with_cpu=True):
    """BASELINE.json configs 3 and 4 and one GPU's share of config 5, each one short timed run with its own roofline
    and (unless --no-cpu-baseline) a sampled CPU baseline from the oracle."""
    from matfac_amd import Ctx, mfx, synth
    want = set(args.secondary.split(","))
    recs = []
    if "als" in want:
        try:
            recs.append(bench_als(np, d2, 64, with_cpu))
        except Exception as e:                  # noqa: BLE001
            recs.append({"config": "C3 ALS", "error": str(e)})
    if "ccd" in want:
        try:
            recs.append(bench_ccd(np, with_cpu))
        except Exception as e:                  # noqa: BLE001
            recs.append({"config": "C4 CCD++", "error": str(e)})
    if "c5" in want:
        try:
            recs.append(bench_c5_shard(np))
        except Exception as e:                  # noqa: BLE001
            recs.append({"config": "C5 shard SGD", "error": str(e)})
    return recs


def bench_als(np, d, K, with_cpu):
    """C3: MovieLens-20M shape, rank 64, ALS (modelMF.cpp:792-886 bracket: one users + items sweep = one step)."""
    from matfac_amd import Ctx, mfx, synth
    tr, va = d["train"], d["val"]
    nU, nI = d["nUsers"], synth.SHAPES["C2"]["nI"]
    reg, iters = 5.0, 5
    U0, V0 = synth.init_factors(1, nU, nI, K)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
        ctx.set_model(nU, nI, K)
        ctx.set_factors(U0, V0)
        ctx.compute_invalid()
        ctx.als_half_sweep(mfx.SIDE_USERS, reg)
        ctx.als_half_sweep(mfx.SIDE_ITEMS, reg)
        ctx.synchronize()
        ctx.prof_enable(True)
        ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(iters):
            ctx.als_half_sweep(mfx.SIDE_USERS, reg)
            ctx.als_half_sweep(mfx.SIDE_ITEMS, reg)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / iters
        g_ms, g_n = ctx.prof_get(mfx.K_ALS_GRAM)
        s_ms, s_n = ctx.prof_get(mfx.K_ALS_SOLVE)
        val = ctx.rmse(mfx.MAT_VAL)
    # 2*(2K^2+2K) flops per rating per iteration (both triangles, as the reference forms them) + K^3/3+2K^2 per row
    flops = 2 * tr.nnz * (2 * K * K + 2 * K) + (nU + nI) * (K ** 3 / 3 + 2 * K * K)
    ev = (g_ms + s_ms) / iters * 1e-3
    rec = {"config": "C3: ML-20M-shape, train nnz=%d, rank=%d, ALS (MFMA Gramian + in-register LDL^T), reg=%.1f" % (tr.nnz, K, reg),
           "metric": "rating-iterations/sec", "value": tr.nnz / wall, "ms_per_iteration": wall * 1e3, "steps": iters,
           "kernel_ms_per_iteration": ev * 1e3, "val_rmse_after": val,
           "roofline": {"bound": "mfma", "achieved": flops / ev / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                        "frac": flops / ev / 1e12 / MFMA_F32_PEAK_TF, "flops_per_iteration": flops, "traffic": None,
                        "kernel": "als_segment_kernel + als_reduce_kernel (HIP events)",
                        "note": "flop count of the reference (both triangles); the kernels form 3 of the 4 32x32 tiles"}}
    if with_cpu:
        from oracle import binding as orc
        threads, phys = host_cores()
        m = int(np.searchsorted(tr.rowptr, 1_000_000))           # users holding the first ~1 M ratings
        X = U0[:m].copy()
        t0 = time.perf_counter()
        orc.als_half(0, X, V0, m, tr.rowptr[: m + 1], tr.rowind, tr.rowval, np.zeros(m, np.uint8), reg, nthreads=threads)
        s = time.perf_counter() - t0
        n = int(tr.rowptr[m])
        rec["cpu_baseline"] = {"value": n / s / 2, "unit": "rating-iterations/sec", "cores": threads, "physical_cores": phys, "kind": "port",
                               "sample": "user half-sweep over the first %d users (%d ratings), modelMF.cpp:805-841 restated; "
                                         "an iteration is two half-sweeps, so ratings / seconds / 2" % (m, n)}
    return rec


def bench_ccd(np, with_cpu):
    """C4: Netflix shape (480 189 x 17 770, 100 M train ratings), rank 128, CCD++ (modelMF.cpp:1025-1126 bracket)."""
    from matfac_amd import Ctx, mfx, synth
    K, reg, nk = 128, 2.0, 16
    shape = dict(synth.SHAPES["C4"])
    shape["nnz"] = int(shape["nnz"] / 0.8)
    t0 = time.time()
    d = synth.make(shape, seed=1)
    gen = time.time() - t0
    tr, va = d["train"], d["val"]
    nU, nI = d["nUsers"], shape["nI"]
    U0, V0 = synth.init_factors(1, nU, nI, K)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
        ctx.set_model(nU, nI, K)
        ctx.set_factors(U0, V0)
        ctx.compute_invalid()
        ctx.ccdpp_begin()
        for k in range(2):
            ctx.ccdpp_rank1(k, reg, reg, add_back=False)
        ctx.synchronize()
        ctx.prof_enable(True)
        ctx.prof_reset()
        t0 = time.perf_counter()
        for k in range(nk):
            ctx.ccdpp_rank1(k, reg, reg, add_back=True)
        ctx.synchronize()
        per_k = (time.perf_counter() - t0) / nk
        r_ms, r_n = ctx.prof_get(mfx.K_CCD_ROW)
        c_ms, c_n = ctx.prof_get(mfx.K_CCD_COL)
        x_ms, x_n = ctx.prof_get(mfx.K_CCD_RESID)
        ctx.ccdpp_end()
        # the order replay of ModelMF::train at THIS size (100 M ratings, rank 128): above MFX_EXACT_SEQ_BELOW the host class takes the
        # lock-free schedule unless MFX_EXACT=1; this is what the replay itself costs here
        replay = None
        try:
            ctx.prof_enable(False)
            ctx.set_factors(U0, V0)
            replay = exact_replay(np, ctx, mfx, tr, U0, V0, 0.0025, 0.01, 0.01, epochs=2)
            replay.pop("test_rmse_vs_reference", None)
        except Exception as e:                  # noqa: BLE001
            replay = {"error": str(e)}
    bytes_per_k = 128 * tr.nnz                    # 128 B per (rating, factor) per outer iteration, T = 5
    rec = {"config": "C4: Netflix-shape %dx%d, train nnz=%d, rank=%d, CCD++ (5 inner sweeps per factor), reg=%.1f" % (nU, nI, tr.nnz, K, reg),
           "metric": "rating-factor updates/sec", "value": tr.nnz / per_k, "ms_per_factor": per_k * 1e3,
           "s_per_outer_iteration": per_k * K, "steps": nk, "datagen_s": gen,
           "row_pass_ms": r_ms / max(r_n, 1), "col_pass_ms": c_ms / max(c_n, 1),
           "resid_update_ms": (x_ms / x_n) if x_n else None,    # None: the update rides on each factor's first sweep (fused pass kernels), inside row/col_pass_ms
           "roofline": {"bound": "hbm", "achieved": bytes_per_k / per_k / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": bytes_per_k / per_k / 1e9 / HBM_PEAK_GBS, "bytes_per_factor": bytes_per_k, "traffic": None,
                        "kernel": "ccd_pass_kernel / colpass_kernel (segmented reduction over padded 128-entry trips; the first sweep of a factor as ccd_pass_fused_kernel / colpass_fused_kernel with the residual update on the way) (whole rank-one step, wall clock)"}}
    rec["exact_replay_of_ModelMF_train_at_this_size"] = replay
    # memory-side bytes of one rank-one step from the committed counter passes (scripts/pmc_c4.sh), time from this run
    try:
        c4src = _first_existing(PMC_C4)
        with open(os.path.join(ROOT, c4src)) as f:
            pmc = json.load(f)
        rec["roofline"]["traffic"] = pmc["hbm_bytes_per_factor"]
        rec["roofline"]["traffic_GBs"] = pmc["hbm_bytes_per_factor"] / per_k / 1e9
        rec["roofline"]["traffic_frac_of_peak"] = pmc["hbm_bytes_per_factor"] / per_k / 1e9 / HBM_PEAK_GBS
        rec["roofline"]["traffic_source"] = (c4src + ": sum over the kernels of a rank-one step of (FETCH_SIZE x2 + WRITE_SIZE) per launch "
                                             "x launches per step, separate --pmc passes of scripts/bench_als_ccd.py; time from this run")
    except (OSError, KeyError, ValueError):
        pass
    if with_cpu:
        from oracle import binding as orc
        threads, phys = host_cores()
        m = int(np.searchsorted(tr.rowptr, 4_000_000))           # first users holding ~4 M ratings
        n = int(tr.rowptr[m])
        rp, ri, rv = tr.rowptr[: m + 1], tr.rowind[:n], tr.rowval[:n]
        cp, ci, cv = orc.create_col_index(m, nI, rp, ri, rv)
        Us, Vs = np.zeros((m, K), np.float32), V0.copy()
        rr, rc = rv.copy(), cv.copy()
        invU = np.zeros(m, np.uint8)
        invI = (np.diff(cp) == 0).astype(np.uint8)
        orc.ccdpp_rank1(0, Us, Vs, m, nI, nI, rp, ri, rr, cp, ci, rc, invU, invI, reg, reg, False, nthreads=threads)
        t0 = time.perf_counter()
        for k in (1, 2):
            orc.ccdpp_rank1(k, Us, Vs, m, nI, nI, rp, ri, rr, cp, ci, rc, invU, invI, reg, reg, True, nthreads=threads)
        s = (time.perf_counter() - t0) / 2
        rec["cpu_baseline"] = {"value": n / s, "unit": "rating-factor updates/sec", "cores": threads, "physical_cores": phys, "kind": "port",
                               "sample": "first %d users (%d ratings), two rank-one steps with add-back, modelMF.cpp:1027-1121 restated" % (m, n)}
    return rec


def bench_c5_shard(np):
    """One GPU's share of C5 (10 M x 1 M, 1 B ratings, rank 256 over 8 GPUs): 1.25 M users x 1 M items, 125 M train
    ratings, U 1.28 GB + V 1.02 GB -- a working set beyond every cache, where HBM does bind the SGD update."""
    from matfac_amd import Ctx, mfx, synth
    K, n = 256, 4
    shape = dict(nU=1_250_000, nI=1_000_000, nnz=int(125_000_000 / 0.8), K=K)
    t0 = time.time()
    d = synth.make(shape, seed=1, r0_i=0.002)
    gen = time.time() - t0
    tr, va = d["train"], d["val"]
    nU, nI = d["nUsers"], shape["nI"]
    U0, V0 = synth.init_factors(1, nU, nI, K)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
        ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
        ctx.set_model(nU, nI, K)
        ctx.set_factors(U0, V0)
        ctx.compute_invalid()
        v0 = ctx.rmse(mfx.MAT_VAL)
        for ep in range(4):                    # (one epoch on each of the four tilings: their slot lists are built here)
            ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=ep)
        ctx.synchronize()
        ctx.prof_enable(True)
        ctx.prof_reset()
        t0 = time.perf_counter()
        for ep in range(4, 4 + n):
            ctx.sgd_epoch(0.0025, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=ep)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / n
        ms, cnt = ctx.prof_get(mfx.K_SGD)
        v1 = ctx.rmse(mfx.MAT_VAL)
    alg = (16 * K + 12) * tr.nnz
    # compulsory HBM traffic of one epoch: every rating record once (16 B) + both factor tables read and written once
    compulsory = 16 * tr.nnz + 2 * 4 * K * (nU + nI)
    launch_ms = ms / max(cnt, 1)
    roof = {"kernel": "sgd_slots_kernel<16,4,F32>", "avg_launch_ms": launch_ms, "launches": cnt,
            "algorithmic": {"bytes_per_step": alg, "achieved": alg / wall / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": alg / wall / 1e9 / HBM_PEAK_GBS,
                            "note": "SURVEY 8(d)'s 16K+12 bytes per update; above 1 because item rows are owned in LDS for a slot and a "
                                    "user row serves several ratings of a tile from L2: fewer bytes cross the memory side than the model counts"},
            "compulsory_bytes_per_step": compulsory}
    try:   # bytes that really crossed the L2's memory side, from the committed PMC passes of scripts/c5_shard.py (scripts/pmc_c5.sh)
        c5src = _first_existing(PMC_C5)
        c = json.load(open(os.path.join(ROOT, c5src)))
        traffic = c["hbm_bytes_per_launch"] * (tr.nnz / 8.0) / c["updates_per_launch"]
        ach = traffic / (launch_ms * 1e-3) / 1e9
        roof.update({"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": c5src + ": FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE per round launch, separate "
                                       "--pmc passes of the same workload; launch time from this run's HIP events",
                     "l2_hit_rate": c.get("l2_hit_rate")})
    except Exception:                             # noqa: BLE001 -- no committed counters: only the model line
        a = roof["algorithmic"]
        roof.update({"bound": "hbm", "achieved": a["achieved"], "peak": a["peak"], "unit": "GB/s", "frac": a["frac"], "traffic": None})
    return {"config": "C5 shard: %dx%d, train nnz=%d, rank=%d, XCD-tiled Hogwild SGD epoch on ONE GPU (1/8 of config 5)" % (nU, nI, tr.nnz, K),
            "metric": "rating-updates/sec @ rank=%d" % K, "value": tr.nnz / wall, "ms_per_step": wall * 1e3, "steps": n,
            "datagen_s": gen, "val_rmse_before": v0, "val_rmse_after": v1, "roofline": roof}


def rmse_parity(np):
    """The fast path against the reference's OWN run-to-run spread (tests/golden/sgd_spread_*.json, produced by
    tests/golden/make_sgd_spread.py from the CPU oracle's full training loops; nothing of the oracle runs here): the
    same seeded synthetic matrix trained by the host classes, default path and forced lock-free schedule."""
    import ctypes as C
    from matfac_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    recs = {}
    for name in ("c1", "mid"):
        path = os.path.join(ROOT, "tests", "golden", "sgd_spread_%s.json" % name)
        if not os.path.exists(path):
            continue
        f = json.load(open(path))
        cfg = f["config"]
        shape = dict(synth.SHAPES[cfg["shape"]]) if isinstance(cfg["shape"], str) else dict(cfg["shape"])
        shape["nnz"] = int(shape["nnz"] / 0.8)
        d = synth.make(shape, seed=cfg["data_seed"])
        seq = np.array([x["test_rmse"] for x in f["sequential"]])
        hog = np.array([x["test_rmse"] for x in f["hogwild"]])
        par = np.array([x["test_rmse"] for x in f.get("sgdpar", [])])
        rec = {"data": "%s synthetic, train nnz=%d, rank=%d, lr=%g, maxiter=%d" % (cfg["shape"], f["train_nnz"], cfg["K"], cfg["lr"], cfg["maxIter"]),
               "reference_sequential_test_rmse_mean": float(seq.mean()), "reference_sequential_test_rmse_std": float(seq.std(ddof=1)),
               "reference_sequential_seeds": int(seq.size), "reference_sequential_seed1_test_rmse": float(f["sequential"][0]["test_rmse"]),
               "reference_hogwild_test_rmse": [float(x) for x in hog],
               "source": "tests/golden/sgd_spread_%s.json" % name}
        if par.size:
            rec.update({"reference_trainSGDPar_test_rmse_mean": float(par.mean()), "reference_trainSGDPar_test_rmse_std": float(par.std(ddof=1)),
                        "reference_trainSGDPar_seeds": int(par.size), "reference_trainSGDPar_parts": f.get("sgdpar_parts")})
        # gpu_default_path: ModelMF::train as the host class runs it by default (order replay up to 32 M ratings): the fixture's
        # seed-1 row.  gpu_lock_free_tiled: MFX_EXACT=0, the schedule of the headline `value`.
        for label, env in (("gpu_default_path", {}), ("gpu_lock_free_tiled", {"MFX_EXACT": "0"})):
            try:
                t = host_train_rmse(C, np, synth, d, cfg, env)
                rec[label + "_test_rmse"] = t
                rec[label + "_sigmas_from_mean"] = (t - float(seq.mean())) / float(seq.std(ddof=1))
                rec[label + "_delta_vs_reference_seed1"] = t - float(f["sequential"][0]["test_rmse"])
                if par.size:
                    rec[label + "_sigmas_from_trainSGDPar_mean"] = (t - float(par.mean())) / float(par.std(ddof=1))
            except Exception as e:              # noqa: BLE001
                rec[label + "_error"] = str(e)
        # the trainer the lock-free tiled schedule stands in for: the reference's hogTrain (modelMF.cpp:1747-1763), the fixture's
        # `hogwild` rows, at the reference's default rate -- test RMSE, distance from the hogwild mean, NaN rollbacks of the run
        try:
            log = []
            st = host_train_stats(C, np, synth, d, cfg, {"MFX_EXACT": "0"}, method=b"hogsgd", log=log)
            t = float(st[1])
            rec.update({"reference_hogwild_test_rmse_mean": float(hog.mean()), "reference_hogwild_test_rmse_std": float(hog.std(ddof=1)),
                        "reference_hogwild_draws": int(hog.size), "reference_hogwild_best_iter": [int(x["best_iter"]) for x in f["hogwild"]],
                        "gpu_lock_free_tiled_hogtrain_test_rmse": t, "gpu_lock_free_tiled_hogtrain_iterations": int(st[7]),
                        "gpu_lock_free_tiled_hogtrain_found_nan": log[0].count("Found nan"),
                        "gpu_lock_free_tiled_hogtrain_delta_from_hogwild_mean": t - float(hog.mean()),
                        "gpu_lock_free_tiled_hogtrain_sigmas_from_hogwild_mean": (t - float(hog.mean())) / float(hog.std(ddof=1)),
                        "gpu_lock_free_tiled_hogtrain_delta_in_sequential_sigmas": (t - float(hog.mean())) / float(seq.std(ddof=1)),
                        "gpu_lock_free_tiled_hogtrain_inside_hogwild_envelope": bool(hog.min() <= t <= hog.max()),
                        "gpu_lock_free_tiled_hogtrain_gate": "|delta| <= 2e-3 (mid) / inside the hogwild envelope +- one sequential sigma (c1): tests/test_parity_spread_gpu.py"})
        except Exception as e:                  # noqa: BLE001
            rec["gpu_lock_free_tiled_hogtrain_error"] = str(e)
        recs[name] = rec
    return recs


def default_rate_record(np, d, K, ureg, ireg, iters=12):
    """What the reference's DEFAULT learning rate (main.cpp:29: 0.005) does on the headline workload: hogTrain through the host
    class on the C2 matrix (20 M ratings > MFX_EXACT_BELOW: the lock-free tiled schedule, first epoch as the sequential replay of a
    shuffled device order), `iters` iterations with the reference's NaN guard.  The reference's own sequential loop leaves its
    first epoch with NaN at this rate on this matrix (tests/test_fullsize_gpu.py::test_reference_loop_needs_the_halved_rate_at_c2),
    and so does an exact replay of it: `found_nan` counts the rollbacks, `final_learnrate` shows the halvings."""
    import ctypes as C
    from matfac_amd import synth
    cfg = {"K": K, "lr": 0.005, "ureg": ureg, "ireg": ireg, "maxIter": iters}
    rec = {"config": "C2 at the reference's default learning rate 0.005: ModelMF::hogTrain through libmfhost.so, %d iterations" % iters}
    for lr in (0.005, 0.0025):
        cfg["lr"] = lr
        log = []
        st = host_train_stats(C, np, synth, d, cfg, {}, method=b"hogsgd", log=log)
        rec["learnrate_%g" % lr] = {"found_nan": log[0].count("Found nan"), "final_learnrate": float(st[3]), "iterations": int(st[7]),
                                     "best_val_rmse": float(st[2]), "test_rmse_of_best": float(st[1]), "loop_s": float(st[6]),
                                     "schedule": "lock-free tiled" if "lock-free tiled schedule" in log[0] else "order replay"}
    return rec


def host_loop_ms(np, d, K, lr, ureg, ireg):
    """ModelMF::train end to end through the host class (libmfhost.so): what one iteration of the reference's loop costs a caller --
    the epoch's std::shuffle order drawn on the host (bit-identical to the library's, mf_model.cpp), its upload, the replay on the
    device and Model::isTerminateModel.  Steady state = (9 iterations - 3 iterations) / 6."""
    import ctypes as C
    from matfac_amd import synth
    if not all(k in d for k in ("train", "val", "test")):
        return {"error": "the workload has no val / test split"}
    cfg = {"K": K, "lr": lr, "ureg": ureg, "ireg": ireg}
    t = {}
    for iters in (3, 9):
        cfg["maxIter"] = iters
        stats = host_train_stats(C, np, synth, d, cfg, {})
        t[iters] = (float(stats[6]), int(stats[7]))
    if t[9][1] <= t[3][1]:
        return {"error": "the loops ended after %d and %d iterations" % (t[3][1], t[9][1])}
    ms = (t[9][0] - t[3][0]) / (t[9][1] - t[3][1]) * 1e3
    nnz = d["train"].nnz
    return {"trainer": "ModelMF::train (order replay, the default up to 128 M train ratings)", "ms_per_iteration": ms,
            "updates_per_s": nnz / (ms * 1e-3), "includes": "the positions of the epoch's std::shuffle drawn on the host (a thread ahead), their "
            "upload, the swaps applied on the device, the replay, objective + validation RMSE and the termination rule of every iteration"}


def host_train_rmse(C, np, synth, d, cfg, env):
    return float(host_train_stats(C, np, synth, d, cfg, env)[1])


def host_train_stats(C, np, synth, d, cfg, env, method=b"sgd", log=None):
    """A trainer of the host class through libmfhost.so (mfh_train): best-validation model's test RMSE.  (The class prints the
    reference's per-iteration lines on stdout; bench.py has already pointed stdout at stderr.)  log = a list: the trainer's
    output is also appended to it as one string (for counting its "Found nan" lines)."""
    import tempfile
    lib = synth._host()
    tr, va, te = d["train"], d["val"], d["test"]
    nU, nI, K = d["nUsers"], d["nItems"], cfg["K"]
    bufs = [np.empty((nU, K), np.float32), np.empty((nI, K), np.float32), np.empty((nU, K), np.float32), np.empty((nI, K), np.float32)]
    stats = np.zeros(8)
    invU, invI = np.empty(nU, np.uint8), np.empty(nI, np.uint8)
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    os.environ["MFX_NO_SAVE"] = "1"
    tmp = saved = None
    if log is not None:
        sys.stdout.flush()
        tmp = tempfile.TemporaryFile(mode="w+b")
        saved = os.dup(1)
        os.dup2(tmp.fileno(), 1)
    try:
        rc = lib.mfh_train(method, C.c_int32(tr.nrows), P(tr.rowptr), P(tr.rowind), P(tr.rowval), C.c_int32(tr.ncols),
                           P(va.rowptr), P(va.rowind), P(va.rowval), C.c_int32(va.ncols), P(te.rowptr), P(te.rowind), P(te.rowval),
                           C.c_int32(te.ncols), C.c_int32(K), C.c_int32(cfg["maxIter"]), C.c_int32(1), C.c_float(cfg["lr"]),
                           C.c_float(cfg["ureg"]), C.c_float(cfg["ireg"]), None, P(bufs[0]), P(bufs[1]), P(bufs[2]), P(bufs[3]),
                           P(stats), P(invU), P(invI))
    finally:
        if log is not None:
            C.CDLL(None).fflush(None)
            os.dup2(saved, 1)
            os.close(saved)
            tmp.seek(0)
            text = tmp.read()
            tmp.close()
            os.write(1, text)                   # (still shown where the rest of the trainer output goes)
            log.append(text.decode(errors="replace"))
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if rc != 0:
        raise RuntimeError("mfh_train returned %d" % rc)
    return stats


if __name__ == "__main__":
    main()
