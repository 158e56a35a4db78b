"""The N > 1 path on the device: two processes, each with its own context and its block of user rows, the
exchanges carried by gloo through mfx_comm_init_external (the same library code runs over RCCL when
mfx_comm_init made the communicator).  Sharded CCD++ / ALS / SGD against the one-context run of the same work.

Round 4: every test here also exists in an RCCL form -- one process per GPU, mfx_comm_init with a broadcast unique id, the
library's own ncclAllReduce / ncclReduceScatter / ncclAllGather / ncclSend + ncclRecv -- that SKIPS unless mfx_device_count()
reports two devices: the first box with two GPUs exercises comm.hip beyond a one-rank communicator (tests/test_comm_gpu.py),
with the same expectations as the gloo forms."""
import os
import socket

import numpy as np
import pytest

from matfac_amd import Ctx, mfx, synth
from matfac_amd import dist as mdist
from oracle import binding as orc

pytestmark = pytest.mark.gpu
K = 16


def _n_devices():
    import ctypes as C
    from matfac_amd import _lib
    n = C.c_int(0)
    return int(n.value) if _lib.load().mfx_device_count(C.byref(n)) == 0 else 0


def _open(rank, world, transport):
    """(context, torch.distributed) of one rank: gloo + mfx_comm_init_external on device 0, or the library's RCCL communicator on
    device `rank` (gloo only carries the unique id and the closing barrier)."""
    import torch
    import torch.distributed as dist
    ctx = Ctx(rank if transport == "rccl" else 0)
    if transport == "rccl":
        uid = [Ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(world, rank, uid[0])
    else:
        ctx.comm_init_external(world, rank, lambda a: dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM))
    return ctx


def _problem():
    d = synth.make(dict(nU=3000, nI=400, nnz=150000, K=K), seed=12)
    tr = d["train"]
    nI = max(d["nItems"], tr.ncols)
    U0, V0 = synth.init_factors(3, tr.nrows, nI, K)
    return tr, nI, (U0 * 30).astype(np.float32), (V0 * 30).astype(np.float32)


def _run(ctx, nI, tr, U0, V0, exchange):
    """The work both layouts do; `exchange` is called where the sharded run needs the item factors agreed."""
    out = {}
    ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
    ctx.set_model(tr.nrows, nI, K)
    invU, invI = ctx.compute_invalid()
    out["invI"] = invI.copy()
    # CCD++ with FreqAdap (modelMF.cpp:1272-1360): 2 outer iterations
    ctx.set_factors(U0, V0)
    ctx.ccdpp_begin()
    for it in range(2):
        for k in range(K):
            ctx.ccdpp_rank1(k, 0.4, 0.6, add_back=it > 0, freq_thresh=75.0)
    ctx.ccdpp_end()
    out["ccd_U"], out["ccd_V"] = ctx.get_factors()
    e = ctx.eval(mfx.MAT_TRAIN)
    out["ccd_sse"] = np.array([e.sse, float(e.n)])
    # ALS: 2 iterations
    ctx.set_factors(U0, V0)
    for it in range(2):
        ctx.als_half_sweep(mfx.SIDE_USERS, 2.0)
        ctx.als_half_sweep(mfx.SIDE_ITEMS, 3.0)
    out["als_U"], out["als_V"] = ctx.get_factors()
    # SGD: CSR order on one group (deterministic), one exchange per epoch
    ctx.set_factors(U0, V0)
    exchange("mark")
    for ep in range(2):
        ctx.sgd_epoch(0.002, 0.05, 0.05, mode=mfx.SGD_SERIAL, order=mfx.ORDER_NATURAL, arith=mfx.ARITH_REF64)
        exchange("sum")
    out["sgd_U"], out["sgd_V"] = ctx.get_factors()
    return out


def _worker(rank, world, port, out_dir, transport="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, nI, U0, V0 = _problem()
    b = mdist.user_blocks(tr.rowptr, world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    sh = mdist.take_rows(tr, lo, hi)
    with _open(rank, world, transport) as ctx:

        def exchange(what):
            if what == "mark":
                ctx.comm_mark_synced()
            else:
                ctx.allreduce_item_factors(mfx.REDUCE_DELTA_SUM)
        out = _run(ctx, nI, sh, U0[lo:hi], V0, exchange)
        out["ccd_sse_all"] = ctx.allreduce_f64(out["ccd_sse"])
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), lo=lo, hi=hi, **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["gloo", "rccl"])
def test_two_shards_equal_one_context(tmp_path, transport):
    if transport == "rccl" and _n_devices() < 2:
        pytest.skip("the RCCL form needs two GPUs (mfx_device_count() = %d)" % _n_devices())
    # stdlib multiprocessing: torch (and the HIP/RCCL copies it bundles) is loaded in the two workers only, never
    # into this process, which already holds libmfx.so and possibly the system RCCL from other tests
    import multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cx = mp.get_context("spawn")
    procs = [cx.Process(target=_worker, args=(g, 2, port, str(tmp_path), transport)) for g in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    r = [np.load(str(tmp_path / ("r%d.npz" % g))) for g in range(2)]
    tr, nI, U0, V0 = _problem()
    with Ctx(0) as ctx:
        one = _run(ctx, nI, tr, U0, V0, lambda what: None)
    assert r[0]["hi"] == r[1]["lo"] and r[0]["lo"] == 0 and r[1]["hi"] == tr.nrows
    # an item rated only by the other rank's users is valid on both
    assert np.array_equal(r[0]["invI"], one["invI"]) and np.array_equal(r[1]["invI"], one["invI"])
    cat = lambda key: np.concatenate([r[0][key], r[1][key]])
    for algo, tol in (("ccd", 2e-5), ("als", 5e-4)):
        scale = float(np.abs(one[algo + "_V"]).max())
        assert np.array_equal(r[0][algo + "_V"], r[1][algo + "_V"]), algo           # replicas agree bit for bit
        assert np.abs(r[0][algo + "_V"] - one[algo + "_V"]).max() < tol * scale, algo
        assert np.abs(cat(algo + "_U") - one[algo + "_U"]).max() < tol * max(scale, float(np.abs(one[algo + "_U"]).max())), algo
    assert np.allclose(r[0]["ccd_sse_all"], r[1]["ccd_sse_all"])
    assert r[0]["ccd_sse_all"][1] == tr.nnz and abs(r[0]["ccd_sse_all"][0] - one["ccd_sse"][0]) < 1e-4 * one["ccd_sse"][0]
    # SGD: each rank sweeps its rows in CSR order against its own copy of V, then V <- V_sync + sum of deltas
    U, V = U0.copy(), V0.copy()
    for ep in range(2):
        parts = []
        for g in range(2):
            lo, hi = int(r[g]["lo"]), int(r[g]["hi"])
            sh = mdist.take_rows(tr, lo, hi)
            Ul, Vl = U[lo:hi].copy(), V.copy()
            orc.sgd_pass(Ul, Vl, sh.rowids(), sh.rowind, sh.rowval, None, 0.002, 0.05, 0.05, orc.ARITH_REF64, orc.DOT_TREE)
            U[lo:hi] = Ul
            parts.append(Vl)
        V = mdist.delta_sum(V, parts)
    assert np.array_equal(r[0]["sgd_V"], r[1]["sgd_V"])
    assert np.array_equal(cat("sgd_U"), U) and np.array_equal(r[0]["sgd_V"], V)


# ---- the rotating exchange through the library (mfx_sgd_set_item_parts / mfx_rotate_item_part / mfx_allgather_item_parts) ----
def _rot_problem():
    """40 000 ratings of pairwise distinct users over 300 items: every item row is updated ~130 times, no user row twice -- so
    that with ONE lane group per slot the visiting order alone determines the result (the kernel requests the lock-free rows
    of the next two steps ahead of time: a user met twice in a row would see its row one update late, by design)."""
    n, nI = 40000, 300
    rng = np.random.default_rng(31)
    tr = synth.CSR(n, nI, np.arange(n + 1, dtype=np.int64), rng.integers(0, nI, n).astype(np.int32),
                   (rng.integers(1, 11, n) * 0.5).astype(np.float32))
    U0 = rng.normal(0, 0.3, (n, K)).astype(np.float32)
    V0 = rng.normal(0, 0.3, (nI, K)).astype(np.float32)
    return tr, nI, U0, V0


def _rot_worker(rank, world, port, out_dir, one_group, transport="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, nI, U0, V0 = _rot_problem()
    b = mdist.user_blocks(tr.rowptr, world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    sh = mdist.take_rows(tr, lo, hi)
    lists = {}
    with _open(rank, world, transport) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, sh.nrows, nI, sh.rowptr, sh.rowind, sh.rowval)
        ctx.set_model(sh.nrows, nI, K)
        ctx.set_factors(U0[lo:hi], V0)
        ctx.compute_invalid()
        ctx.set_item_parts(world)
        steps, held = mdist.rotation_schedule(rank, world)
        visits = 0
        for ep in range(1):
            for part, send, recv in steps:
                flags = mfx.SGD_F_COUNT_VISITS | (mfx.SGD_F_ONE_GROUP if one_group else 0)
                ctx.sgd_epoch(0.002, 0.05, 0.05, mode=mfx.SGD_TILED, order=mfx.ORDER_DEVICE, arith=mfx.ARITH_REF64, seed=5, epoch=ep,
                              flags=flags, item_part=part + 1)
                v = ctx.debug_visit_counts()
                assert np.all(v == 1)
                visits += int(v.size)
                u, i, r = ctx.debug_epoch_list()
                assert np.all(i % world == part)                       # only ratings of the part
                lists["e%d_p%d" % (ep, part)] = np.stack([u, i, r.view(np.int32)])
                if send is not None:
                    ctx.rotate_item_part(send, recv)
            ctx.allgather_item_parts(held)
        assert visits == sh.nnz                                        # every rating of the block once per epoch
        U, V = ctx.get_factors()
    np.savez(os.path.join(out_dir, "rot%d.npz" % rank), lo=lo, hi=hi, U=U, V=V, **lists)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["gloo", "rccl"])
@pytest.mark.parametrize("one_group", [True, False])
def test_two_ranks_rotating_item_parts_through_the_library(tmp_path, one_group, transport):
    """Two processes, one GPU, gloo behind mfx_comm_init_external: an epoch = 2 part-restricted tiled epochs with a ring shift in
    between and the closing all-gather.  Every rating is visited exactly once per epoch (counted by the kernel), a sub-epoch
    touches only its part's items, both replicas of V agree bit for bit afterwards -- and with ONE lane group per slot (the
    deterministic test mode: the list mfx_debug_epoch_list returns IS the visiting order) the factors are the oracle's
    sequential replay of the recorded lists, sub-epoch by sub-epoch, rank by rank."""
    if transport == "rccl" and _n_devices() < 2:
        pytest.skip("the RCCL form (ncclSend / ncclRecv ring shift, ncclAllGather) needs two GPUs (mfx_device_count() = %d)" % _n_devices())
    import multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cx = mp.get_context("spawn")
    procs = [cx.Process(target=_rot_worker, args=(g, 2, port, str(tmp_path), one_group, transport)) for g in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    r = [np.load(str(tmp_path / ("rot%d.npz" % g))) for g in range(2)]
    assert np.array_equal(r[0]["V"], r[1]["V"])                        # complete and identical after the all-gather
    tr, nI, U0, V0 = _rot_problem()
    assert np.isfinite(r[0]["V"]).all() and np.isfinite(r[0]["U"]).all() and np.isfinite(r[1]["U"]).all()
    assert np.abs(r[0]["V"] - V0).max() > 1e-3
    if not one_group:
        return
    U, V = U0.copy(), V0.copy()
    for ep in range(1):
        for s_ in range(2):
            for g in range(2):
                part = mdist.rotation_schedule(g, 2)[0][s_][0]
                u, i, rb = r[g]["e%d_p%d" % (ep, part)]
                lo = int(r[g]["lo"])
                Ug = U[lo:int(r[g]["hi"])]
                orc.sgd_pass(Ug, V, u.astype(np.int32), i.astype(np.int32), rb.view(np.float32), None, 0.002, 0.05, 0.05, orc.ARITH_REF64,
                             orc.DOT_TREE)
    # (the owned item rows are accumulated in 2^-24 fixed point in LDS: the random walk of their roundings, as in
    # tests/test_sgd_gpu.py's one-group replay)
    print("rotation vs sequential replay: max |dV| %.3g" % np.abs(r[0]["V"] - V).max())
    assert np.abs(V - V0).max() > 0.05
    assert np.abs(r[0]["V"] - V).max() < 5e-6
    for g in range(2):
        assert np.abs(r[g]["U"] - U[int(r[g]["lo"]):int(r[g]["hi"])]).max() < 5e-6
