"""Multi-process (gloo, world_size 2, CPU) test of the N > 1 path: user-row-block sharding and the
item-factor delta all-reduce.  The local epochs are run by the CPU oracle here (no GPU in this
container); the GPU build runs the identical algebra in mfx_allreduce_item_factors over RCCL."""
import os
import socket

import numpy as np
import pytest

from matfac_amd import dist as mdist
from matfac_amd import synth
from oracle import binding as orc

K, LR, REG = 8, 0.01, 0.02


def _problem():
    d = synth.make(dict(nU=400, nI=150, nnz=9000, K=K), seed=6)
    return d["train"], d["nUsers"], d["nItems"]


def _local_epoch(tr, lo, hi, U, V, seed):
    sh = mdist.take_rows(tr, lo, hi)
    order = np.arange(sh.nnz, dtype=np.uint64)
    orc.MT(seed).shuffle_u64(order)
    Ul = U[lo:hi].copy()
    Vl = V.copy()
    orc.sgd_pass(Ul, Vl, sh.rowids(), sh.rowind, sh.rowval, order, LR, REG, REG)
    return Ul, Vl


def test_user_blocks_partition_and_balance():
    tr, nU, nI = _problem()
    for n in (1, 2, 3, 8):
        b = mdist.user_blocks(tr.rowptr, n)
        assert b[0] == 0 and b[-1] == tr.nrows and np.all(np.diff(b) >= 0) and len(b) == n + 1
        per = np.diff(tr.rowptr[b])
        assert per.sum() == tr.nnz
        assert per.max() <= tr.nnz / n + np.diff(tr.rowptr).max()       # balanced up to one row
        got = np.concatenate([mdist.take_rows(tr, b[g], b[g + 1]).rowind for g in range(n)])
        assert np.array_equal(got, tr.rowind)


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, nU, nI = _problem()
    U, V = orc.init_factors(1, nU, nI, K)
    U *= 20
    V *= 20
    b = mdist.user_blocks(tr.rowptr, world)
    V_sync = V.copy()
    for ep in range(3):
        Ul, Vl = _local_epoch(tr, b[rank], b[rank + 1], U, V, seed=100 * ep + rank)
        U[b[rank]:b[rank + 1]] = Ul
        delta = torch.from_numpy(Vl - V_sync)
        dist.all_reduce(delta, op=dist.ReduceOp.SUM)        # the one exchange step of the path
        V = V_sync + delta.numpy()
        V_sync = V.copy()
    np.save(os.path.join(out_dir, "V%d.npy" % rank), V)
    np.save(os.path.join(out_dir, "U%d.npy" % rank), U[b[rank]:b[rank + 1]])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_delta_allreduce_matches_single_process_simulation(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    # single-process simulation of the same schedule
    tr, nU, nI = _problem()
    U, V = orc.init_factors(1, nU, nI, K)
    U *= 20
    V *= 20
    b = mdist.user_blocks(tr.rowptr, world)
    for ep in range(3):
        locals_ = []
        for g in range(world):
            Ul, Vl = _local_epoch(tr, b[g], b[g + 1], U, V, seed=100 * ep + g)
            locals_.append((Ul, Vl))
        for g in range(world):
            U[b[g]:b[g + 1]] = locals_[g][0]
        V = mdist.delta_sum(V, [vl for _, vl in locals_])
    V0 = np.load(tmp_path / "V0.npy")
    V1 = np.load(tmp_path / "V1.npy")
    assert np.array_equal(V0, V1)                         # replicas agree bit for bit after the exchange
    assert np.allclose(V0, V, rtol=0, atol=1e-6)          # fp32 sum order of the two deltas may differ
    for g in range(world):
        assert np.array_equal(np.load(tmp_path / ("U%d.npy" % g)), U[b[g]:b[g + 1]])
    # and the sharded result is a sensible SGD step: training error went down
    invU, invI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
    U0, V00 = orc.init_factors(1, nU, nI, K)
    r0, _, _ = orc.rmse(U0 * 20, V00 * 20, nU, nI, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, invI)
    r1, _, _ = orc.rmse(U, V, nU, nI, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, invI)
    assert r1 < r0


# ---- the rotating exchange (item parts handed round a ring): matfac_amd.dist.rotation_schedule ------------------------------
def test_rotation_schedule_is_a_latin_square():
    for n in (1, 2, 3, 8):
        sched = [mdist.rotation_schedule(g, n) for g in range(n)]
        for s in range(n):
            parts = [sched[g][0][s][0] for g in range(n)]
            assert sorted(parts) == list(range(n))                     # a sub-epoch: every part on exactly one rank
            for g in range(n):
                part, send, recv = sched[g][0][s]
                if s < n - 1:
                    assert send == part and recv == sched[g][0][s + 1][0]
                    assert sched[(g + 1) % n][0][s][1] == recv         # what arrives is what the next rank sends on
                else:
                    assert send is None and recv is None
        for g in range(n):
            assert sorted(st[0] for st in sched[g][0]) == list(range(n))   # a rank sees every part once per epoch
            assert sched[g][1] == sched[g][0][-1][0]                   # it ends holding the part of its last sub-epoch
        assert sorted(sched[g][1] for g in range(n)) == list(range(n))


def _rot_problem():
    d = synth.make(dict(nU=2000, nI=600, nnz=150_000, K=K), seed=21)
    return d["train"], d["val"], d["nUsers"], d["nItems"]


def _part_lists(sh, nparts):
    """ratings of a user block grouped by item part (item % nparts), CSR order inside a part"""
    u, i, r = sh.rowids(), sh.rowind, sh.rowval
    out = []
    for p in range(nparts):
        sel = np.nonzero(i % nparts == p)[0]
        out.append((u[sel].astype(np.int32), i[sel].astype(np.int32), r[sel].astype(np.float32)))
    return out


def _rot_worker(rank, world, port, out_dir, epochs):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, va, nU, nI = _rot_problem()
    U, V = orc.init_factors(1, nU, nI, K)
    b = mdist.user_blocks(tr.rowptr, world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    lists = _part_lists(mdist.take_rows(tr, lo, hi), world)
    Ul = U[lo:hi].copy()
    steps, held = mdist.rotation_schedule(rank, world)
    for ep in range(epochs):
        for part, send, recv in steps:
            u, i, r = lists[part]
            order = np.arange(u.size, dtype=np.uint64)
            orc.MT(1000 * ep + 10 * part + rank).shuffle_u64(order)
            orc.sgd_pass(Ul, V, u, i, r, order, LR, REG, REG)          # only rows of `part` of V are touched
            if send is not None:                                       # ring shift: rows of `send` to rank - 1, rows of `recv` from rank + 1
                out_rows = torch.from_numpy(np.ascontiguousarray(V[send::world]))
                in_rows = torch.empty((len(range(recv, nI, world)), K), dtype=torch.float32)
                reqs = [dist.isend(out_rows, (rank - 1) % world), dist.irecv(in_rows, (rank + 1) % world)]
                for q in reqs:
                    q.wait()
                V[recv::world] = in_rows.numpy()
        # closing all-gather: rank r contributes part (held - rank + r) % world
        mine = torch.from_numpy(np.ascontiguousarray(V[held::world]))
        rows = (nI + world - 1) // world
        pad = torch.zeros((rows, K), dtype=torch.float32)
        pad[: mine.shape[0]] = mine
        got = [torch.zeros((rows, K), dtype=torch.float32) for _ in range(world)]
        dist.all_gather(got, pad)
        for r_ in range(world):
            p = (held - rank + r_) % world
            V[p::world] = got[r_][: len(range(p, nI, world))].numpy()
    np.save(os.path.join(out_dir, "rotV%d.npy" % rank), V)
    np.save(os.path.join(out_dir, "rotU%d.npy" % rank), Ul)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_rotation_equals_its_sequential_statement_and_converges_like_one_rank(tmp_path):
    """World size 2 over gloo: the rotating exchange run by two processes (local sub-epochs by the CPU oracle) is bit for bit the
    single-process statement of the same schedule -- in a sub-epoch the ranks touch disjoint user rows AND disjoint item rows --
    and after E epochs its validation RMSE is the one-rank run's (every rating visited once per epoch with its full step),
    while averaging the replicas (MFX_REDUCE_AVERAGE) halves every item step and lags."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world, epochs = 2, 12
    mp.spawn(_rot_worker, args=(world, port, str(tmp_path), epochs), nprocs=world, join=True)
    tr, va, nU, nI = _rot_problem()
    b = mdist.user_blocks(tr.rowptr, world)
    U, V = orc.init_factors(1, nU, nI, K)
    lists = [_part_lists(mdist.take_rows(tr, int(b[g]), int(b[g + 1])), world) for g in range(world)]
    sched = [mdist.rotation_schedule(g, world)[0] for g in range(world)]
    for ep in range(epochs):
        for s_ in range(world):
            for g in range(world):
                part = sched[g][s_][0]
                u, i, r = lists[g][part]
                order = np.arange(u.size, dtype=np.uint64)
                orc.MT(1000 * ep + 10 * part + g).shuffle_u64(order)
                Ug = U[int(b[g]):int(b[g + 1])]
                orc.sgd_pass(Ug, V, u, i, r, order, LR, REG, REG)
    V0, V1 = np.load(tmp_path / "rotV0.npy"), np.load(tmp_path / "rotV1.npy")
    assert np.array_equal(V0, V1) and np.array_equal(V0, V)
    for g in range(world):
        assert np.array_equal(np.load(tmp_path / ("rotU%d.npy" % g)), U[int(b[g]):int(b[g + 1])])
    invU, invI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
    rot, _, _ = orc.rmse(U, V, nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, invU, invI)
    # one rank, the same number of epochs over a freshly shuffled list, three shuffle seeds: its own seed-to-seed envelope
    ones = []
    for seed in (7, 8, 9):
        U1, V1s = orc.init_factors(1, nU, nI, K)
        order = np.arange(tr.nnz, dtype=np.uint64)
        mt = orc.MT(seed)
        for ep in range(epochs):
            mt.shuffle_u64(order)
            orc.sgd_pass(U1, V1s, tr.rowids(), tr.rowind, tr.rowval, order, LR, REG, REG)
        ones.append(orc.rmse(U1, V1s, nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, invU, invI)[0])
    # the same two shards with the replicas of V averaged after every local epoch
    Ua, Va = orc.init_factors(1, nU, nI, K)
    for ep in range(epochs):
        reps = []
        for g in range(world):
            sh = mdist.take_rows(tr, int(b[g]), int(b[g + 1]))
            o = np.arange(sh.nnz, dtype=np.uint64)
            orc.MT(1000 * ep + g).shuffle_u64(o)
            Vg = Va.copy()
            orc.sgd_pass(Ua[int(b[g]):int(b[g + 1])], Vg, sh.rowids(), sh.rowind, sh.rowval, o, LR, REG, REG)
            reps.append(Vg)
        Va = (reps[0] + reps[1]) / np.float32(world)
    avg, _, _ = orc.rmse(Ua, Va, nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, invU, invI)
    print("validation RMSE after %d epochs: one rank %s (three shuffle seeds), two ranks rotating %.5f, two ranks averaging %.5f"
          % (epochs, np.round(ones, 5), rot, avg))
    # rotating: inside the one-rank run's own seed-to-seed envelope (+- 1e-3); averaging: far outside it (every item step halved)
    assert min(ones) - 1e-3 <= rot <= max(ones) + 1e-3
    assert avg > max(ones) + 5 * (max(ones) - min(ones))
