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
