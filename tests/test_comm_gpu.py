"""RCCL path on the one GPU of the test box: a 1-rank communicator exercises librccl loading,
ncclCommInitRank, the delta kernels and ncclAllReduce.  (RCCL refuses two ranks on one device, so
world_size 2 is covered by the gloo test in test_dist_cpu.py and by the driver's multi-GPU run.)"""
import numpy as np
import pytest

from matfac_amd import Ctx, mfx
from oracle import binding as orc
from tests.util import load_ctx, small

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_roundtrip():
    d = small(K=64)
    K = 64
    U0, V0 = orc.init_factors(1, d["nUsers"], d["nItems"], K)
    uid = Ctx.comm_unique_id()
    assert len(uid) == 128
    with Ctx(0) as ctx:
        load_ctx(ctx, d, K, U0, V0)
        ctx.comm_init(1, 0, uid)
        ctx.comm_mark_synced()
        ctx.sgd_epoch(0.01, 0.01, 0.01, mode=mfx.SGD_TILED, seed=1, epoch=0)
        _, V1 = ctx.get_factors()
        ctx.allreduce_item_factors(mfx.REDUCE_DELTA_SUM)      # V_sync + (V - V_sync)
        _, V2 = ctx.get_factors()
        assert not np.array_equal(V1, V0)
        assert np.allclose(V2, V1, rtol=0, atol=1e-7)
        ctx.allreduce_item_factors(mfx.REDUCE_AVERAGE)        # mean over one rank
        _, V3 = ctx.get_factors()
        assert np.array_equal(V3, V2)
        s = ctx.allreduce_f64([1.5, 2.5])
        assert s.tolist() == [1.5, 2.5]
        ctx.comm_destroy()


def test_single_rank_rccl_reduce_scatter_allgather_and_ring_shift(monkeypatch):
    """The RCCL collectives the N > 1 paths use, on a 1-rank communicator (all this box can run): the sharded ALS item half-sweep
    -- ncclReduceScatter of the (A, b) slab, solve of the rank's item slice, ncclAllGather of the solved rows -- must equal the
    unsharded sweep bit for bit; the rotation's ring shift (ncclSend / ncclRecv in one group, rank 0 to itself:
    MFX_COMM_SELF_TEST=1) and its closing ncclAllGather must hand V back unchanged."""
    monkeypatch.setenv("MFX_COMM_SELF_TEST", "1")
    d = small(nU=900, nI=257, nnz=40000, K=64, seed=17)       # 257 items: a last slice that is short
    K = 64
    U0, V0 = orc.init_factors(1, d["nUsers"], d["nItems"], K)
    U0 *= 30
    V0 *= 30
    with Ctx(0) as ctx:
        load_ctx(ctx, d, K, U0, V0)
        ctx.als_half_sweep(mfx.SIDE_ITEMS, 3.0)
        _, Vref = ctx.get_factors()
        uid = Ctx.comm_unique_id()
        ctx.comm_init(1, 0, uid)
        ctx.set_factors(U0, V0)
        ctx.als_half_sweep(mfx.SIDE_ITEMS, 3.0)                # sharded path: reduce-scatter, slice solve, all-gather
        _, Vsh = ctx.get_factors()
        assert np.array_equal(Vsh, Vref)
        assert not np.array_equal(Vref, V0)
        ctx.rotate_item_part(0, 0)                             # part 0 of 1 = every row: packed, sent round the ring of one, unpacked
        ctx.allgather_item_parts(0)
        _, V2 = ctx.get_factors()
        assert np.array_equal(V2, Vsh)
        ctx.comm_destroy()
