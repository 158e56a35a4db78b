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
