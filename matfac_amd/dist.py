"""User-row-block sharding for the multi-GPU path (the reference is single-process).

Rank g owns a contiguous, nnz-balanced block of user rows (its CSR rows and its U shard, never
communicated) and a full replica of the item factors V.  Two exchanges:
  * rotation (default of bench.py): the reference's own stratification (trainSGDPar, modelMF.cpp:273-304)
    at GPU granularity.  Items are cut into N parts (item % N); an epoch is N sub-epochs, in sub-epoch s
    rank g updates only the ratings of part (g + s) % N -- on the ONLY current copy of that part's rows --
    and hands the part to rank g - 1 (a ring shift of 1/N of V).  No update is lost, summed twice or
    down-weighted; one all-gather after the last sub-epoch completes every replica for the evaluation.
  * all-reduce: after a local epoch V <- V_sync + sum_g (V_g - V_sync) or the mean of the replicas with ONE
    all-reduce (mfx_allreduce_item_factors) -- what north_star names; averaging divides every item step by N.
Index arithmetic and the call sequences only -- no compute here.
"""
import numpy as np

from .synth import CSR


def user_blocks(rowptr, nranks):
    """Boundaries b[0..nranks] of contiguous user blocks with ~equal numbers of ratings."""
    rowptr = np.asarray(rowptr, np.int64)
    nrows = len(rowptr) - 1
    nnz = int(rowptr[-1])
    targets = (np.arange(1, nranks, dtype=np.float64) * nnz / nranks)
    cuts = np.searchsorted(rowptr[1:], targets, side="left") + 1
    b = np.concatenate([[0], np.minimum(cuts, nrows), [nrows]]).astype(np.int64)
    return np.maximum.accumulate(b)


def take_rows(m, lo, hi):
    """Rows [lo, hi) of a CSR as a CSR of its own (local user ids 0..hi-lo, same item ids)."""
    s, e = int(m.rowptr[lo]), int(m.rowptr[hi])
    return CSR(hi - lo, m.ncols, m.rowptr[lo:hi + 1] - s, m.rowind[s:e], m.rowval[s:e])


def delta_sum(V_sync, V_local_list):
    """What the all-reduce computes: V_sync + sum over ranks of (V_g - V_sync), fp32."""
    acc = np.zeros_like(V_sync)
    for Vg in V_local_list:
        acc = acc + (Vg - V_sync)
    return V_sync + acc


def rotation_schedule(rank, nranks):
    """The sub-epochs of one rotating epoch for `rank`: [(part to update, part to send on or None, part to receive or None)],
    and the part this rank holds at the end (its contribution to the closing all-gather).  Across the ranks every (rank, part)
    pair occurs exactly once and no part is updated by two ranks in the same sub-epoch (a Latin square)."""
    steps = []
    for s in range(nranks):
        part = (rank + s) % nranks
        last = s == nranks - 1
        steps.append((part, None if last else part, None if last else (rank + s + 1) % nranks))
    return steps, (rank + nranks - 1) % nranks


def rotating_epoch(ctx, rank, nranks, lr, ureg, ireg, **sgd_kwargs):
    """One epoch of the rotating exchange on an mfx context with mfx_sgd_set_item_parts(nranks) and a communicator set up:
    N part-restricted tiled epochs with a ring shift in between, then the all-gather (V complete and identical on all ranks)."""
    steps, held = rotation_schedule(rank, nranks)
    for part, send, recv in steps:
        ctx.sgd_epoch(lr, ureg, ireg, item_part=part + 1, **sgd_kwargs)
        if send is not None:
            ctx.rotate_item_part(send, recv)
    ctx.allgather_item_parts(held)
