"""User-row-block sharding for the multi-GPU path (the reference is single-process).

Rank g owns a contiguous, nnz-balanced block of user rows (its CSR rows and its U shard, never
communicated) and a full replica of the item factors V.  After a local (sub-)epoch
V <- V_sync + sum_g (V_g - V_sync) is formed with ONE all-reduce (mfx_allreduce_item_factors on
the GPU; the CPU tests run the same algebra over gloo).  Index arithmetic only -- no compute here.
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
