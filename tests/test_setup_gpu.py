"""One-time preprocessing on the device (setup.hip) against its host statements: the column view against the
oracle's gk_csr_CreateIndex restatement, the slot lists of MFX_SGD_TILED against the host builder
(MFX_SLOTS_HOST=1).  Integer work: bit-exact."""
import os

import numpy as np
import pytest

from matfac_amd import Ctx, mfx, synth
from oracle import binding as orc

pytestmark = pytest.mark.gpu


def small(nU, nI, nnz, seed):
    d = synth.make(dict(nU=nU, nI=nI, nnz=nnz, K=8), seed=seed)
    return d["train"], d["nUsers"], max(d["nItems"], d["train"].ncols)


@pytest.mark.parametrize("nU,nI,nnz", [(7, 5, 20), (300, 70, 4000), (20000, 3000, 600000), (3000, 50000, 200000)])
def test_device_column_view_is_the_reference_counting_sort(nU, nI, nnz):
    tr, nU, nI = small(nU, nI, nnz, 3)
    cp, ci, cv = orc.create_col_index(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)       # no column view given
        gp, gi, gv = ctx.debug_col_view(tr.ncols, tr.nnz)
    assert np.array_equal(gp, cp) and np.array_equal(gi, ci) and np.array_equal(gv, cv)


def digest(tr, nU, nI, K, host, own=0):
    if host:
        os.environ["MFX_SLOTS_HOST"] = "1"
    else:
        os.environ.pop("MFX_SLOTS_HOST", None)
    try:
        with Ctx(0) as ctx:
            ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
            ctx.set_model(nU, nI, K)
            ctx.sgd_epoch(0.0, 0.0, 0.0, mode=mfx.SGD_TILED, seed=5, epoch=0, own=own)
            counts, sums = ctx.debug_slots_digest()
            u, i, r = ctx.debug_epoch_list()
        return counts, sums, (u, i, r)
    finally:
        os.environ.pop("MFX_SLOTS_HOST", None)


@pytest.mark.parametrize("nU,nI,nnz,K,own", [(7, 5, 20, 4, 0), (300, 70, 4000, 8, 0), (20000, 3000, 600000, 64, 0),
                                             (20000, 3000, 600000, 128, 0), (3000, 50000, 200000, 32, 0),
                                             (20000, 3000, 600000, 64, 1), (4000, 40, 150000, 16, 0)])
def test_device_slot_lists_equal_the_host_builder(nU, nI, nnz, K, own):
    tr, nU, nI = small(nU, nI, nnz, 11)          # (4000 x 40: every item is "popular" in its tiles)
    ch, sh, lh = digest(tr, nU, nI, K, host=True, own=own)
    cd, sd, ld = digest(tr, nU, nI, K, host=False, own=own)
    assert ch == cd, (ch, cd)
    assert sh == sd
    for a, b in zip(lh, ld):
        assert np.array_equal(a, b)
    # and the list is the train matrix, every rating once
    order = np.lexsort((ld[1], ld[0]))
    ref = np.lexsort((tr.rowind, tr.rowids()))
    assert np.array_equal(ld[0][order], tr.rowids()[ref]) and np.array_equal(ld[1][order], tr.rowind[ref])
    assert np.array_equal(ld[2][order], tr.rowval[ref])


@pytest.mark.parametrize("n", [2, 3, 17, 1000, 70001, (1 << 20) + 3])
def test_swaps_of_std_shuffle_applied_on_the_device(n):
    """mfx_sgd_apply_swaps32: the list of mfx_sgd_set_order32 after `for i in 1 .. n-1: swap(a[i], a[pos[i]])` (pos[i] <= i), without
    walking the swaps one by one (one stable sort of the steps by the place they hit, pointer jumping along the forward chains): the
    same list as the plain loop, for random positions, for the worst chains (every step hits place 0; every step hits the place
    before it; no step moves anything) and applied twice in a row to the list it keeps."""
    from matfac_amd import Ctx, mfx, synth

    def loop(a, pos):
        a = a.copy()
        for i in range(1, a.size):
            j = pos[i]
            a[i], a[j] = a[j], a[i]
        return a
    rng = np.random.default_rng(n)
    cases = [np.array([0] + [int(rng.integers(0, i + 1)) for i in range(1, n)], np.uint32),
             np.zeros(n, np.uint32),                                            # every step swaps with place 0
             np.maximum(np.arange(n, dtype=np.int64) - 1, 0).astype(np.uint32),  # every step swaps with its left neighbour
             np.arange(n, dtype=np.uint32)]                                      # nothing moves
    d = synth.make(dict(nU=50, nI=40, nnz=500, K=8), seed=1)
    tr = d["train"]
    with Ctx(0) as ctx:
        ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, tr.ncols, tr.rowptr, tr.rowind, tr.rowval)
        for pos in cases if n <= 70001 else cases[:1]:
            a0 = rng.permutation(n).astype(np.uint32)
            ctx.sgd_set_order32(a0)
            ctx.sgd_apply_swaps32(pos)
            want = loop(a0, pos)
            got = ctx.debug_order32()
            assert np.array_equal(got, want)
            pos2 = np.array([0] + [int(rng.integers(0, i + 1)) for i in range(1, n)], np.uint32) if n <= 70001 else pos
            ctx.sgd_apply_swaps32(pos2)                                          # the next epoch shuffles the SAME list again
            assert np.array_equal(ctx.debug_order32(), loop(want, pos2))
