"""The arithmetic of the CCD++ block loop (matfac_amd/csrc/ccd_blocks.h) restated in numpy, against the sequential sums of
modelMF.cpp:1066-1073: float products, double accumulation, one rounding of the quotient to float.

The device forms a row's (num, den) from pieces: per lane the sum of eight consecutive (padded) entries, an inclusive scan over the
sixteen lanes of a 128-entry trip, the DIFFERENCE of two prefixes for a piece that starts inside the trip, and the pieces of a row added
in order.  The claim DESIGN.md 3.4 makes -- that the differences of double prefixes over at most 128 float products do not move the
float the quotient is rounded to -- is checked here on rows of every length class, with cancelling products (ratings times factors of
both signs), without a GPU."""
import numpy as np
import pytest


def _sequential(val, v, ind, rp, reg):
    out = np.zeros(len(rp) - 1, np.float32)
    for r in range(len(rp) - 1):
        b, e = rp[r], rp[r + 1]
        if e == b:
            continue
        o = v[ind[b:e]]
        p = (val[b:e] * o).astype(np.float32).astype(np.float64)
        q = (o * o).astype(np.float32).astype(np.float64)
        num = den = 0.0
        for x, y in zip(p, q):
            num += x
            den += y
        out[r] = np.float32(num / (np.float64(reg) + den))
    return out


def _blocks(val, v, ind, rp, reg):
    nrows, nI = len(rp) - 1, len(v)
    lens = np.diff(rp)
    rpos = np.concatenate([[0], np.cumsum((lens + 7) // 8 * 8)])
    nnzp = (rpos[-1] + 127) // 128 * 128
    R = np.zeros(nnzp, np.float32)
    I = np.full(nnzp, nI, np.int64)                        # padding entries: the +0.0 slot behind the vector
    for r in range(nrows):
        R[rpos[r]:rpos[r] + lens[r]] = val[rp[r]:rp[r + 1]]
        I[rpos[r]:rpos[r] + lens[r]] = ind[rp[r]:rp[r + 1]]
    o = np.concatenate([v, [np.float32(0)]]).astype(np.float32)[I]
    pn = (R * o).astype(np.float32).astype(np.float64)
    pd = (o * o).astype(np.float32).astype(np.float64)
    ln, ld = pn[0::8].copy(), pd[0::8].copy()              # a lane's eight entries, in order
    for q in range(1, 8):
        ln = ln + pn[q::8]
        ld = ld + pd[q::8]
    Pn, Pd = ln.reshape(-1, 16).copy(), ld.reshape(-1, 16).copy()
    for s in (1, 2, 4, 8):                                  # the four row_shr levels of the scan
        for P in (Pn, Pd):
            sh = np.zeros_like(P)
            sh[:, s:] = P[:, :-s]
            P += sh
    out = np.zeros(nrows, np.float32)
    for r in range(nrows):
        if lens[r] == 0:
            continue
        fl, el = rpos[r] // 8, rpos[r + 1] // 8 - 1
        num = den = None
        for t in range(fl >> 4, (el >> 4) + 1):             # the row's pieces, one per trip it touches
            a = fl & 15 if t == fl >> 4 else 0
            b = el & 15 if t == el >> 4 else 15
            n_ = Pn[t, b] - (Pn[t, a - 1] if a > 0 else 0.0)
            d_ = Pd[t, b] - (Pd[t, a - 1] if a > 0 else 0.0)
            num, den = (n_, d_) if num is None else (num + n_, den + d_)
        out[r] = np.float32(num / (np.float64(reg) + den))
    return out


@pytest.mark.parametrize("seed,scale", [(1, 1.0), (2, 30.0), (3, 1e-3)])
def test_block_sums_round_to_the_sequential_quotients(seed, scale):
    rng = np.random.default_rng(seed)
    nI = 300
    lens = np.concatenate([rng.integers(0, 3, 40), rng.integers(1, 40, 200), rng.integers(100, 400, 40), [1500, 4099, 0, 8, 16, 128, 129]])
    rng.shuffle(lens)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ind = rng.integers(0, nI, rp[-1])
    val = (rng.integers(1, 11, rp[-1]) * 0.5).astype(np.float32)
    v = (rng.normal(0, 1, nI) * scale).astype(np.float32)   # both signs: the sums cancel
    a = _sequential(val, v, ind, rp, np.float32(0.3))
    b = _blocks(val, v, ind, rp, np.float32(0.3))
    ai, bi = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    assert np.abs(ai - bi).max() <= 1                       # one ulp at most ...
    assert (ai != bi).mean() <= 0.01                        # ... and almost never at all
