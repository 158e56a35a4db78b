"""What the VISITING ORDER alone does to the converged model (no concurrency at all): the oracle's sequential pass
(modelMF.cpp:83-105 arithmetic, hogTrain's float bracket) over differently structured epoch lists on the `mid` fixture problem
(tests/golden/sgd_spread_mid.json), with the termination rule reduced to "best validation RMSE -> test RMSE".
Test infrastructure (uses the oracle).   ORDERS=uniform,tile_slots,tile_shuffled python tests/tools/order_effect.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from matfac_amd import synth
from oracle import binding as orc

which = os.environ.get("WHICH", "mid")
f = json.load(open(os.path.join(ROOT, "tests", "golden", "sgd_spread_%s.json" % which)))
cfg = f["config"]
shape = dict(synth.SHAPES[cfg["shape"]]) if isinstance(cfg["shape"], str) else dict(cfg["shape"])
shape["nnz"] = int(shape["nnz"] / 0.8)
d = synth.make(shape, seed=cfg["data_seed"])
tr, va, te = d["train"], d["val"], d["test"]
nU, nI, K = d["nUsers"], d["nItems"], cfg["K"]
ru = tr.rowids(); ci = tr.rowind; rv = tr.rowval
lr, ureg, ireg = cfg["lr"], cfg["ureg"], cfg["ireg"]
invU, invI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
EPOCHS = int(os.environ.get("EPOCHS", "60"))
NB = int(os.environ.get("NB", "8"))     # item blocks (= tiles per round)
NBU = int(os.environ.get("NBU", str(NB)))   # user blocks (NBU / NB rounds per item-block offset, as -DMFX_SUB)


def blocks_of(n, rng, nb):
    return rng.integers(0, nb, n)


def make_order(kind, rng, epoch, ub, ib, first_uniform):
    n = tr.nnz
    if kind == "uniform" or (first_uniform and epoch == 0):
        return rng.permutation(n).astype(np.uint64)
    tile_u, tile_i = ub[ru], ib[ci]
    out = []
    SUBR = NBU // NB
    PASSES = int(os.environ.get("PASSES", "1"))          # every tile in PASSES random parts, the rounds walked PASSES times
    part = rng.integers(0, PASSES, n) if PASSES > 1 else np.zeros(n, np.int64)
    for r_ in range(NBU * PASSES):
        ps, r = r_ // NBU, r_ % NBU
        if os.environ.get("ROUND_PERM") == "1":
            if r == 0: rperm = rng.permutation(NBU)
            r = int(rperm[r])
        for x in range(NB):
            idx = np.nonzero((tile_u == x * SUBR + r % SUBR) & (tile_i == (x + r // SUBR) % NB) & (part == ps))[0]
            if kind == "tile_shuffled":            # the tile's ratings in a uniformly random order
                out.append(rng.permutation(idx))
            elif kind.startswith("tile_slots"):    # item-major slots of <= 1024 ratings / <= 64 items, W slots interleaved
                W = int(kind.split(":")[1]) if ":" in kind else 1
                items = ci[idx]
                o = np.argsort(items, kind="stable")
                idx, items = idx[o], items[o]
                # greedy cut into slots
                bounds = [0]; cnt = 0; nitems = 0; last = -1
                starts = np.nonzero(np.diff(items, prepend=-1))[0]
                ends = np.append(starts[1:], len(items))
                for s, e2 in zip(starts, ends):
                    ln = e2 - s
                    if cnt and (cnt + ln > 1024 or nitems + 1 > 64):
                        bounds.append(s); cnt = 0; nitems = 0
                    cnt += ln; nitems += 1
                bounds.append(len(items))
                slots = [rng.permutation(idx[bounds[k]:bounds[k + 1]]) for k in range(len(bounds) - 1)]
                order_slots = rng.permutation(len(slots))
                for g in range(0, len(slots), W):   # W slots at a time, their ratings interleaved at random
                    grp = np.concatenate([slots[k] for k in order_slots[g:g + W]])
                    if W > 1: grp = rng.permutation(grp)
                    out.append(grp)
    return np.concatenate(out).astype(np.uint64)


for kind in os.environ.get("ORDERS", "uniform,tile_slots:1,tile_slots:16,tile_shuffled").split(","):
    for seed in [int(x) for x in os.environ.get("SEEDS", "1").split(",")]:
        rng = np.random.default_rng(seed)
        U, V = orc.init_factors(1, nU, nI, K)
        ub, ib = blocks_of(nU, rng, NBU), blocks_of(nI, rng, NB)
        best = (1e9, None, -1); t0 = time.time()
        for ep in range(EPOCHS):
            if os.environ.get("REBLOCK") == "1": ub, ib = blocks_of(nU, rng, NBU), blocks_of(nI, rng, NB)
            NT = int(os.environ.get("NTILINGS", "1"))     # NT tilings drawn once, used in turn
            if NT > 1:
                if ep == 0: tilings = [(blocks_of(nU, rng, NBU), blocks_of(nI, rng, NB)) for _ in range(NT)]
                ub, ib = tilings[ep % NT]
            if os.environ.get("RELABEL") == "1":          # the same groups, paired in a freshly permuted Latin square
                ub0, ib0 = (ub, ib) if NT > 1 or ep == 0 or os.environ.get("REBLOCK") == "1" else (ub0, ib0)
                ub, ib = rng.permutation(NBU)[ub0], rng.permutation(NB)[ib0]
            order = make_order(kind, rng, ep, ub, ib, os.environ.get("FIRST_UNIFORM", "1") == "1")
            orc.sgd_pass(U, V, ru, ci, rv, order, lr, ureg, ireg, orc.ARITH_F32, orc.DOT_SEQ)
            v, _, _ = orc.rmse(U, V, nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, invU, invI, orc.DOT_SEQ)
            if not np.isfinite(v): print(kind, "NaN at epoch", ep); break
            if v < best[0]:
                t, _, _ = orc.rmse(U, V, nU, nI, te.nrows, te.rowptr, te.rowind, te.rowval, invU, invI, orc.DOT_SEQ)
                best = (v, t, ep)
            if ep - best[2] > 12: break
        print("%s NBU=%d NB=%d passes=%s roundperm=%s reblock=%s ntilings=%s relabel=%s %-16s seed %d: best val %.5f at epoch %d -> test RMSE %.5f  (%.0f s)" % (which, NBU, NB, os.environ.get("PASSES", "1"), os.environ.get("ROUND_PERM", "0"), os.environ.get("REBLOCK", "0"), os.environ.get("NTILINGS", "1"), os.environ.get("RELABEL", "0"), kind, seed, best[0], best[2], best[1], time.time() - t0), flush=True)
