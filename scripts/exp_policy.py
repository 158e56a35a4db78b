"""Experiment: convergence + speed of the Hogwild kernel under each cache policy
(MFX_SGD_POLICY) on the C2 shape, against the sequential CPU oracle.  Diagnostic only."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(policy, epochs, scale, K, mode=0, lr=0.0025):
    import numpy as np
    from matfac_amd import Ctx, mfx, synth
    shape = dict(synth.SHAPES["C2"])
    shape["nnz"] = int(shape["nnz"] * scale / 0.8)
    d = synth.make(shape, seed=1)
    tr, va = d["train"], d["val"]
    nU, nI = d["nUsers"], shape["nI"]
    U0, V0 = synth.init_factors(1, nU, nI, K)
    ctx = Ctx(0)
    ctx.set_csr(mfx.MAT_TRAIN, tr.nrows, nI, tr.rowptr, tr.rowind, tr.rowval)
    ctx.set_csr(mfx.MAT_VAL, va.nrows, nI, va.rowptr, va.rowind, va.rowval)
    ctx.set_model(nU, nI, K)
    ctx.set_factors(U0, V0)
    ctx.compute_invalid()
    ctx.prof_enable(True)
    traj = []
    for ep in range(epochs):
        ctx.sgd_epoch(lr, 0.01, 0.01, mode=mode, seed=1, epoch=ep)
        traj.append((ctx.rmse(mfx.MAT_TRAIN), ctx.rmse(mfx.MAT_VAL)))
    ms, n = ctx.prof_get(mfx.K_SGD)
    pms, _ = ctx.prof_get(mfx.K_PERMUTE)
    ems, en = ctx.prof_get(mfx.K_EVAL)
    print(json.dumps(dict(policy=policy, mode=mode, blocks=os.environ.get('MFX_SGD_BLOCKS',''), var=os.environ.get('MFX_TILED_VARIANT',''), nnz=tr.nnz, sgd_ms=ms / n, perm_ms=pms / n, eval_ms=ems / en,
                          gups=tr.nnz * epochs / ms / 1e6, traj=traj)))


def cpu(epochs, scale, K):
    import numpy as np
    from matfac_amd import synth
    from oracle import binding as orc
    shape = dict(synth.SHAPES["C2"])
    shape["nnz"] = int(shape["nnz"] * scale / 0.8)
    d = synth.make(shape, seed=1)
    tr, va = d["train"], d["val"]
    nU, nI = d["nUsers"], shape["nI"]
    U, V = synth.init_factors(1, nU, nI, K)
    invU, invI = orc.invalid(tr.nrows, tr.ncols, tr.rowptr, tr.rowind, nU, nI)
    ru = tr.rowids()
    order = np.arange(tr.nnz, dtype=np.uint64)
    mt = orc.MT(1)
    traj = []
    t0 = time.time()
    for ep in range(epochs):
        mt.shuffle_u64(order)
        orc.sgd_pass(U, V, ru, tr.rowind, tr.rowval, order, float(os.environ.get('LR', '0.0025')), 0.01, 0.01, orc.ARITH_REF64, orc.DOT_SEQ)
        a, _, _ = orc.rmse(U, V, nU, nI, tr.nrows, tr.rowptr, tr.rowind, tr.rowval, invU, invI)
        b, _, _ = orc.rmse(U, V, nU, nI, va.nrows, va.rowptr, va.rowind, va.rowval, invU, invI)
        traj.append((a, b))
    print(json.dumps(dict(policy="cpu-sequential", s_per_epoch=(time.time() - t0) / epochs, traj=traj)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7]))
    elif len(sys.argv) > 1 and sys.argv[1] == "cpu":
        cpu(int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]))
    else:
        epochs, scale, K = int(os.environ.get("EPOCHS", 12)), float(os.environ.get("SCALE", 1.0)), 64
        lr = os.environ.get("LR", "0.0025")
        # CONFIGS: mode:policy:blocks triples
        for cfg in os.environ.get("CONFIGS", "0:0:2048,0:1:2048,3:0:2048").split(","):
            mode, pol, blocks = cfg.split(":")[:3]
            var = cfg.split(":")[3] if cfg.count(":") >= 3 else "0"
            env = dict(os.environ, MFX_SGD_POLICY=pol, MFX_SGD_BLOCKS=blocks, MFX_TILED_VARIANT=var)
            subprocess.run([sys.executable, __file__, "child", pol, str(epochs), str(scale), str(K), mode, lr], env=env, check=False)
            sys.stdout.flush()
        if os.environ.get("CPU", "1") == "1":
            subprocess.run([sys.executable, __file__, "cpu", str(epochs), str(scale), str(K)], check=False)
