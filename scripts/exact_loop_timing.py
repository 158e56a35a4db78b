"""End-to-end time per iteration of the EXACT trainers through the host classes (ModelMF::train, trainSGDPar, trainUShuffle: the
reference's visiting order replayed bit for bit, Model::isTerminateModel every iteration) on the C2 shape: the host builds the
epoch's order (std::shuffle of the index list / the stratified rounds), the GPU replays it.  MFX_STD_SHUFFLE=1: the library
shuffle instead of the block-ahead one."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matfac_amd import synth
from tests.test_host_gpu import host_train
shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
d = synth.make(shape, seed=1)
out = {}
for method in os.environ.get("METHODS", "sgd sgdpar sgdu").split():
    r = {}
    for iters in (3, 9):
        h = host_train(method, d, 64, iters, 1, 0.0025, 0.01, 0.01)
        r[iters] = h["loop_s"]
    out[method] = dict(steady_ms_per_iter=(r[9] - r[3]) / 6 * 1e3, first3_s=r[3], val_rmse=h["val"])
    print(method, out[method], flush=True)
print(json.dumps(out))
