"""What bounds the exact replay (MFX_SGD_LEVELS, dataflow schedule) at C2: a critical-path model on the real list.
Every queue is a serial processor of its owned items' visits in list order (step = ns per visit); a user's row that is visited in
another queue than its previous visit pays a hand-off (ns) on top.  Makespan of one epoch for a grid of (step, hand-off):
    python scripts/flow_model.py           (builds the C2 synthetic matrix, ~1 min on one core)
Measured on the GPU: 14.4 ms with the round-3 step (~ 290 ns: the most popular item's 49 777 visits bound it) AND with
the round-4 pole step (~ 120 ns: now the busiest user's 10 717 hand-offs at ~ 1.3 us bound it) -- the two terms happen to coincide."""
import heapq, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from matfac_amd import synth

SRC = r'''
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
int main(int argc, char** argv) {
  long n = atol(argv[2]); int nU = atoi(argv[3]), nI = atoi(argv[4]), nq = atoi(argv[5]);
  char path[512];
  int32_t *u = malloc(4*n), *i = malloc(4*n), *own = malloc(4*(long)nI);
  snprintf(path, 512, "%s/u.bin", argv[1]); FILE* f = fopen(path,"rb"); if (fread(u,4,n,f) != (size_t)n) return 1; fclose(f);
  snprintf(path, 512, "%s/i.bin", argv[1]); f = fopen(path,"rb"); if (fread(i,4,n,f) != (size_t)n) return 1; fclose(f);
  snprintf(path, 512, "%s/own.bin", argv[1]); f = fopen(path,"rb"); if (fread(own,4,nI,f) != (size_t)nI) return 1; fclose(f);
  double *tq = malloc(8*(long)nq), *tu = malloc(8*(long)nU); int32_t* lastq = malloc(4*(long)nU);
  for (int a = 6; a + 1 < argc; a += 2) {
    double c = atof(argv[a]), h = atof(argv[a+1]), mx = 0;
    for (int k=0;k<nq;k++) tq[k]=0; for (int k=0;k<nU;k++) { tu[k]=0; lastq[k]=-1; }
    for (long t=0;t<n;t++) {
      int q = own[i[t]]; double start = tq[q];
      double ready = tu[u[t]] + ((lastq[u[t]]!=q && lastq[u[t]]>=0) ? h : 0.0);
      if (ready > start) start = ready;
      double fin = start + c; tq[q]=fin; tu[u[t]]=fin; lastq[u[t]]=q; if (fin>mx) mx=fin;
    }
    printf("step %4.0f ns, hand-off %5.0f ns: epoch %6.2f ms\n", c, h, mx*1e-6);
  }
  return 0;
}
'''
shape = dict(synth.SHAPES["C2"]); shape["nnz"] = int(shape["nnz"] / 0.8)
tr = synth.make(shape, seed=1)["train"]
order = np.random.default_rng(1).permutation(tr.nnz)
du, di = np.diff(tr.rowptr), np.bincount(tr.rowind, minlength=tr.ncols)
print("C2: %d ratings; longest item chain %d, longest user chain %d" % (tr.nnz, di.max(), du.max()))
nq = int(os.environ.get("QUEUES", "4096"))
own = np.zeros(tr.ncols, np.int32)
h = [(0, q) for q in range(nq)]
heapq.heapify(h)
for it in np.argsort(-di, kind="stable"):          # longest chain first onto the lightest queue (the device builder's dealing)
    if di[it]:
        l, q = heapq.heappop(h); own[it] = q; heapq.heappush(h, (l + int(di[it]), q))
with tempfile.TemporaryDirectory() as d:
    tr.rowids()[order].astype(np.int32).tofile(os.path.join(d, "u.bin"))
    tr.rowind[order].astype(np.int32).tofile(os.path.join(d, "i.bin"))
    own.tofile(os.path.join(d, "own.bin"))
    open(os.path.join(d, "m.c"), "w").write(SRC)
    subprocess.check_call(["gcc", "-O2", "-o", os.path.join(d, "m"), os.path.join(d, "m.c")])
    grid = []
    for c in (290, 120, 60):
        for ho in (1200, 800, 500, 250, 0):
            grid += [str(c), str(ho)]
    subprocess.check_call([os.path.join(d, "m"), d, str(tr.nnz), str(tr.nrows), str(tr.ncols), str(nq)] + grid)
