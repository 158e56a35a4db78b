#!/bin/bash
# What bounds sgd_slots_kernel<16,1,F32>?  Builds diagnostic variants of the rank-64 instantiation (MFX_EXP, see
# sgd_slots_kernel.h) ON the GPU box and times the bench epoch with each.  Results of the variants are wrong on purpose.
#   bash scripts/exp_bound.sh "0 1 2 3 4 5"
set -e
ROOT=$(pwd)
CS=$ROOT/matfac_amd/csrc
for v in ${1:-0 1 2 3 4 5}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I$ROOT/include -I$CS -w -DMFX_EXP=$v -c $CS/sgd_slots_inst_16x1.hip -o /tmp/inst_$v.o
  OBJS=$(ls $CS/*.o | grep -v sgd_slots_inst_16x1.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmfx_$v.so $OBJS /tmp/inst_$v.o -ldl
  echo -n "MFX_EXP=$v  "
  MFX_LIBRARY=/tmp/libmfx_$v.so python3 bench.py --steps 50 --no-cpu-baseline --no-secondary --no-parity 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step %.4f  round launch %.4f ms  val rmse %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['val_rmse_after']))"
done
