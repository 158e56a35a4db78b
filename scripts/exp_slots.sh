#!/bin/bash
# Slot-shape experiments of the tiled SGD schedule: builds libmfx with the given -D flags (MFX_SLOT_CAP, MFX_SLOT_WORDS,
# MFX_SLOT_ROWS, MFX_SUB ...; sgd_slots.h) into /tmp on the GPU box and times the bench epoch.
#   bash scripts/exp_slots.sh "-DMFX_SLOT_WORDS=8192 -DMFX_SLOT_ROWS=128" "-DMFX_SLOT_CAP=512"
set -e
ROOT=$(pwd)
CS=$ROOT/matfac_amd/csrc
n=0
for D in "$@"; do
  n=$((n+1))
  B=/tmp/slots_$n
  mkdir -p $B
  FL="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I$ROOT/include -I$CS -w $D"
  for f in sgd_slots.hip setup.hip sgd_slots_inst_16x1.hip; do
    /opt/rocm/bin/hipcc $FL -c $CS/$f -o $B/${f%.hip}.o &
  done
  wait
  OBJS=$(ls $CS/*.o | grep -v -e sgd_slots.o -e sgd_slots_inst_16x1.o -e setup.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libmfx.so $OBJS $B/*.o -ldl
  MFX_LIBRARY=$B/libmfx.so python3 bench.py --steps 50 --no-cpu-baseline --no-secondary --no-parity --no-exact 2>/dev/null | D="$D" python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-50s ms/step %.4f  round launch %.4f ms  %.2f G updates/s  val rmse %.4f' % (os.environ['D'] or '(default)', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value'] / 1e9, d['val_rmse_after']))"
done
