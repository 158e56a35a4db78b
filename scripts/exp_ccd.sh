#!/bin/bash
# What bounds the CCD++ pass kernels at C4?  Builds diagnostic variants (MFX_CCD_EXP, see mfx_internal.h) ON the GPU box and
# times the passes with each.  Results of the variants are wrong on purpose.
#   bash scripts/exp_ccd.sh "0 1 2 4 8 16"
set -e
ROOT=$(pwd)
CS=$ROOT/matfac_amd/csrc
for v in ${1:-0 1 2 4 8 16}; do
  # bit 6 (synthetic positions) is only safe in the row view, whose arrays span all positions: the column view keeps variant 0 then
  for f in ccd ccd_cols; do
    w=$v; if [ $f = ccd_cols ] && [ $((v & 64)) -ne 0 ]; then w=0; fi
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I$ROOT/include -I$CS -w -DMFX_CCD_EXP=$w $CCD_DEFS -c $CS/$f.hip -o /tmp/${f}_$v.o
  done
  OBJS=$(ls $CS/*.o | grep -v "/ccd.o\|/ccd_cols.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmfx_c$v.so $OBJS /tmp/ccd_$v.o /tmp/ccd_cols_$v.o -ldl
  echo -n "MFX_CCD_EXP=$v  "
  MFX_LIBRARY=/tmp/libmfx_c$v.so WHAT=ccd CCD_NK=4 python3 scripts/bench_als_ccd.py 2>/dev/null | grep -a "CCD++ C4" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('row pass %.4f ms  col pass %.4f ms  resid %.4f ms' % (d['row_pass_ms'], d['col_pass_ms'], d['resid_ms']))"
done
