#!/bin/bash
# Builds the -DMFX_EXP=8 variant of the rank-64 tiled SGD instantiation on the GPU box and prints the workgroups' finish times
# (scripts/slot_times.py).  Diagnostic: the variant's visit counts are not counts.
set -e
ROOT=$(pwd)
CS=$ROOT/matfac_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I$ROOT/include -I$CS -w -DMFX_EXP=8 -c $CS/sgd_slots_inst_16x1.hip -o /tmp/inst_8.o
OBJS=$(ls $CS/*.o | grep -v sgd_slots_inst_16x1.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmfx_8.so $OBJS /tmp/inst_8.o -ldl
MFX_LIBRARY=/tmp/libmfx_8.so python3 scripts/slot_times.py
