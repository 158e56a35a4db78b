#!/bin/bash
# Where the tagged dataflow kernel spends its iterations (diagnostic build -DMFX_FLOW_STATS, built ON the GPU box).
#   bash scripts/flow_stats.sh ["extra -D flags"]
set -e
ROOT=$(pwd)
CS=$ROOT/matfac_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I$ROOT/include -I$CS -w -DMFX_FLOW_STATS $1 -c $CS/sgd_flow.hip -o /tmp/flow_stats.o
OBJS=$(ls $CS/*.o | grep -v sgd_flow.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmfx_stats.so $OBJS /tmp/flow_stats.o -ldl
MFX_LIBRARY=/tmp/libmfx_stats.so FLOW_STATS=${FLOW_STATS:-1} python3 scripts/flow_tune.py
