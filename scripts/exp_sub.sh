#!/bin/bash
# MFX_SUB (user blocks per XCD) experiment: build libmfx with -DMFX_SUB=$1 into /tmp on the GPU box, time the bench epoch
# and take the FETCH_SIZE / WRITE_SIZE passes.   bash scripts/exp_sub.sh 2
set -e
SUB=${1:-2}
ROOT=$(pwd)
CS=$ROOT/matfac_amd/csrc
B=/tmp/sub_$SUB
mkdir -p $B
FL="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I$ROOT/include -I$CS -w -DMFX_SUB=$SUB"
for f in sgd_slots.hip setup.hip $(cd $CS; ls sgd_slots_inst_*.hip); do
  /opt/rocm/bin/hipcc $FL -c $CS/$f -o $B/${f%.hip}.o &
done
wait
OBJS=$(ls $CS/*.o | grep -v -e sgd_slots -e setup.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libmfx.so $OBJS $B/*.o -ldl
export MFX_LIBRARY=$B/libmfx.so
python3 bench.py --steps 50 --no-cpu-baseline --no-secondary --no-parity 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('SUB=$SUB ms/step %.4f  round launch %.4f ms x %d  val rmse %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches_per_step'], d['val_rmse_after']))"
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $B/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d $B/pmc_$c -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-parity > $B/pmc_$c.log 2>&1
  python3 - $B/pmc_$c $c <<'PY'
import csv, glob, sys, os
v=[]
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "sgd_slots_kernel" in row["Kernel_Name"] and ", false, false, 0>" in row["Kernel_Name"] and row["Counter_Name"] == sys.argv[2]:
            v.append(float(row["Counter_Value"]))
print(sys.argv[2], "per launch (KB): %.0f over %d launches; per epoch (MB): %.0f" % (sum(v)/len(v), len(v), sum(v)/5/1024))
PY
done
