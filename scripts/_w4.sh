export WHICH=mid WARM=flow:1 SEEDS=1,2,3
L=gpurun_out/r4_warm4.log; : > $L
for T in 1 2 4 8; do for RP in 0 1; do
echo "== TILINGS=$T ROUND_PERM=$RP" >> $L; MFX_SGD_TILINGS=$T MFX_SGD_ROUND_PERM=$RP python scripts/warm_epochs.py 2>&1 | grep -v reference >> $L
done; done
echo "== TILINGS=4 RP=1 WAVES=4" >> $L; WAVES=4 MFX_SGD_TILINGS=4 python scripts/warm_epochs.py 2>&1 | grep -v reference >> $L
echo "== TILINGS=4 RP=1 BLOCKS=32 WAVES=1" >> $L; WAVES=1 MFX_SGD_BLOCKS=32 MFX_SGD_TILINGS=4 python scripts/warm_epochs.py 2>&1 | grep -v reference >> $L
echo "== c1 TILINGS=4" >> $L; WHICH=c1 MFX_SGD_TILINGS=4 python scripts/warm_epochs.py 2>&1 | grep -v reference >> $L
cat $L
python bench.py --steps 40 --warmup 8 > gpurun_out/r4_bench_t4.json 2> gpurun_out/r4_bench_t4.err; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r4_bench_t4.json') if l.startswith('{')][-1]); print('bench value', d['value'], d['ms_per_step'])"
