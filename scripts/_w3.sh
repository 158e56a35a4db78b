export WHICH=mid WARM=flow:1 SEEDS=1,2,3
L=gpurun_out/r4_warm3.log; : > $L
echo "== WAVES=16 BLOCKS=32 (8 WGs)" >> $L; MFX_SGD_BLOCKS=32 python scripts/warm_epochs.py >> $L 2>&1
echo "== WAVES=1 default blocks" >> $L; WAVES=1 python scripts/warm_epochs.py >> $L 2>&1
echo "== WAVES=1 BLOCKS=32" >> $L; WAVES=1 MFX_SGD_BLOCKS=32 python scripts/warm_epochs.py >> $L 2>&1
echo "== WAVES=2" >> $L; WAVES=2 python scripts/warm_epochs.py >> $L 2>&1
echo "== WAVES=8" >> $L; WAVES=8 python scripts/warm_epochs.py >> $L 2>&1
echo "== WAVES=4 BLOCKS=128 (32 WGs)" >> $L; WAVES=4 MFX_SGD_BLOCKS=128 python scripts/warm_epochs.py >> $L 2>&1
grep -v reference $L
