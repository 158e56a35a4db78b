L=gpurun_out/r4_flow2.log; : > $L
timeout -k 10 600 python -m pytest tests/test_sgd_gpu.py -m gpu -x -q -k "pole_blocks or level_schedule_is" >> $L 2>&1
tail -3 $L
echo "== pole on" >> $L;  CONFIGS="tag:4" python scripts/flow_tune.py >> $L 2>&1
echo "== pole off" >> $L; MFX_FLOW_POLE=0 CONFIGS="tag:4" python scripts/flow_tune.py >> $L 2>&1
echo "== K=128 pole on" >> $L; RANK=128 CONFIGS="tag:2" python scripts/flow_tune.py >> $L 2>&1
echo "== K=256 pole on" >> $L; RANK=256 CONFIGS="tag:2" python scripts/flow_tune.py >> $L 2>&1
echo "== f32 pole on" >> $L; ARITH=f32 CONFIGS="tag:4" python scripts/flow_tune.py >> $L 2>&1
grep -E "==|C2 K" $L
