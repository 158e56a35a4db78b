L=gpurun_out/r4_flow1.log; : > $L
echo "== prio on" >> $L;  CONFIGS="tag:4 tag:2" python scripts/flow_tune.py >> $L 2>&1
echo "== prio off" >> $L; MFX_FLOW_PRIO=0 CONFIGS="tag:4 tag:2" python scripts/flow_tune.py >> $L 2>&1
echo "== K=128 prio on" >> $L; RANK=128 CONFIGS="tag:2" python scripts/flow_tune.py >> $L 2>&1
echo "== K=128 prio off" >> $L; MFX_FLOW_PRIO=0 RANK=128 CONFIGS="tag:2" python scripts/flow_tune.py >> $L 2>&1
cat $L
