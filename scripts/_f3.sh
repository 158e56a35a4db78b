L=gpurun_out/r4_flow3.log; : > $L
for pf in 0 4 16 256 65535; do echo "== pollfull $pf" >> $L; MFX_FLOW_POLLFULL=$pf CONFIGS="tag:4" python scripts/flow_tune.py >> $L 2>&1; done
echo "== K=128 pollfull 65535" >> $L; MFX_FLOW_POLLFULL=65535 RANK=128 CONFIGS="tag:2" python scripts/flow_tune.py >> $L 2>&1
grep -E "==|C2 K" $L
