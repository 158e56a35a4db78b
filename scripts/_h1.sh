set -o pipefail
timeout -k 10 600 python -m pytest tests/test_sgd_gpu.py -x -q -m gpu -k "hybrid or pole or dataflow or flow or replay or sequential" > gpurun_out/r4_hybdev_test.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4_hybdev_test.log
timeout -k 10 300 python scripts/c2_replay_time.py > gpurun_out/r4_hybdev_time.log 2>&1; echo "time rc=$?"; tail -6 gpurun_out/r4_hybdev_time.log
