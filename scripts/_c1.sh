set -o pipefail
timeout -k 10 100 python scripts/_dbg_ccd.py 2>&1 | grep -v worst
timeout -k 10 600 python -m pytest tests/test_ccd_gpu.py -x -q -m gpu > gpurun_out/r4_ccdblk_test.log 2>&1; echo "tests rc=$?" 
tail -5 gpurun_out/r4_ccdblk_test.log
WHAT=ccd timeout -k 10 400 python scripts/bench_als_ccd.py > gpurun_out/r4_ccdblk_bench.log 2>&1; echo "bench rc=$?"
tail -3 gpurun_out/r4_ccdblk_bench.log
