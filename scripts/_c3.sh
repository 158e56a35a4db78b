set -o pipefail
timeout -k 10 300 python -m pytest tests/test_ccd_gpu.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2 3; do WHAT=ccd timeout -k 10 400 python scripts/bench_als_ccd.py 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: round(d[k],4) for k in ('ms_per_factor','row_pass_ms','col_pass_ms','resid_ms')})"; done
