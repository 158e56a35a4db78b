set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_ccdblk; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; export WHAT=ccd CCD_NK=8
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/scripts/bench_als_ccd.py > $OUT/trace.log 2>&1; echo "trace rc=$?"
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-200
