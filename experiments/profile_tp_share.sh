set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/tp_profile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o tp -- \
  python3 $ROOT/experiments/bench_tp_share.py 1 1024 > $OUT/bench.log 2>&1
rm -f $OUT/stats/*kernel_trace.csv
cut -c1-170 $OUT/stats/tp_kernel_stats.csv | head -24
