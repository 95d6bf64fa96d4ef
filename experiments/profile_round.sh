#!/bin/bash
# Kernel trace + statistics of the headline bench, then HBM counters of the two
# roofline kernels (one counter per pass).  Run from the repository root on the GPU box.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${ROUND:-r2}_profile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
  python3 $ROOT/bench.py --steps 16 --warmup 4 --no-sweep --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_linear_$c -o pmc -- \
    python3 $ROOT/experiments/pmc_linear.py > $OUT/pmc_linear_$c.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_attn_$c -o pmc -- \
    python3 $ROOT/experiments/pmc_attention.py > $OUT/pmc_attn_$c.log 2>&1
done
find $OUT -name "*.csv" | head -30
