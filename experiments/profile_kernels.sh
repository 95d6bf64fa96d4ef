#!/bin/bash
# Kernel statistics (average duration) of the two roofline kernels alone, same launches as
# the counter passes.  Run from the repository root on the GPU box.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${ROUND:-r2}_profile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_linear -o k -- \
  python3 $ROOT/experiments/pmc_linear.py > $OUT/stats_linear.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_attn -o k -- \
  python3 $ROOT/experiments/pmc_attention.py > $OUT/stats_attn.log 2>&1
grep -h -E "linear_kernel|paged_attn|attn_merge" $OUT/stats_linear/k_kernel_stats.csv $OUT/stats_attn/k_kernel_stats.csv
