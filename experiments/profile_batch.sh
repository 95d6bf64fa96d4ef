#!/bin/bash
# Kernel statistics of the batched bench (Qwen2.5-7B + 0.5B shapes, B = 32): which kernels the
# 15 ms speculative step is made of.  Run from the repository root on the GPU box.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/batch_profile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o batch -- \
  python3 $ROOT/experiments/bench_batch.py 32 256 ${DRAFT_ROWS:-auto} > $OUT/bench.log 2>&1
head -30 $OUT/stats/batch_kernel_stats.csv
