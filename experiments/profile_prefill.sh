#!/bin/bash
# Kernel statistics of a 1024-token prefill (Llama-3-8B + 1B shapes, 256-row passes of the wide
# family): which kernels it is made of.  Run from the repository root on the GPU box.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prefill_profile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o prefill -- \
  python3 $ROOT/experiments/bench_prefill.py "$@" > $OUT/bench.log 2>&1
head -14 $OUT/stats/prefill_kernel_stats.csv
