#!/bin/bash
# Per-kernel HBM counters and durations of one eager draft pass and one eager verify pass
# (experiments/pmc_passes.py): a kernel trace, then one counter per rocprofv3 run.  Run from
# the repository root on the GPU box; profiles/collect_passes.py turns the output into
# profiles/<round>_pmc_hbm.json.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${ROUND:-r3}_profile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/passes_trace -o k -- \
  python3 $ROOT/experiments/pmc_passes.py > $OUT/passes_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $OUT/passes_$c -o pmc -- \
    python3 $ROOT/experiments/pmc_passes.py > $OUT/passes_$c.log 2>&1
done
find $OUT -name "*.csv" | grep passes
