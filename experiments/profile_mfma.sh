#!/bin/bash
# MFMA utilisation of the two roofline kernels: matrix-pipe busy cycles against the time the
# GPU was active (one counter pass per script).  Run from the repository root on the GPU box.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${ROUND:-r2}_profile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_linear_mfma -o pmc -- \
  python3 $ROOT/experiments/pmc_linear.py > $OUT/pmc_linear_mfma.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_attn_mfma -o pmc -- \
  python3 $ROOT/experiments/pmc_attention.py > $OUT/pmc_attn_mfma.log 2>&1
ls $OUT/pmc_linear_mfma $OUT/pmc_attn_mfma
