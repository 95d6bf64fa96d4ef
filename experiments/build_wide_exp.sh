#!/bin/bash
# Quick experimental build of the wide linear kernel (192-row instantiations only, bf16 weights,
# runtime switch between workgroup shapes): experiments/build_wide_exp.sh [extra -D flags]
set -e
cd "$(dirname "$0")/.."
mkdir -p build/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function \
  -DSD_WIDE_SWITCH -DSD_WIDE_EXP "$@" -c vllm_omni_draft_amd/csrc/sd_linear_wide.hip -o build/exp/sd_linear_wide_exp.o
cd vllm_omni_draft_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/exp/libsd_amd_exp.so sd_runtime.o sd_sampling.o \
  sd_linear.o ../../build/exp/sd_linear_wide_exp.o sd_attention.o sd_ipc.o sd_chain.o sd_draft_layer.o \
  -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
