#!/bin/bash
# Same-box A/B of two builds of the library: the baseline (build/base/libsd_amd_base.so, built
# from the last commit) and the working tree's.  Devices differ by a few percent, so variants
# are only compared inside one GPU call.  Usage: experiments/ab.sh <script.py> [args...]
for round in 1 2; do
  for lib in build/base/libsd_amd_base.so ""; do
    if [ -n "$lib" ]; then export SD_AMD_LIB=$lib; tag="base"; else unset SD_AMD_LIB; tag="new "; fi
    timeout -k 10 300 python -u "$@" 2>&1 | grep -v amdgpu.ids | sed "s/^/[$tag] /"
  done
done
