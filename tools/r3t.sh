#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3t; mkdir -p $O
for D in 1 3 1 3 1 3; do
  echo "== GROUP_DEPTH=$D"
  if [ $D = 3 ]; then export STDADK_LIB=$PWD/st-dadk_amd/lib/libstdadk_d3.so; else unset STDADK_LIB; fi
  python tools/prof_step.py --batch 4096,16384,65536 2>&1 | grep -v amdgpu.ids | grep -E "kernel sum|dw_all"
  python tools/prof_step.py --batch 4096,65536 --dtype bf16 2>&1 | grep -v amdgpu.ids | grep -E "kernel sum|dw_all"
done 2>&1 | tee $O/depth.log
