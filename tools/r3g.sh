#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3g; mkdir -p $O
python tools/prof_step.py --batch 4096,65536 2>&1 | grep -E "kernel sum|l1_tail|tail_fwd_bwd" | tee $O/prof_bias.log
rm -f st-dadk_amd/csrc/obj/*.o
STDADK_EXTRA_FLAGS="-DSTDADK_GEMM_PRIO" bash st-dadk_amd/csrc/build.sh > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
python tools/prof_step.py --batch 4096,65536 2>&1 | grep -E "kernel sum|l1_tail|tail_fwd_bwd" | tee $O/prof_prio.log
rm -f st-dadk_amd/csrc/obj/*.o
STDADK_EXTRA_FLAGS="-DSTDADK_DIAG -DSTDADK_GEMM_PRIO" bash st-dadk_amd/csrc/build.sh > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
python tools/diag/wave_stamps.py 65536 2>&1 | grep -v amdgpu.ids | tee $O/wave_stamps_prio.txt
