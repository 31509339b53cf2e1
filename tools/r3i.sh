#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3i; mkdir -p $O
python tools/prof_step.py --batch 4096,16384,65536 2>&1 | grep -E "kernel sum|l1_tail|tail_fwd_bwd" | tee $O/prof.log
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
rm -f st-dadk_amd/csrc/obj/*.o
STDADK_EXTRA_FLAGS="-DSTDADK_DIAG" bash st-dadk_amd/csrc/build.sh > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
python tools/diag/wave_stamps.py 65536 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/wave_stamps.txt
python tools/stamp_tail.py 65536 2>&1 | grep -v amdgpu.ids > $O/stamps_b65536.txt; cat $O/stamps_b65536.txt
python tools/stamp_tail.py 4096 2>&1 | grep -v amdgpu.ids > $O/stamps_b4096.txt; grep -E "gemm|total|fused" $O/stamps_b4096.txt
