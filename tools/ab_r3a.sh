#!/bin/bash
# round 3, first A/B on the MI355X box: rotated K-chunk start (STDADK_KROT) and the DPP wave sum (-DSTDADK_SHFL_SUM = old)
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3a; mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -8 $O/tests.log
for K in 1 0; do STDADK_KROT=$K python tools/diag/grad_err.py 2>&1 | grep -v amdgpu.ids; done > $O/grad_err_dpp.log 2>&1
for K in 1 0 1 0; do echo "== KROT=$K (dpp sum)"; STDADK_KROT=$K python tools/prof_step.py --batch 4096,65536 2>&1 | grep -v amdgpu.ids; done > $O/prof_dpp.log 2>&1
rm -f st-dadk_amd/csrc/obj/*.o
STDADK_EXTRA_FLAGS=-DSTDADK_SHFL_SUM bash st-dadk_amd/csrc/build.sh > $O/build_shfl.log 2>&1 || exit 1
for K in 1 0; do STDADK_KROT=$K python tools/diag/grad_err.py 2>&1 | grep -v amdgpu.ids; done > $O/grad_err_shfl.log 2>&1
for K in 1 0; do echo "== KROT=$K (shfl sum)"; STDADK_KROT=$K python tools/prof_step.py --batch 4096,65536 2>&1 | grep -v amdgpu.ids; done > $O/prof_shfl.log 2>&1
rm -f st-dadk_amd/csrc/obj/*.o
STDADK_EXTRA_FLAGS=-DSTDADK_DIAG bash st-dadk_amd/csrc/build.sh > $O/build_diag.log 2>&1 || exit 1
for K in 1 0; do echo "== KROT=$K"; STDADK_KROT=$K python tools/stamp_tail.py 4096 2>&1 | grep -v amdgpu.ids; done > $O/stamps_4096.log 2>&1
cat $O/grad_err_dpp.log $O/grad_err_shfl.log
cat $O/prof_dpp.log $O/prof_shfl.log | grep -E "==|kernel sum|l1_tail|tail_fwd_bwd|dw_all"
