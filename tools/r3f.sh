#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3f; mkdir -p $O
rm -f st-dadk_amd/csrc/obj/*.o
STDADK_EXTRA_FLAGS=-DSTDADK_DIAG bash st-dadk_amd/csrc/build.sh > $O/build_diag.log 2>&1 || { tail $O/build_diag.log; exit 1; }
python tools/diag/wave_stamps.py 65536 2>&1 | grep -v amdgpu.ids | tee $O/wave_stamps.txt
