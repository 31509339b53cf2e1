#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3j; mkdir -p $O
echo "== A double buffer"; python tools/prof_step.py --batch 4096,65536 2>&1 | grep -E "l1_tail|tail_fwd_bwd" | tee $O/prof_apipe.log
rm -f st-dadk_amd/csrc/obj/*.o
STDADK_EXTRA_FLAGS="-DSTDADK_SCHED_GROUPS" bash st-dadk_amd/csrc/build.sh > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
echo "== A double buffer + sched groups"; python tools/prof_step.py --batch 4096,65536 2>&1 | grep -E "l1_tail|tail_fwd_bwd" | tee $O/prof_sched.log
