#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3d; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -4 $O/tests.log
python tools/prof_step.py --batch 4096,16384,65536 2>&1 | grep -v amdgpu.ids > $O/prof_f32.log; python tools/prof_step.py --batch 4096,65536 --dtype bf16 2>&1 | grep -v amdgpu.ids > $O/prof_bf16.log
grep -E "kernel sum|l1_tail|tail_fwd_bwd|dw_all|l1_window" $O/prof_f32.log $O/prof_bf16.log
