#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3l; mkdir -p $O
python tools/bf16_error_table.py 2>&1 | grep -v amdgpu.ids | tail -25
cp tests/golden/bf16_achieved.json $O/
python -m pytest tests/test_gpu_bf16.py -m gpu -q 2>&1 | tail -3
python tools/prof_step.py --batch 4096,65536 --dtype bf16 2>&1 | grep -E "kernel sum|l1_tail|tail_fwd_bwd" | tee $O/prof_bf16.log
