#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3k; mkdir -p $O
python tools/prof_step.py --batch 4096 2>&1 | grep -E "kernel sum|l1_tail" | tee $O/prof.log
python tools/prof_step.py --batch 4096 2>&1 | grep -E "kernel sum|l1_tail" | tee -a $O/prof.log
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
