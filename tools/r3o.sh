#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
python tools/prof_step.py --batch 4096 2>&1 | grep -E "kernel sum|l1_tail"
python tools/prof_step.py --batch 4096 2>&1 | grep -E "kernel sum|l1_tail"
python -m pytest tests -m gpu -q -x 2>&1 | tail -2
