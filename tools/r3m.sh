#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3m; mkdir -p $O
python -m pytest tests/test_gpu_round3.py -m gpu -q -k cooperative 2>&1 | tail -15
for c in 0 1; do echo "== STDADK_L1_COOP=$c"; STDADK_L1_COOP=$c python tools/prof_step.py --batch 16384,65536 2>&1 | grep -E "kernel sum|l1_window|l1_coop"; done | tee $O/prof.log
