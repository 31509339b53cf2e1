#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3c; mkdir -p $O
python -m pytest tests -m gpu -q -k "full_batch_sizes" -s 2>&1 | grep -E "B=|passed|failed" 
for us in 0 3 6 10 20 40; do echo "== TAIL_WAVES=8 stagger ${us} us"; STDADK_TAIL_WAVES=8 STDADK_TAIL_STAGGER_US=$us python tools/prof_step.py --batch 16384,65536 2>&1 | grep -E "kernel sum|tail_fwd_bwd"; done > $O/stagger.log 2>&1
echo "== 16 waves"; python tools/prof_step.py --batch 16384,65536 2>&1 | grep -E "kernel sum|tail_fwd_bwd" >> $O/stagger.log
cat $O/stagger.log
STDADK_TAIL_WAVES=8 STDADK_TAIL_STAGGER_US=10 python -m pytest tests -m gpu -q > $O/tests_waves8.log 2>&1; tail -5 $O/tests_waves8.log
