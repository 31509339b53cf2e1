#!/bin/bash
# round 3: reductions folded into the merged weight-gradient launch (STDADK_DW_FIN=2) against the reductions launch (=0)
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3s; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "reductions_inside or nonfinite or sharded_optimizer_virtual" > $O/tests_new.log 2>&1; rc=$?; tail -5 $O/tests_new.log
[ $rc -eq 0 ] || exit $rc
for F in 0 1 0 1; do
  if [ $F = 1 ]; then export STDADK_DW_FIN=0; else export STDADK_DW_FIN=2; fi
  echo "== STDADK_DW_FIN=$([ $F = 1 ] && echo 0 || echo 2)"
  python tools/prof_step.py --batch 4096,65536 2>&1 | grep -v amdgpu.ids | grep -E "kernel sum|l1_tail|tail_fwd_bwd|dw_all|reduce_jobs|adamw|step"
  python bench.py --no-cpu-baseline --no-sweep --steps 200 --warmup 20 2> $O/bench_$F.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value']/1e6,2), 'M obs/s', round(d['ms_per_step']*1e3,2), 'us')"
done 2>&1 | tee $O/ab.log
unset STDADK_DW_FIN
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -4 $O/tests.log
