#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3u; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -25 $O/tests.log
python tools/prof_step.py --batch 4096,16384,65536 2>&1 | grep -v amdgpu.ids | grep -E "kernel sum|l1_|tail_fwd|dw_all"
for i in 1 2; do python bench.py --no-cpu-baseline --no-sweep --steps 200 --warmup 20 2> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value']/1e6,2), 'M obs/s', round(d['ms_per_step']*1e3,2), 'us', d['kernels_us_per_step'])"; done
