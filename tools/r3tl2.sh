#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3tl2; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --no-sweep --no-pipeline --steps 200 --warmup 20 --windows 2 > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python tools/trace_timeline.py $O/trace > $O/timeline.txt; cat $O/timeline.txt
rm -rf $O/trace
python bench.py --no-cpu-baseline --no-sweep --no-pipeline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-pipeline bench', round(d['value']/1e6,2), round(d['ms_per_step']*1e3,1), d['kernels_us_per_step'])"
