#!/bin/bash
# kernel timeline (durations and gaps of the main stream's launches) of the default, pipelined bench loop
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3tl; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --no-sweep --steps 200 --warmup 20 --windows 2 > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python tools/trace_timeline.py $O/trace > $O/timeline.txt; cat $O/timeline.txt
rm -rf $O/trace
