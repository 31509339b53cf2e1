#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3fin; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -3 $O/tests.log
bash tools/r3v.sh 2>&1 | tail -8
