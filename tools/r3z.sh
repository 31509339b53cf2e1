#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3z; mkdir -p $O
python tools/host_vs_kernels.py 2>&1 | grep -v amdgpu.ids | grep -v "^      "
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -2 $O/tests.log
python tools/host_profile.py learn 2>&1 | grep -v amdgpu.ids | head -24
