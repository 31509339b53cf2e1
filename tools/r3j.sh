#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3j; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -q -x -k "next_batch_binned or one_call_step or bin_obs or indexed" 2>&1 | tail -3
python tools/host_vs_kernels.py 2>&1 | grep -v amdgpu.ids | head -2 | cut -c1-220
for E in 0 1 0 1; do
  echo "== STNF_NO_INLINE_PREP=$E"
  STNF_NO_INLINE_PREP=$E python bench.py --no-cpu-baseline --no-sweep --steps 200 --warmup 20 2> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value']/1e6,2), 'M obs/s', round(d['ms_per_step']*1e3,2), 'us')"
done
