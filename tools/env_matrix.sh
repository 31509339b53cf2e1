#!/bin/bash
# pytest -m gpu under each diagnostic switch of the library (the alternative code paths must pass the same parity tests)
set -uo pipefail
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/env_matrix.txt}
: > $O
for v in STDADK_NO_L1_TAIL=1 STDADK_NO_DW_ALL=1 STDADK_NO_TAIL_FWD_BWD=1 STDADK_TAIL_ROWS=16 STDADK_TAIL_ROWS=32 STDADK_TAIL_ROWS=64 \
         STDADK_GEMM_XCD=0 STDADK_KNOTS_PER_WAVE=1 STDADK_KNOT_XCD=0 STDADK_L1_GROUP=1 STDADK_NO_DENSE0_TAIL=1 STDADK_KROT=1; do
  echo "== $v" >> $O
  env $v python -m pytest tests -m gpu -q 2>&1 | tail -2 >> $O
done
echo "== STDADK_NO_FUSED_TAIL=1 (the bf16 tests fail loudly by design: STDADK_FLAG_BF16 needs the fused tail kernels)" >> $O
STDADK_NO_FUSED_TAIL=1 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bf16.py -k "not bf16" 2>&1 | tail -2 >> $O
cat $O
