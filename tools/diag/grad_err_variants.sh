#!/bin/bash
# where does the 1e-5-level gradient error of the C2 model at B = 20000 come from?  (tools/diag/grad_err.py per variant)
cd "$(dirname "$0")/../.."
for v in "" "STDADK_TAIL_ROWS=16" "STDADK_TAIL_ROWS=64" "STDADK_NO_FUSED_TAIL=1" "STDADK_L1_GROUP=1" "STDADK_KNOTS_PER_WAVE=1" "STDADK_NO_DW_ALL=1"; do
  echo "== variant: ${v:-default}"
  env $v python tools/diag/grad_err.py "$@" 2>&1 | grep -v amdgpu.ids
done
