#!/bin/bash
# Measurement sweeps over the library's environment knobs (MI355X box): AdamW grid size, rbf_build rows per workgroup.
cd "$(dirname "$0")/.."
for b in default 512 1024 1536 1792 2048 2304; do
  if [ "$b" = default ]; then unset STDADK_ADAMW_BLOCKS; else export STDADK_ADAMW_BLOCKS=$b; fi
  echo "adamw blocks=$b: $(python tools/bench_adamw.py 2>/dev/null | grep adamw)"
done
unset STDADK_ADAMW_BLOCKS
python tools/bench_rbf_rows.py 2>/dev/null
