#!/bin/bash
# Builds st-dadk_amd/lib/libstdadk.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
mkdir -p ../lib obj
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function ${STDADK_EXTRA_FLAGS:-}"
pids=()
for f in rbf_build gemm_f32 mlp optim window tail loss knots fused_step dw_all sparsity; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ common.h -nt obj/$f.o ] || [ gemm_f32.h -nt obj/$f.o ] || [ window.h -nt obj/$f.o ] || [ tail.h -nt obj/$f.o ] || [ loss.h -nt obj/$f.o ] || [ knots.h -nt obj/$f.o ] || [ l1_body.h -nt obj/$f.o ] || [ bin_body.h -nt obj/$f.o ] || [ tail_body.h -nt obj/$f.o ] || [ gemm_body.h -nt obj/$f.o ] || [ l1_bwd_body.h -nt obj/$f.o ] || [ basis.h -nt obj/$f.o ] || [ ../../include/stdadk.h -nt obj/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
if [ ! -f obj/api.o ] || [ api.cpp -nt obj/api.o ] || [ ../../include/stdadk.h -nt obj/api.o ]; then
  $HIPCC $FLAGS -x hip -c api.cpp -o obj/api.o &
  pids+=($!)
fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC -shared -fPIC --offload-arch=gfx950 obj/*.o -o ../lib/libstdadk.so
echo "built $(cd ../lib && pwd)/libstdadk.so"
