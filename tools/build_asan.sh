#!/bin/bash
# Host-side sanitizer build of the C-ABI library (CPU build container only):
# the HOST code of every translation unit -- descriptor validation, workspace planner, job tables, launch code -- with
# AddressSanitizer + UndefinedBehaviorSanitizer; the device code is compiled unsanitized (-fno-gpu-sanitize: GPU
# sanitizers and XNACK are not available on this pool).  Output: st-dadk_amd/lib/libstdadk_asan.so (git-ignored).
# It is exercised with STDADK_DRY_RUN=1 (no HIP call) by tests/test_host_sanitizer.py.
set -euo pipefail
cd "$(dirname "$0")/../st-dadk_amd/csrc"
mkdir -p ../lib obj_asan
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
pids=()
for f in rbf_build gemm_f32 mlp optim window tail loss knots fused_step dw_all sparsity; do
  if [ ! -f obj_asan/$f.o ] || [ -n "$(find . ../../include -maxdepth 1 \( -name '*.h' -o -name "$f.hip" \) -newer obj_asan/$f.o)" ]; then
    $HIPCC $FLAGS -c $f.hip -o obj_asan/$f.o &
    pids+=($!)
  fi
done
if [ ! -f obj_asan/api.o ] || [ api.cpp -nt obj_asan/api.o ] || [ ../../include/stdadk.h -nt obj_asan/api.o ]; then
  $HIPCC $FLAGS -x hip -c api.cpp -o obj_asan/api.o &
  pids+=($!)
fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC -shared -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -shared-libsan obj_asan/*.o -o ../lib/libstdadk_asan.so
echo "built $(cd ../lib && pwd)/libstdadk_asan.so"
