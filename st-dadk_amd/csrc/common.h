// Shared helpers of libstdadk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/stdadk.h"

namespace stdadk {

void set_error(const char *fmt, ...);

#define STDADK_REQUIRE(cond, code, ...)          \
  do {                                            \
    if (!(cond)) {                                \
      ::stdadk::set_error(__VA_ARGS__);           \
      return (code);                              \
    }                                             \
  } while (0)

// Host-side dry run (environment STDADK_DRY_RUN=1, read once when the library is loaded): every entry point runs its
// argument validation, planning and job-table construction, and NO HIP call is made -- the launches and the LDS
// attribute calls are skipped.  It exists for the host-side AddressSanitizer / UBSan pass (tools/build_asan.sh,
// tests/test_host_sanitizer.py), which runs in a container without a GPU; pointers are never dereferenced on the host.
extern bool g_dry_run;

// Launch errors surface as positive hipError_t values.
#define STDADK_CHECK_LAUNCH(what)                                             \
  do {                                                                        \
    if (::stdadk::g_dry_run) break;                                           \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      ::stdadk::set_error("%s: %s", what, hipGetErrorString(e__));            \
      return (int)e__;                                                        \
    }                                                                         \
  } while (0)

// Every kernel launch goes through this macro so that stdadk_profile_enable() can bracket it with
// HIP events on the launch stream (per-kernel device time without an external profiler).
void prof_before(const char *name, hipStream_t st);
void prof_after(hipStream_t st);
extern bool g_prof_on;
#define STDADK_LAUNCH(kern, grid, block, lds, st, ...)                         \
  do {                                                                        \
    if (::stdadk::g_dry_run) break;                                           \
    if (::stdadk::g_prof_on) ::stdadk::prof_before(#kern, (st));              \
    hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);              \
    if (::stdadk::g_prof_on) ::stdadk::prof_after((st));                      \
  } while (0)

#define STDADK_LAUNCH_NAMED(name, kern, grid, block, lds, st, ...)              \
  do {                                                                        \
    if (::stdadk::g_dry_run) break;                                           \
    if (::stdadk::g_prof_on) ::stdadk::prof_before((name), (st));             \
    hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);              \
    if (::stdadk::g_prof_on) ::stdadk::prof_after((st));                      \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel (skipped in a dry run)
static inline hipError_t set_max_dynamic_lds(const void *kernel, int bytes) {
  return g_dry_run ? hipSuccess : hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int kWave = 64;

// Sum over the 64 lanes of a wave, the same value in every lane.  Data-parallel primitives (DPP) inside the rows of
// 16 lanes -- quad swaps, then the half-row and row mirrors: four v_add_f32_dpp, no LDS crossbar -- then the row sums
// chained over the four rows (row_bcast15 / row_bcast31) and lane 63 read back through an SGPR.  The butterfly through
// ds_bpermute (__shfl_xor) this replaces was six dependent LDS round trips per sum: at one row per wave the two sums of
// a LayerNorm phase were ~40 % of the phase.  Fixed summation tree: ((quad pairs) half-rows) rows, (r0+r1)+(r2+r3).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
#ifdef STDADK_SHFL_SUM      // the butterfly through the LDS crossbar (kept for A/B measurements)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#else
  v += dpp_f32<0xB1>(v);            // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_f32<0x4E>(v);            // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_f32<0x141>(v);           // row_half_mirror: the other quad of the 8
  v += dpp_f32<0x140>(v);           // row_mirror: the other half of the 16 -> every lane holds its row's sum
  v += dpp_f32<0x142, 0xA>(v);      // row_bcast15 into rows 1 and 3: r0 + r1, r2 + r3
  v += dpp_f32<0x143, 0xC>(v);      // row_bcast31 into rows 2 and 3: row 3 = (r0 + r1) + (r2 + r3)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
}

// Counter-based keep-mask generator for dropout: one 32-bit hash per (seed, layer, element).
// Stateless, so backward regenerates exactly the mask forward used.
__device__ __forceinline__ uint32_t mix32(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return (uint32_t)x;
}
// Two stages: a 64-bit mix per (seed, layer, ROW) -- once per row, wave-uniform wherever a wave owns a row, so it
// runs on the scalar unit -- and a 32-bit finisher (two multiplies, three xor-shifts) that serves TWO columns: the
// columns c and c + 64 of a 128-column block share one hash, the low 16 bits decide the first, the high 16 bits the
// second.  A lane of the row-local phases owns the columns lane + 64 cc, i.e. exactly such pairs: one finisher per two
// elements (round 3; the finisher per element was ~13 of the ~40 vector instructions a LayerNorm-phase element cost,
// two of them quarter-rate multiplies).  The 64-bit mix per ELEMENT of round 1 was ~40 on its own.
__device__ __forceinline__ uint32_t drop_rowkey(uint64_t seed, int layer, int64_t row) {
  return mix32(seed ^ (0x9E3779B97F4A7C15ULL * (uint64_t)(layer + 1)) ^ ((uint64_t)row * 0xD1B54A32D192ED03ULL));
}
// keep when u >= p for u = v / 2^16 uniform in [0,1), v = the column's 16 bits: v >= ceil(p 2^16)   (P(keep) = 1-p)
__device__ __forceinline__ uint32_t drop_threshold(float p) { return (uint32_t)ceilf(p * 65536.0f); }
__device__ __forceinline__ int drop_pair(int col) { return (col & 63) | ((col >> 7) << 6); }
__device__ __forceinline__ uint32_t drop_hash(uint32_t rowkey, int pair) {
  uint32_t h = rowkey ^ ((uint32_t)pair * 0x9E3779B1u);
  h ^= h >> 16; h *= 0x21f0aaadu;
  h ^= h >> 15; h *= 0x735a2d97u;
  h ^= h >> 15;
  return h;
}
__device__ __forceinline__ bool drop_keep(uint32_t rowkey, int col, uint32_t thr) {
  const uint32_t h = drop_hash(rowkey, drop_pair(col));
  return (((col >> 6) & 1) ? (h >> 16) : (h & 0xffffu)) >= thr;
}

// LayerNorm statistics, the same expressions in every kernel that normalises a row (tail bodies, layer-0 window
// bodies, the per-layer row kernels), so that the paths stay bitwise interchangeable:
//   mean = sum * inv_n, var = sum of squared deviations * inv_n, with inv_n = 1.0f / n taken once per phase (an IEEE
//   division per ROW was ~10 vector instructions; for the power-of-two widths of every shipped configuration the
//   product equals the quotient bit for bit);
//   rstd = 1 / sqrt(var + eps) from v_rsq_f32 (1 ulp) and one Newton step -- the correctly rounded sqrtf followed by
//   an IEEE division was ~25 vector instructions per row.
__device__ __forceinline__ float ln_rstd(float var, float eps) {
  const float x = var + eps;
  const float y = __builtin_amdgcn_rsqf(x);
  return y * fmaf(-0.5f * x * y, y, 1.5f);
}

}  // namespace stdadk
