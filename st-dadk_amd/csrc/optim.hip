// A9: clip_grad_norm_ + AdamW + EMA over flat fp32 buffers (HBM-bound: 5 reads + 4 writes of 4 B
// per parameter).  Replaces scripts/train_st_interp.py:696-712 (clip_grad_norm_, optimizer.step,
// ema.update), torch.optim.AdamW's update rule and stnf/utils/ema.py:52-66.
#include "common.h"
#include "bin_body.h"
#include "optim.h"

#include <stdlib.h>

namespace stdadk {

// parts[blockIdx.x] = sum of squares of this block's slice (every one of the SUMSQ_PARTS entries is
// written, so the buffer needs no zeroing and the sum order is fixed).
constexpr int SUMSQ_PARTS = STDADK_SUMSQ_PARTS;
__device__ __forceinline__ void sumsq_block(const float *__restrict__ g, int64_t n, float *__restrict__ parts,
                                            int *__restrict__ step_inc, const int block) {
  if (step_inc && block == 0 && threadIdx.x == 0) step_inc[0] += 1;   // nobody else touches it here
  float acc = 0.f;
  const int64_t stride = (int64_t)SUMSQ_PARTS * blockDim.x;
  int64_t i = (int64_t)block * blockDim.x + threadIdx.x;
  const int64_t n4 = n / 4;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    for (int64_t j = i; j < n4; j += stride) {
      float4 v = g4[j];
      acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc);
      acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
    }
    for (int64_t j = n4 * 4 + i; j < n; j += stride) acc = fmaf(g[j], g[j], acc);
  } else {
    for (int64_t j = i; j < n; j += stride) acc = fmaf(g[j], g[j], acc);
  }
  __shared__ float red[4];
  float s = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) parts[block] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float *__restrict__ g, int64_t n,
                                                    float *__restrict__ parts, int *__restrict__ step_inc) {
  sumsq_block(g, n, parts, step_inc, (int)blockIdx.x);
}

// the clip norms of two parameter groups (MLP / learnable knots) in one launch: blocks [0, PARTS) take
// group 0 (and advance the step counter), blocks [PARTS, 2 PARTS) group 1
__global__ __launch_bounds__(256) void sumsq2_kernel(const float *__restrict__ g0, int64_t n0, float *__restrict__ parts0,
                                                     const float *__restrict__ g1, int64_t n1, float *__restrict__ parts1,
                                                     int *__restrict__ step_inc) {
  if ((int)blockIdx.x < SUMSQ_PARTS) sumsq_block(g0, n0, parts0, step_inc, (int)blockIdx.x);
  else sumsq_block(g1, n1, parts1, nullptr, (int)blockIdx.x - SUMSQ_PARTS);
}

// bf16 operand copies of some matrices of the flat buffer (stdadk_bf16_shadow), by value in the kernel arguments
struct ShadowArgs {
  int n;
  int64_t lo, hi;                       // union of the regions' element ranges: one compare rejects the rest
  stdadk_bf16_region r[STDADK_MAX_HIDDEN];
};

struct AdamArgs {
  float *p; const float *g; float *m; float *v; float *ema;
  int64_t n;
  float lr; const float *lr_dev;
  float beta1, beta2, eps, wd;
  int step; const int *step_dev;
  float max_norm; const float *sumsq; int n_parts; float grad_mul; float ema_decay;
  const float *watch; int *bad_step;      // non-finite guard (stdadk.h): NULL = off
  ShadowArgs sh;
};

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)a, (__bf16)b};          // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return __builtin_bit_cast(uint32_t, v);
}

// the four values p of flat elements e .. e+3 into whichever region holds them (regions start and end on
// multiples of 4 elements and have cols % 4 == 0, so the four share a region and a row)
__device__ __forceinline__ void shadow_store(const ShadowArgs &sh, int64_t e, float4 p) {
#pragma unroll 1
  for (int i = 0; i < sh.n; ++i) {
    const stdadk_bf16_region &r = sh.r[i];
    const int64_t idx = e - r.off;
    if (idx < 0 || idx >= (int64_t)r.rows * r.cols) continue;
    const uint32_t lo = pack_bf16(p.x, p.y), hi = pack_bf16(p.z, p.w);
    if (r.dst) *reinterpret_cast<uint2 *>(r.dst + idx) = make_uint2(lo, hi);
    if (r.dst_t) {
      const int row = (int)(idx / r.cols), col = (int)(idx - (int64_t)row * r.cols);
      uint16_t *t = r.dst_t + (size_t)col * r.rows + row;
      t[0] = (uint16_t)(lo & 0xffffu); t[r.rows] = (uint16_t)(lo >> 16);
      t[2 * (size_t)r.rows] = (uint16_t)(hi & 0xffffu); t[3 * (size_t)r.rows] = (uint16_t)(hi >> 16);
    }
    return;
  }
}

__global__ __launch_bounds__(256) void shadow_refresh_kernel(const float *__restrict__ p, ShadowArgs sh) {
  const int64_t n4 = (sh.hi - sh.lo) / 4;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n4; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = sh.lo + 4 * j;
    shadow_store(sh, e, *reinterpret_cast<const float4 *>(p + e));
  }
}

static int fill_shadow(ShadowArgs &o, const stdadk_bf16_shadow *sh, const float *p, int64_t n) {
  o.n = 0; o.lo = 0; o.hi = 0;
  if (!sh || sh->n == 0) return 0;
  STDADK_REQUIRE(sh->n > 0 && sh->n <= STDADK_MAX_HIDDEN, STDADK_E_ARG, "bf16 shadow: %d regions (1..%d)", sh->n,
                 STDADK_MAX_HIDDEN);
  STDADK_REQUIRE(aligned16(p), STDADK_E_ALIGN, "bf16 shadow: the parameter buffer must be 16-byte aligned");
  for (int i = 0; i < sh->n; ++i) {
    const stdadk_bf16_region &r = sh->r[i];
    const int64_t cnt = (int64_t)r.rows * r.cols;
    STDADK_REQUIRE(r.rows > 0 && r.cols > 0 && (r.cols & 3) == 0 && (r.off & 3) == 0 && r.off >= 0 &&
                       (n < 0 || r.off + cnt <= n),
                   STDADK_E_SHAPE, "bf16 shadow: region %d (off %lld, %d x %d) outside the buffer or not 4-aligned", i,
                   (long long)r.off, r.rows, r.cols);
    STDADK_REQUIRE((!r.dst || (reinterpret_cast<uintptr_t>(r.dst) & 7) == 0) && (r.dst || r.dst_t), STDADK_E_ALIGN,
                   "bf16 shadow: region %d copies NULL or not 8-byte aligned", i);
    o.r[i] = r;
    if (i == 0 || r.off < o.lo) o.lo = r.off;
    if (i == 0 || r.off + cnt > o.hi) o.hi = r.off + cnt;
  }
  o.n = sh->n;
  return 0;
}

// (the EMA value travels by reference + flag: a pointer to a local would put it in scratch memory)
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float &ema, bool has_ema, float gm,
                                         float decay_mul, float b1, float b2, float step_size,
                                         float inv_sqrt_bc2, float eps, float ema_decay) {
  g *= gm;
  p *= decay_mul;
  m = fmaf(b1, m, (1.f - b1) * g);
  v = fmaf(b2, v, (1.f - b2) * g * g);
  float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p -= step_size * (m / denom);
  if (has_ema) ema = fmaf(ema_decay, ema, (1.f - ema_decay) * p);
}

// `block` of `nblocks` groups of 256 threads; `tid` = the thread's index in its group, `red4` = four floats of LDS of
// the group (a group is a whole workgroup in adamw_ema_kernel, a quarter of one in adamw_bin_kernel: the barrier in
// the middle is the workgroup's, which every group of the workgroup reaches)
__device__ __forceinline__ void adamw_ema_block(const AdamArgs &a, const int block, const int nblocks, const int tid,
                                                float *red4) {
  // The first (for up to 4 M parameters: the only) float4 group of every stream is requested BEFORE the
  // prologue below (partials of the clip norm, bias corrections), whose latency then hides behind it.
  typedef float nt4 __attribute__((ext_vector_type(4)));
  const int64_t stride = (int64_t)nblocks * 256;
  const int64_t i0 = (int64_t)block * 256 + tid;
  const bool al = ((reinterpret_cast<uintptr_t>(a.p) | reinterpret_cast<uintptr_t>(a.g) |
                    reinterpret_cast<uintptr_t>(a.m) | reinterpret_cast<uintptr_t>(a.v) |
                    reinterpret_cast<uintptr_t>(a.ema)) & 15) == 0;
  const int64_t n4 = al ? a.n / 4 : 0;
  const int64_t jc = i0 < n4 ? i0 : (n4 > 0 ? n4 - 1 : 0);      // clamped: unconditional loads
  float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), g0 = p0;
  nt4 m0 = {0, 0, 0, 0}, v0 = m0, e0 = m0;
  if (n4 > 0) {
    p0 = reinterpret_cast<float4 *>(a.p)[jc];
    g0 = reinterpret_cast<const float4 *>(a.g)[jc];
    m0 = *(reinterpret_cast<nt4 *>(a.m) + jc);
    v0 = *(reinterpret_cast<nt4 *>(a.v) + jc);
    if (a.ema) e0 = *(reinterpret_cast<nt4 *>(a.ema) + jc);
  }
  const float lr = a.lr_dev ? a.lr_dev[0] : a.lr;
  const int step = a.step_dev ? a.step_dev[0] : a.step;
  // non-finite guard: the objective accumulator is a running sum, so the first step that leaves it non-finite is the
  // first step whose batch objective was (scripts/train_st_interp.py:724-733 stops the epoch there); one thread of
  // the launch, launches of a stream run in order: no atomics
  if (a.bad_step && block == 0 && tid == 0) {
    const float l = a.watch[0];
    if (!(fabsf(l) <= 3.402823466e38f) && a.bad_step[0] == 0) a.bad_step[0] = step;
  }
  float coef = 1.f;
  if (a.max_norm > 0.f && a.sumsq) {
    // sum of the partials by the group's 256 threads, same order in every group (L2-resident, a few hundred floats)
    float ss = 0.f;
    for (int i = tid; i < a.n_parts; i += 256) ss += a.sumsq[i];
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red4[tid >> 6] = ss;
    __syncthreads();
    ss = (red4[0] + red4[1]) + (red4[2] + red4[3]);
    coef = fminf(1.f, a.max_norm / (sqrtf(ss) + 1e-6f));
  }
  const float gm = coef * a.grad_mul;
  const float bc1 = 1.f - powf(a.beta1, (float)step);
  const float bc2 = 1.f - powf(a.beta2, (float)step);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const float decay_mul = 1.f - lr * a.wd;
  int64_t done = 0;
  if (al) {
    // m, v and the EMA shadow are pure streams (touched once per step): non-temporal STORES keep them from
    // evicting the parameters and activations the next step wants in L2 / Infinity Cache (their loads are
    // plain: measured 0.5-1 us faster than non-temporal loads, the 55 MB of optimiser state can sit in the
    // 256 MB Infinity Cache from step to step)
    for (int64_t j = i0; j < n4; j += stride) {
      float4 p = p0, g = g0;
      nt4 mt = m0, vt = v0, et = e0;
      if (j != i0) {
        p = reinterpret_cast<float4 *>(a.p)[j];
        g = reinterpret_cast<const float4 *>(a.g)[j];
        mt = *(reinterpret_cast<nt4 *>(a.m) + j);
        vt = *(reinterpret_cast<nt4 *>(a.v) + j);
        et = a.ema ? *(reinterpret_cast<nt4 *>(a.ema) + j) : (nt4){0, 0, 0, 0};
      }
      float4 m = make_float4(mt.x, mt.y, mt.z, mt.w), v = make_float4(vt.x, vt.y, vt.z, vt.w);
      float4 e = make_float4(et.x, et.y, et.z, et.w);
      const bool he = a.ema != nullptr;
      adam_one(p.x, g.x, m.x, v.x, e.x, he, gm, decay_mul, a.beta1, a.beta2, step_size, inv_sqrt_bc2, a.eps, a.ema_decay);
      adam_one(p.y, g.y, m.y, v.y, e.y, he, gm, decay_mul, a.beta1, a.beta2, step_size, inv_sqrt_bc2, a.eps, a.ema_decay);
      adam_one(p.z, g.z, m.z, v.z, e.z, he, gm, decay_mul, a.beta1, a.beta2, step_size, inv_sqrt_bc2, a.eps, a.ema_decay);
      adam_one(p.w, g.w, m.w, v.w, e.w, he, gm, decay_mul, a.beta1, a.beta2, step_size, inv_sqrt_bc2, a.eps, a.ema_decay);
      reinterpret_cast<float4 *>(a.p)[j] = p;
      __builtin_nontemporal_store((nt4){m.x, m.y, m.z, m.w}, reinterpret_cast<nt4 *>(a.m) + j);
      __builtin_nontemporal_store((nt4){v.x, v.y, v.z, v.w}, reinterpret_cast<nt4 *>(a.v) + j);
      if (a.ema) __builtin_nontemporal_store((nt4){e.x, e.y, e.z, e.w}, reinterpret_cast<nt4 *>(a.ema) + j);
    }
    done = n4 * 4;
    // bf16 operand copies (STDADK_FLAG_BF16): a pass of its own over this thread's float4 groups inside the
    // regions' range, re-reading what the thread itself has just stored -- inside the loop above the table walk
    // cost the whole stream 40 VGPRs (5 instead of 8 waves per SIMD)
    if (a.sh.n > 0) {
      for (int64_t j = i0; j < n4; j += stride)
        if (4 * j >= a.sh.lo && 4 * j < a.sh.hi) shadow_store(a.sh, 4 * j, reinterpret_cast<const float4 *>(a.p)[j]);
    }
  }
  for (int64_t j = done + i0; j < a.n; j += stride) {
    float p = a.p[j], m = a.m[j], v = a.v[j];
    float e = a.ema ? a.ema[j] : 0.f;
    adam_one(p, a.g[j], m, v, e, a.ema != nullptr, gm, decay_mul, a.beta1, a.beta2, step_size, inv_sqrt_bc2, a.eps, a.ema_decay);
    a.p[j] = p; a.m[j] = m; a.v[j] = v;
    if (a.ema) a.ema[j] = e;
  }
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(AdamArgs a) {
  __shared__ float red[4];
  adamw_ema_block(a, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, red);
}

// two parameter groups (own lr, clip norm and clip partials each) in one launch: blocks [0, nb0) update
// group 0, the rest group 1
__global__ __launch_bounds__(256) void adamw_ema2_kernel(AdamArgs a0, AdamArgs a1, int nb0) {
  __shared__ float red[4];
  if ((int)blockIdx.x < nb0) adamw_ema_block(a0, (int)blockIdx.x, nb0, (int)threadIdx.x, red);
  else adamw_ema_block(a1, (int)blockIdx.x - nb0, (int)gridDim.x - nb0, (int)threadIdx.x, red);
}

// The optimiser launch of a step that also bins the NEXT batch (bin_body.h): workgroups [0, n_bin) are the
// independent binning workgroups (1024 threads, 68-100 KiB of dynamic LDS -- which every workgroup of the launch then
// reserves: two workgroups per CU up to 4 096 rows, one beyond, so the optimiser part runs as 1024-thread workgroups
// too, with five float4 loads in flight per thread), the rest update the parameters.  The binning's latency chain (~10 us on eight
// workgroups) hides under the optimiser's HBM stream; the batch preparation needs no launch on the step's critical
// path, no side stream and none of its cross-stream packets (profiles/r03_step_timeline.txt: 6-7 us per step).
__global__ __launch_bounds__(1024) void adamw_bin_kernel(AdamArgs a, BinSmallArgs b, int n_bin) {
  extern __shared__ __attribute__((aligned(16))) int bin_smem[];
  __shared__ float red[4][4];
  if ((int)blockIdx.x < n_bin) { bin_small_body(b, (int)blockIdx.x, n_bin, bin_smem); return; }
  // four groups of 256 threads per workgroup: the access pattern of adamw_ema_kernel with four times the blocks
  const int grp = (int)threadIdx.x >> 8;
  adamw_ema_block(a, 4 * ((int)blockIdx.x - n_bin) + grp, 4 * ((int)gridDim.x - n_bin), (int)threadIdx.x & 255, red[grp]);
}

__global__ void step_advance_kernel(int *s) { s[0] += 1; }

}  // namespace stdadk

using namespace stdadk;

extern "C" int stdadk_step_advance(int32_t *step_dev, stdadk_stream_t stream) {
  STDADK_REQUIRE(step_dev, STDADK_E_ARG, "step_advance: NULL pointer");
  STDADK_LAUNCH(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  STDADK_CHECK_LAUNCH("step_advance");
  return 0;
}

extern "C" int stdadk_sumsq2_f32(const float *g0, int64_t n0, float *parts0, const float *g1, int64_t n1,
                                 float *parts1, int32_t *step_inc, stdadk_stream_t stream) {
  STDADK_REQUIRE(n0 >= 0 && n1 >= 0, STDADK_E_ARG, "sumsq2: negative n");
  STDADK_REQUIRE(parts0 && parts1 && (g0 || n0 == 0) && (g1 || n1 == 0), STDADK_E_ARG, "sumsq2: NULL pointer");
  STDADK_LAUNCH(sumsq2_kernel, dim3(2 * SUMSQ_PARTS), dim3(256), 0, (hipStream_t)stream, g0, n0, parts0, g1, n1,
                parts1, step_inc);
  STDADK_CHECK_LAUNCH("sumsq2");
  return 0;
}

extern "C" int stdadk_sumsq_f32(const float *g, int64_t n, float *parts, int32_t *step_inc,
                                stdadk_stream_t stream) {
  STDADK_REQUIRE(n >= 0, STDADK_E_ARG, "sumsq: negative n");
  STDADK_REQUIRE(parts && (g || n == 0), STDADK_E_ARG, "sumsq: NULL pointer");
  STDADK_LAUNCH(sumsq_kernel, dim3(SUMSQ_PARTS), dim3(256), 0, (hipStream_t)stream, g, n, parts, step_inc);
  STDADK_CHECK_LAUNCH("sumsq");
  return 0;
}

static int adamw_impl(float *p, const float *g, float *m, float *v, float *ema, int64_t n,
                      float lr, const float *lr_dev, float beta1, float beta2, float eps,
                      float weight_decay, int32_t step, const int32_t *step_dev, float max_norm,
                      const float *sumsq_parts, int32_t n_parts, float grad_mul,
                      float ema_decay, const stdadk_bf16_shadow *shadow,
                      const float *loss_watch, int32_t *nonfinite_step, stdadk_stream_t stream,
                      const stdadk::BinSmallArgs *bin) {
  STDADK_REQUIRE(n >= 0, STDADK_E_ARG, "adamw: negative n");
  STDADK_REQUIRE((loss_watch != nullptr) == (nonfinite_step != nullptr), STDADK_E_ARG,
                 "adamw: loss_watch and nonfinite_step go together (both or neither)");
  if (n == 0) return 0;
  STDADK_REQUIRE(p && g && m && v, STDADK_E_ARG, "adamw: NULL pointer");
  STDADK_REQUIRE(step_dev || step >= 1, STDADK_E_ARG, "adamw: step must be >= 1");
  STDADK_REQUIRE(max_norm <= 0.f || (sumsq_parts && n_parts > 0), STDADK_E_ARG, "adamw: max_norm > 0 needs sumsq parts");
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.ema = ema; a.n = n; a.lr = lr; a.lr_dev = lr_dev;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.step = step; a.step_dev = step_dev;
  a.max_norm = max_norm; a.sumsq = sumsq_parts; a.n_parts = n_parts; a.grad_mul = grad_mul; a.ema_decay = ema_decay;
  a.watch = loss_watch; a.bad_step = nonfinite_step;
  if (int rc = fill_shadow(a.sh, shadow, p, n)) return rc;
  STDADK_REQUIRE(a.sh.n == 0 || ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                                  reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(ema)) & 15) == 0,
                 STDADK_E_ALIGN, "adamw: bf16 shadows need 16-byte aligned buffers");
  // grid: at most 6 workgroups per CU = ONE resident round of the chip (72 VGPRs: 7 fit), each thread walking
  // its float4 groups with a grid stride -- a second, partly filled round of one-group threads cost 3 us of the
  // 20 (MI355X, 2.76 M parameters, tools/sweep_knobs.sh: 2 695 blocks 20.6 us, 1 536 blocks 17.6 us = 5.6 TB/s)
  int64_t blocks = ceil_div(n, 256 * 4);
  if (blocks > 1536) blocks = 1536;
  { const char *e = getenv("STDADK_ADAMW_BLOCKS"); if (e && atoi(e) > 0) blocks = atoi(e); }   // measurement aid
  if (bin) {
    // one resident round of 1024-thread workgroups (the dynamic LDS of the binning allows one per CU): the binning
    // workgroups first, the optimiser on the other CUs
    int n_bin = bin->B >= 1024 ? SMALL_WG : 1;
    { const char *e = getenv("STDADK_BIN_WG"); if (e && atoi(e) > 0 && bin->B >= 1024) n_bin = atoi(e); }   // measurement aid
    const size_t lds = (size_t)bin_small_lds_ints(bin->B) * sizeof(int);
    const int per_cu = lds <= 78 * 1024 ? 2 : 1;               // workgroups of this launch a CU holds (160 KiB of LDS)
    int64_t ab = (blocks + 3) / 4;                                // the same groups of 256 threads as the plain launch
    if (ab > 256 * per_cu - n_bin) ab = 256 * per_cu - n_bin;
    static bool attr = false;
    if (!attr) {
      hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void *>(adamw_bin_kernel), BIN_SMALL_LDS_INTS * (int)sizeof(int));
      if (e != hipSuccess) { set_error("adamw_bin: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
      attr = true;
    }
    STDADK_LAUNCH_NAMED("adamw_bin_kernel", adamw_bin_kernel, dim3((unsigned)(ab + n_bin)), dim3(1024), lds,
                        (hipStream_t)stream, a, *bin, n_bin);
    STDADK_CHECK_LAUNCH("adamw_bin");
    return 0;
  }
  STDADK_LAUNCH(adamw_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  STDADK_CHECK_LAUNCH("adamw_ema");
  return 0;
}

extern "C" int stdadk_adamw_ema_f32(float *p, const float *g, float *m, float *v, float *ema, int64_t n,
                                    float lr, const float *lr_dev, float beta1, float beta2, float eps,
                                    float weight_decay, int32_t step, const int32_t *step_dev, float max_norm,
                                    const float *sumsq_parts, int32_t n_parts, float grad_mul,
                                    float ema_decay, const stdadk_bf16_shadow *shadow,
                                    const float *loss_watch, int32_t *nonfinite_step, stdadk_stream_t stream) {
  return adamw_impl(p, g, m, v, ema, n, lr, lr_dev, beta1, beta2, eps, weight_decay, step, step_dev, max_norm,
                    sumsq_parts, n_parts, grad_mul, ema_decay, shadow, loss_watch, nonfinite_step, stream, nullptr);
}

namespace stdadk {
int adamw_ema_with_binning(float *p, const float *g, float *m, float *v, float *ema, int64_t n, float lr,
                           const float *lr_dev, float beta1, float beta2, float eps, float weight_decay,
                           const int32_t *step_dev, float max_norm, const float *sumsq_parts, int32_t n_parts,
                           float ema_decay, const stdadk_bf16_shadow *shadow, const float *loss_watch,
                           int32_t *nonfinite_step, stdadk_stream_t stream, const BinSmallArgs &bin) {
  return adamw_impl(p, g, m, v, ema, n, lr, lr_dev, beta1, beta2, eps, weight_decay, 1, step_dev, max_norm, sumsq_parts,
                    n_parts, 1.0f, ema_decay, shadow, loss_watch, nonfinite_step, stream, &bin);
}
}  // namespace stdadk

static int fill_group(AdamArgs &a, const stdadk_adam_group *gr, float beta1, float beta2, float eps, float wd,
                      int32_t step, const int32_t *step_dev, float grad_mul, float ema_decay) {
  STDADK_REQUIRE(gr && gr->n > 0 && gr->p && gr->g && gr->m && gr->v, STDADK_E_ARG, "adamw2: NULL pointer or empty group");
  STDADK_REQUIRE(gr->max_norm <= 0.f || (gr->sumsq_parts && gr->n_parts > 0), STDADK_E_ARG,
                 "adamw2: max_norm > 0 needs sumsq parts");
  a.p = gr->p; a.g = gr->g; a.m = gr->m; a.v = gr->v; a.ema = gr->ema; a.n = gr->n; a.lr = gr->lr; a.lr_dev = gr->lr_dev;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = wd; a.step = step; a.step_dev = step_dev;
  a.max_norm = gr->max_norm; a.sumsq = gr->sumsq_parts; a.n_parts = gr->n_parts; a.grad_mul = grad_mul;
  a.ema_decay = ema_decay;
  a.watch = nullptr; a.bad_step = nullptr;
  if (int rc = fill_shadow(a.sh, gr->shadow, gr->p, gr->n)) return rc;
  STDADK_REQUIRE(a.sh.n == 0 || ((reinterpret_cast<uintptr_t>(gr->g) | reinterpret_cast<uintptr_t>(gr->m) |
                                  reinterpret_cast<uintptr_t>(gr->v) | reinterpret_cast<uintptr_t>(gr->ema)) & 15) == 0,
                 STDADK_E_ALIGN, "adamw2: bf16 shadows need 16-byte aligned buffers");
  return 0;
}

extern "C" int stdadk_bf16_shadow_refresh(const float *p, const stdadk_bf16_shadow *shadow, stdadk_stream_t stream) {
  STDADK_REQUIRE(p && shadow, STDADK_E_ARG, "bf16_shadow_refresh: NULL pointer");
  ShadowArgs sh;
  if (int rc = fill_shadow(sh, shadow, p, -1)) return rc;
  if (sh.n == 0) return 0;
  int64_t blocks = ceil_div((sh.hi - sh.lo) / 4, 256);
  if (blocks > 1024) blocks = 1024;
  STDADK_LAUNCH(shadow_refresh_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, sh);
  STDADK_CHECK_LAUNCH("bf16_shadow_refresh");
  return 0;
}

extern "C" int stdadk_adamw_ema2_f32(const stdadk_adam_group *g0, const stdadk_adam_group *g1, float beta1,
                                     float beta2, float eps, float weight_decay, int32_t step,
                                     const int32_t *step_dev, float grad_mul, float ema_decay,
                                     const float *loss_watch, int32_t *nonfinite_step, stdadk_stream_t stream) {
  STDADK_REQUIRE(step_dev || step >= 1, STDADK_E_ARG, "adamw2: step must be >= 1");
  STDADK_REQUIRE((loss_watch != nullptr) == (nonfinite_step != nullptr), STDADK_E_ARG,
                 "adamw2: loss_watch and nonfinite_step go together (both or neither)");
  AdamArgs a0, a1;
  int rc = fill_group(a0, g0, beta1, beta2, eps, weight_decay, step, step_dev, grad_mul, ema_decay);
  if (rc) return rc;
  rc = fill_group(a1, g1, beta1, beta2, eps, weight_decay, step, step_dev, grad_mul, ema_decay);
  if (rc) return rc;
  a0.watch = loss_watch; a0.bad_step = nonfinite_step;        // group 0's first block keeps the guard
  int64_t nb0 = ceil_div(a0.n, 256 * 4), nb1 = ceil_div(a1.n, 256 * 4);
  if (nb0 > 1536) nb0 = 1536;        // one resident round of the chip (see stdadk_adamw_ema_f32)
  if (nb1 > 1536) nb1 = 1536;
  STDADK_LAUNCH(adamw_ema2_kernel, dim3((unsigned)(nb0 + nb1)), dim3(256), 0, (hipStream_t)stream, a0, a1, (int)nb0);
  STDADK_CHECK_LAUNCH("adamw_ema2");
  return 0;
}
