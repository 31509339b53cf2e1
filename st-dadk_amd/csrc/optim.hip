// A9: clip_grad_norm_ + AdamW + EMA over flat fp32 buffers (HBM-bound: 5 reads + 4 writes of 4 B
// per parameter).  Replaces scripts/train_st_interp.py:696-712 (clip_grad_norm_, optimizer.step,
// ema.update), torch.optim.AdamW's update rule and stnf/utils/ema.py:52-66.
#include "common.h"

namespace stdadk {

// parts[blockIdx.x] = sum of squares of this block's slice (every one of the SUMSQ_PARTS entries is
// written, so the buffer needs no zeroing and the sum order is fixed).
constexpr int SUMSQ_PARTS = STDADK_SUMSQ_PARTS;
__global__ __launch_bounds__(256) void sumsq_kernel(const float *__restrict__ g, int64_t n,
                                                    float *__restrict__ parts, int *__restrict__ step_inc) {
  if (step_inc && blockIdx.x == 0 && threadIdx.x == 0) step_inc[0] += 1;   // nobody else touches it here
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
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
  if (threadIdx.x == 0) parts[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamArgs {
  float *p; const float *g; float *m; float *v; float *ema;
  int64_t n;
  float lr; const float *lr_dev;
  float beta1, beta2, eps, wd;
  int step; const int *step_dev;
  float max_norm; const float *sumsq; int n_parts; float grad_mul; float ema_decay;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float *ema, float gm,
                                         float decay_mul, float b1, float b2, float step_size,
                                         float inv_sqrt_bc2, float eps, float ema_decay) {
  g *= gm;
  p *= decay_mul;
  m = fmaf(b1, m, (1.f - b1) * g);
  v = fmaf(b2, v, (1.f - b2) * g * g);
  float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p -= step_size * (m / denom);
  if (ema) *ema = fmaf(ema_decay, *ema, (1.f - ema_decay) * p);
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(AdamArgs a) {
  // The first (for up to 4 M parameters: the only) float4 group of every stream is requested BEFORE the
  // prologue below (partials of the clip norm, bias corrections), whose latency then hides behind it.
  typedef float nt4 __attribute__((ext_vector_type(4)));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
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
    m0 = __builtin_nontemporal_load(reinterpret_cast<nt4 *>(a.m) + jc);
    v0 = __builtin_nontemporal_load(reinterpret_cast<nt4 *>(a.v) + jc);
    if (a.ema) e0 = __builtin_nontemporal_load(reinterpret_cast<nt4 *>(a.ema) + jc);
  }
  const float lr = a.lr_dev ? a.lr_dev[0] : a.lr;
  const int step = a.step_dev ? a.step_dev[0] : a.step;
  float coef = 1.f;
  if (a.max_norm > 0.f && a.sumsq) {
    // block-wide sum of the partials, same order in every block (L2-resident, a few hundred floats)
    __shared__ float red[4];
    float ss = 0.f;
    for (int i = threadIdx.x; i < a.n_parts; i += 256) ss += a.sumsq[i];
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = (red[0] + red[1]) + (red[2] + red[3]);
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
    // m, v and the EMA shadow are pure streams (touched once per step): non-temporal accesses keep
    // them from evicting the parameters and activations the next step wants in L2 / Infinity Cache
    for (int64_t j = i0; j < n4; j += stride) {
      float4 p = p0, g = g0;
      nt4 mt = m0, vt = v0, et = e0;
      if (j != i0) {
        p = reinterpret_cast<float4 *>(a.p)[j];
        g = reinterpret_cast<const float4 *>(a.g)[j];
        mt = __builtin_nontemporal_load(reinterpret_cast<nt4 *>(a.m) + j);
        vt = __builtin_nontemporal_load(reinterpret_cast<nt4 *>(a.v) + j);
        et = a.ema ? __builtin_nontemporal_load(reinterpret_cast<nt4 *>(a.ema) + j) : (nt4){0, 0, 0, 0};
      }
      float4 m = make_float4(mt.x, mt.y, mt.z, mt.w), v = make_float4(vt.x, vt.y, vt.z, vt.w);
      float4 e = make_float4(et.x, et.y, et.z, et.w);
      float *ep = a.ema ? &e.x : nullptr;
      adam_one(p.x, g.x, m.x, v.x, ep, gm, decay_mul, a.beta1, a.beta2, step_size, inv_sqrt_bc2, a.eps, a.ema_decay);
      adam_one(p.y, g.y, m.y, v.y, ep ? ep + 1 : nullptr, gm, decay_mul, a.beta1, a.beta2, step_size, inv_sqrt_bc2, a.eps, a.ema_decay);
      adam_one(p.z, g.z, m.z, v.z, ep ? ep + 2 : nullptr, gm, decay_mul, a.beta1, a.beta2, step_size, inv_sqrt_bc2, a.eps, a.ema_decay);
      adam_one(p.w, g.w, m.w, v.w, ep ? ep + 3 : nullptr, gm, decay_mul, a.beta1, a.beta2, step_size, inv_sqrt_bc2, a.eps, a.ema_decay);
      reinterpret_cast<float4 *>(a.p)[j] = p;
      __builtin_nontemporal_store((nt4){m.x, m.y, m.z, m.w}, reinterpret_cast<nt4 *>(a.m) + j);
      __builtin_nontemporal_store((nt4){v.x, v.y, v.z, v.w}, reinterpret_cast<nt4 *>(a.v) + j);
      if (a.ema) __builtin_nontemporal_store((nt4){e.x, e.y, e.z, e.w}, reinterpret_cast<nt4 *>(a.ema) + j);
    }
    done = n4 * 4;
  }
  for (int64_t j = done + i0; j < a.n; j += stride) {
    float p = a.p[j], m = a.m[j], v = a.v[j];
    float e = a.ema ? a.ema[j] : 0.f;
    adam_one(p, a.g[j], m, v, a.ema ? &e : nullptr, gm, decay_mul, a.beta1, a.beta2, step_size, inv_sqrt_bc2, a.eps, a.ema_decay);
    a.p[j] = p; a.m[j] = m; a.v[j] = v;
    if (a.ema) a.ema[j] = e;
  }
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

extern "C" int stdadk_sumsq_f32(const float *g, int64_t n, float *parts, int32_t *step_inc,
                                stdadk_stream_t stream) {
  STDADK_REQUIRE(n >= 0, STDADK_E_ARG, "sumsq: negative n");
  STDADK_REQUIRE(parts && (g || n == 0), STDADK_E_ARG, "sumsq: NULL pointer");
  STDADK_LAUNCH(sumsq_kernel, dim3(SUMSQ_PARTS), dim3(256), 0, (hipStream_t)stream, g, n, parts, step_inc);
  STDADK_CHECK_LAUNCH("sumsq");
  return 0;
}

extern "C" int stdadk_adamw_ema_f32(float *p, const float *g, float *m, float *v, float *ema, int64_t n,
                                    float lr, const float *lr_dev, float beta1, float beta2, float eps,
                                    float weight_decay, int32_t step, const int32_t *step_dev, float max_norm,
                                    const float *sumsq_parts, int32_t n_parts, float grad_mul,
                                    float ema_decay, stdadk_stream_t stream) {
  STDADK_REQUIRE(n >= 0, STDADK_E_ARG, "adamw: negative n");
  if (n == 0) return 0;
  STDADK_REQUIRE(p && g && m && v, STDADK_E_ARG, "adamw: NULL pointer");
  STDADK_REQUIRE(step_dev || step >= 1, STDADK_E_ARG, "adamw: step must be >= 1");
  STDADK_REQUIRE(max_norm <= 0.f || (sumsq_parts && n_parts > 0), STDADK_E_ARG, "adamw: max_norm > 0 needs sumsq parts");
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.ema = ema; a.n = n; a.lr = lr; a.lr_dev = lr_dev;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.step = step; a.step_dev = step_dev;
  a.max_norm = max_norm; a.sumsq = sumsq_parts; a.n_parts = n_parts; a.grad_mul = grad_mul; a.ema_decay = ema_decay;
  int64_t blocks = ceil_div(n, 256 * 4);
  if (blocks > 4096) blocks = 4096;
  STDADK_LAUNCH(adamw_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  STDADK_CHECK_LAUNCH("adamw_ema");
  return 0;
}
