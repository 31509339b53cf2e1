// N2 kernels: see knots.h.  Dense formulation: dFeat = dZ0 . W0 comes from the fp32 MFMA GEMM, the
// kernels here turn its spatial columns into per-knot gradients.
#include "knots.h"
#include "basis.h"

namespace stdadk {

int knot_slabs(int64_t B) {
  int64_t s = ceil_div(B, 128);
  return (int)(s < 1 ? 1 : (s > KNOT_MAX_SLABS ? KNOT_MAX_SLABS : s));
}

__global__ void exp_kernel(const float *__restrict__ in, int64_t n, float *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = expf(in[i]);
}

int launch_exp(const float *in, int64_t n, float *out, hipStream_t st) {
  if (n <= 0) return 0;
  STDADK_LAUNCH(exp_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, in, n, out);
  STDADK_CHECK_LAUNCH("exp");
  return 0;
}

// d phi / d r as autograd differentiates the forward formulas: Wendland (r clamped at 1)
// -(56/3) r (1-r)^5 (5r+1); Gaussian -r exp(-r^2/2); triangular -1 where 1-r >= 0.
template <int BASIS>
__device__ __forceinline__ float basis_prime(float r) {
  if (BASIS == STDADK_BASIS_WENDLAND) {
    if (!(r < 1.0f)) return 0.f;
    float om = 1.0f - r;
    float om2 = om * om;
    return (-56.0f / 3.0f) * r * (om2 * om2 * om) * fmaf(5.0f, r, 1.0f);
  } else if (BASIS == STDADK_BASIS_GAUSSIAN) {
    return -r * expf(-0.5f * r * r);
  } else {
    return r <= 1.0f ? -1.0f : 0.f;
  }
}

// Lane <-> knot, the 4 waves of a workgroup split the rows of one slab; the per-row observation is
// wave-uniform (scalar loads), the dFeat row segment is one coalesced 256-byte load per wave.
template <int BASIS>
__global__ __launch_bounds__(256) void knot_grad_kernel(KnotGradArgs a, float cal) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = blockIdx.x * KNOT_TILE + lane;
  const int kc = k < a.Ks ? k : a.Ks - 1;           // clamped: loads stay in bounds, result discarded
  const int rows_per = (a.B + a.slabs - 1) / a.slabs;
  const int r0 = blockIdx.y * rows_per;
  const int r1 = min(a.B, r0 + rows_per);
  const float cx = a.centers[2 * kc], cy = a.centers[2 * kc + 1];
  const float inv_s = knot_scale(a.bw[kc], cal);
  const float *g = a.dFeat + a.p + kc;
  float acx = 0.f, acy = 0.f, alb = 0.f;
  for (int b = r0 + wave; b < r1; b += 4) {
    const float x = a.coords[2 * b], y = a.coords[2 * b + 1];
    const float G = g[(int64_t)b * a.ld];
    const float dx = x - cx, dy = y - cy;
    const float d = __builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy));
    const float r = d * inv_s;
    const float gr = G * basis_prime<BASIS>(r);
    const float q = d > 0.f ? gr * inv_s / d : 0.f;     // cdist's backward: no pull at zero distance
    acx = fmaf(-q, dx, acx);
    acy = fmaf(-q, dy, acy);
    alb = fmaf(-gr, r, alb);
  }
  __shared__ float red[3][4][KNOT_TILE];
  red[0][wave][lane] = acx; red[1][wave][lane] = acy; red[2][wave][lane] = alb;
  __syncthreads();
  if (threadIdx.x < 3 * KNOT_TILE) {
    const int c = threadIdx.x / KNOT_TILE, l = threadIdx.x - c * KNOT_TILE;
    const int kk = blockIdx.x * KNOT_TILE + l;
    if (kk < a.Ks)
      a.part[((size_t)blockIdx.y * 3 + c) * a.Ks + kk] = (red[c][0][l] + red[c][1][l]) + (red[c][2][l] + red[c][3][l]);
  }
}

int launch_knot_grad(const KnotGradArgs &a, hipStream_t st) {
  static const float cals[3] = {1.000000f, 0.223477f, 0.654714f};   // st_interp.py:56-60
  dim3 grid((unsigned)ceil_div(a.Ks, KNOT_TILE), (unsigned)a.slabs);
  switch (a.basis) {
    case STDADK_BASIS_WENDLAND:
      STDADK_LAUNCH(knot_grad_kernel<STDADK_BASIS_WENDLAND>, grid, dim3(256), 0, st, a, cals[0]); break;
    case STDADK_BASIS_GAUSSIAN:
      STDADK_LAUNCH(knot_grad_kernel<STDADK_BASIS_GAUSSIAN>, grid, dim3(256), 0, st, a, cals[1]); break;
    default:
      STDADK_LAUNCH(knot_grad_kernel<STDADK_BASIS_TRIANGULAR>, grid, dim3(256), 0, st, a, cals[2]); break;
  }
  STDADK_CHECK_LAUNCH("knot_grad");
  return 0;
}

// per knot: fixed-order sum of the slab partials, + penalty gradients, x damping factor.  64 knots per
// workgroup; its 4 waves each sum every 4th slab (independent loads), the four partial sums meet in LDS.
__global__ __launch_bounds__(256) void knot_finish_kernel(KnotFinishArgs a) {
  __shared__ float part[3][4][64];
  __shared__ float red[4];
  const int kl = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kl;
  float gx = 0.f, gy = 0.f, gl = 0.f;
  if (k < a.Ks) {
    for (int s = sg; s < a.slabs; s += 4) {
      gx += a.part[((size_t)s * 3 + 0) * a.Ks + k];
      gy += a.part[((size_t)s * 3 + 1) * a.Ks + k];
      gl += a.part[((size_t)s * 3 + 2) * a.Ks + k];
    }
  }
  part[0][sg][kl] = gx; part[1][sg][kl] = gy; part[2][sg][kl] = gl;
  __syncthreads();
  float pen = 0.f;
  if (sg == 0 && k < a.Ks) {
    gx = (part[0][0][kl] + part[0][1][kl]) + (part[0][2][kl] + part[0][3][kl]);
    gy = (part[1][0][kl] + part[1][1][kl]) + (part[1][2][kl] + part[1][3][kl]);
    gl = (part[2][0][kl] + part[2][1][kl]) + (part[2][2][kl] + part[2][3][kl]);
    const float cx = a.centers[2 * k], cy = a.centers[2 * k + 1];
    float mx = 0.f, my = 0.f;
    if (a.centers_init) { mx = cx - a.centers_init[2 * k]; my = cy - a.centers_init[2 * k + 1]; }
    if (a.dom_w > 0.f) {
      // (max(0, 0-c) + max(0, c-1))^2 per coordinate (st_interp.py:511-524)
      const float vx = fmaxf(0.f - cx, 0.f) + fmaxf(cx - 1.0f, 0.f);
      const float vy = fmaxf(0.f - cy, 0.f) + fmaxf(cy - 1.0f, 0.f);
      pen = fmaf(a.dom_w, vx * vx + vy * vy, pen);
      gx = fmaf(a.pen_grad_scale * a.dom_w, 2.0f * vx * (cx > 1.0f ? 1.0f : (cx < 0.f ? -1.0f : 0.f)), gx);
      gy = fmaf(a.pen_grad_scale * a.dom_w, 2.0f * vy * (cy > 1.0f ? 1.0f : (cy < 0.f ? -1.0f : 0.f)), gy);
    }
    if (a.mov_w > 0.f) {
      pen = fmaf(a.mov_w, mx * mx + my * my, pen);
      gx = fmaf(a.pen_grad_scale * a.mov_w, 2.0f * mx, gx);
      gy = fmaf(a.pen_grad_scale * a.mov_w, 2.0f * my, gy);
    }
    if (a.damping) {
      const float dist = __builtin_amdgcn_sqrtf(fmaf(mx, mx, my * my));
      const float f = expf(-a.strength * fmaxf(dist - a.thr, 0.f));
      gx *= f; gy *= f;
    }
    a.d_centers[2 * k] = gx; a.d_centers[2 * k + 1] = gy;
    a.d_log_bw[k] = gl;
  }
  if (a.loss_sum && a.pen_loss_scale != 0.f && (a.dom_w > 0.f || a.mov_w > 0.f)) {
    const float s = wave_sum(pen);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.loss_sum, a.pen_loss_scale * ((red[0] + red[1]) + (red[2] + red[3])));
  }
}

int launch_knot_finish(const KnotFinishArgs &a, hipStream_t st) {
  STDADK_LAUNCH(knot_finish_kernel, dim3((unsigned)ceil_div(a.Ks, 64)), dim3(256), 0, st, a);
  STDADK_CHECK_LAUNCH("knot_finish");
  return 0;
}

}  // namespace stdadk

// ---------------------------------------------------------------------------------------------------------
// Module-level autograd of SpatialBasisEmbedding.forward with learnable knots: dL/dphi in, knot gradients out
// ---------------------------------------------------------------------------------------------------------
using namespace stdadk;

extern "C" size_t stdadk_knot_grad_workspace_bytes(int64_t B, int64_t Ks) {
  if (B < 0 || Ks < 0) return 0;
  const size_t slabs = (size_t)knot_slabs(B > 0 ? B : 1);
  return (align_up((size_t)(Ks > 0 ? Ks : 1), 64) + align_up(slabs * 3 * (size_t)(Ks > 0 ? Ks : 1), 64)) * sizeof(float);
}

extern "C" int stdadk_knot_grad_f32(const float *coords, int64_t B, const float *d_phi, int64_t ld,
                                    const float *centers, const float *log_bw, int64_t Ks, int32_t basis,
                                    float *d_centers, float *d_log_bw, void *workspace, size_t workspace_bytes,
                                    stdadk_stream_t stream) {
  STDADK_REQUIRE(B >= 0 && B < (1ll << 31) && Ks >= 0 && Ks < (1ll << 31) && basis >= 0 && basis <= 2, STDADK_E_ARG,
                 "knot_grad: bad sizes / basis");
  if (Ks == 0) return 0;
  STDADK_REQUIRE(centers && log_bw && d_centers && d_log_bw, STDADK_E_ARG, "knot_grad: NULL pointer");
  STDADK_REQUIRE(B == 0 || (coords && d_phi && ld >= Ks), STDADK_E_ARG, "knot_grad: NULL batch pointer or ld < Ks");
  STDADK_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= stdadk_knot_grad_workspace_bytes(B, Ks),
                 STDADK_E_WORKSPACE, "knot_grad: workspace NULL, unaligned or smaller than stdadk_knot_grad_workspace_bytes");
  hipStream_t st = (hipStream_t)stream;
  float *bw = (float *)workspace;
  float *part = bw + align_up((size_t)Ks, 64);
  int rc = launch_exp(log_bw, Ks, bw, st);
  if (rc) return rc;
  KnotFinishArgs fa;
  fa.part = part; fa.slabs = B > 0 ? knot_slabs(B) : 0; fa.Ks = (int)Ks; fa.centers = centers; fa.centers_init = nullptr;
  fa.damping = 0; fa.thr = fa.strength = 0.f; fa.dom_w = fa.mov_w = 0.f; fa.pen_grad_scale = fa.pen_loss_scale = 0.f;
  fa.d_centers = d_centers; fa.d_log_bw = d_log_bw; fa.loss_sum = nullptr;
  if (B > 0) {
    KnotGradArgs ka;
    ka.coords = coords; ka.B = (int)B; ka.dFeat = d_phi; ka.ld = ld; ka.p = 0;
    ka.centers = centers; ka.bw = bw; ka.Ks = (int)Ks; ka.basis = basis; ka.part = part; ka.slabs = fa.slabs;
    rc = launch_knot_grad(ka, st);
    if (rc) return rc;
  }
  return launch_knot_finish(fa, st);       // B == 0: no slabs, the gradients come out as zeros
}
