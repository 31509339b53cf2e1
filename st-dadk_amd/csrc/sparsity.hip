// Sparsity penalties on the first Linear's weights (stdadk_sparsity_f32, see include/stdadk.h): element-wise
// L1, group lasso per basis function, or both -- penalty value and its (sub-)gradient added to dW0.
// A basis function's group is its row of W0^T (D, H0) resp. its column of W0 (H0, D).  Streaming work:
// one read of W0 and one read-modify-write of dW0.
#include <algorithm>
#include "common.h"
#include "../../include/stdadk.h"

namespace stdadk {

struct SparsityArgs {
  const float *W;
  float *dW;          // may be NULL: penalty values only
  int64_t ld;         // leading dimension of W and dW
  int H;              // hidden[0]
  int row0, n_s, n_t; // basis functions: [row0, row0+n_s) spatial, then n_t temporal
  int apply_s, apply_t;
  float l1, lg;       // 0 switches a term off
  float gscale;       // gradient multiplier (1/world under data parallelism)
  float lscale;       // loss_sum gets lscale x (applied penalties)
  float *loss_sum;    // may be NULL
  float *pen;         // may be NULL: pen[0] += spatial penalty, pen[1] += temporal penalty
};

__device__ __forceinline__ float sgn(float w) { return w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f); }

// block-level sums of the two penalties, then one atomic each.  Adds to ONE address serialise at the memory
// side (measured: ~12 ns each), so the launches below keep the number of workgroups in the low hundreds.
__device__ __forceinline__ void publish(const SparsityArgs &a, float pen_s, float pen_t) {
  __shared__ float red[2][16];
  const float s = wave_sum(pen_s), t = wave_sum(pen_t);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ps = 0.f, pt = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ps += red[0][w]; pt += red[1][w]; }
    if (a.pen) {
      if (ps != 0.f) atomicAdd(a.pen, ps);
      if (pt != 0.f) atomicAdd(a.pen + 1, pt);
    }
    const float applied = (a.apply_s ? ps : 0.f) + (a.apply_t ? pt : 0.f);
    if (a.loss_sum && applied != 0.f) atomicAdd(a.loss_sum, a.lscale * applied);
  }
}

constexpr int ROWS_T = 1024;   // one 16-wave workgroup per CU

// W0^T layout (D, H0): one wave per basis row (contiguous), grid-stride over the rows.  NV > 0: H0 is a multiple
// of 4 and at most 256 * NV, rows 16-byte aligned: a row sits in registers (NV float4 per lane) between the norm
// and the gradient, and the NEXT row of the wave (weights and gradient) is requested before the current one is
// reduced, so a wave always has a row in flight.  NV = 0: any H0, scalar accesses.
template <int NV>
__global__ void __launch_bounds__(ROWS_T) sparsity_rows_kernel(SparsityArgs a) {
  constexpr int WPB = ROWS_T / 64;                        // rows (waves) per workgroup and pass
  const int lane = threadIdx.x & 63;
  const int n = a.n_s + a.n_t, stride = gridDim.x * WPB;
  float pen_s = 0.f, pen_t = 0.f;
  const bool want_pen = a.pen != nullptr;
  if constexpr (NV > 0) {
    const int nv = a.H >> 2;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 x[NV], gv[NV], xn[NV], gn[NV];
    auto fetch = [&](int j, float4 *xo, float4 *go) {
      const bool temporal = j >= a.n_s;
      const bool apply = temporal ? a.apply_t : a.apply_s;
      const float4 *w = reinterpret_cast<const float4 *>(a.W + (int64_t)(a.row0 + j) * a.ld);
      const float4 *g = reinterpret_cast<const float4 *>(a.dW + (int64_t)(a.row0 + j) * a.ld);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        xo[k] = (i < nv && (apply || want_pen)) ? w[i] : zero;
        go[k] = (i < nv && apply && a.dW) ? g[i] : zero;
      }
    };
    int j = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (j < n) fetch(j, x, gv);
    for (; j < n; j += stride) {
      const int jn = j + stride;
      if (jn < n) fetch(jn, xn, gn);
      const bool temporal = j >= a.n_s;
      const bool apply = temporal ? a.apply_t : a.apply_s;
      float sq = 0.f, ab = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        sq += x[k].x * x[k].x + x[k].y * x[k].y + x[k].z * x[k].z + x[k].w * x[k].w;
        ab += fabsf(x[k].x) + fabsf(x[k].y) + fabsf(x[k].z) + fabsf(x[k].w);
      }
      sq = wave_sum(sq); ab = wave_sum(ab);
      const float nrm = sqrtf(sq);
      if (lane == 0 && (apply || want_pen)) {
        const float pen = a.l1 * ab + a.lg * nrm;
        if (temporal) pen_t += pen; else pen_s += pen;
      }
      if (apply && a.dW) {
        float4 *g = reinterpret_cast<float4 *>(a.dW + (int64_t)(a.row0 + j) * a.ld);
        const float inv = nrm > 0.f ? a.lg / nrm : 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const int i = lane + 64 * k;
          if (i < nv) {
            gv[k].x += a.gscale * (a.l1 * sgn(x[k].x) + inv * x[k].x);
            gv[k].y += a.gscale * (a.l1 * sgn(x[k].y) + inv * x[k].y);
            gv[k].z += a.gscale * (a.l1 * sgn(x[k].z) + inv * x[k].z);
            gv[k].w += a.gscale * (a.l1 * sgn(x[k].w) + inv * x[k].w);
            g[i] = gv[k];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) { x[k] = xn[k]; gv[k] = gn[k]; }
    }
  } else {
    for (int j = blockIdx.x * WPB + (threadIdx.x >> 6); j < n; j += stride) {
      const bool temporal = j >= a.n_s;
      const bool apply = temporal ? a.apply_t : a.apply_s;
      if (!(apply || want_pen)) continue;
      const float *w = a.W + (int64_t)(a.row0 + j) * a.ld;
      float sq = 0.f, ab = 0.f;
      for (int i = lane; i < a.H; i += 64) { const float v = w[i]; sq += v * v; ab += fabsf(v); }
      sq = wave_sum(sq); ab = wave_sum(ab);
      const float nrm = sqrtf(sq);
      if (lane == 0) { const float pen = a.l1 * ab + a.lg * nrm; if (temporal) pen_t += pen; else pen_s += pen; }
      if (apply && a.dW) {
        float *g = a.dW + (int64_t)(a.row0 + j) * a.ld;
        const float inv = nrm > 0.f ? a.lg / nrm : 0.f;
        for (int i = lane; i < a.H; i += 64) {
          const float v = w[i];
          g[i] += a.gscale * (a.l1 * sgn(v) + inv * v);
        }
      }
    }
  }
  publish(a, pen_s, pen_t);
}

// W0 layout (H0, D): a workgroup takes 64 adjacent basis columns; its 4 waves split the H0 rows, so every
// load instruction of a wave reads 64 consecutive floats of one row.
__global__ void __launch_bounds__(256) sparsity_cols_kernel(SparsityArgs a) {
  __shared__ float part[2][4][64];
  const int c = threadIdx.x & 63, hs = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  const bool live = j < a.n_s + a.n_t;
  const bool temporal = j >= a.n_s;
  const bool apply = live && (temporal ? a.apply_t : a.apply_s);
  const float *w = a.W + a.row0 + j;
  float sq = 0.f, ab = 0.f;
  if (live && (apply || a.pen))
    for (int h = hs; h < a.H; h += 4) { const float x = w[(int64_t)h * a.ld]; sq += x * x; ab += fabsf(x); }
  part[0][hs][c] = sq; part[1][hs][c] = ab;
  __syncthreads();
  sq = part[0][0][c] + part[0][1][c] + part[0][2][c] + part[0][3][c];
  ab = part[1][0][c] + part[1][1][c] + part[1][2][c] + part[1][3][c];
  const float nrm = sqrtf(sq);
  const float pen = hs == 0 ? a.l1 * ab + a.lg * nrm : 0.f;
  if (apply && a.dW) {
    float *g = a.dW + a.row0 + j;
    const float inv = nrm > 0.f ? a.lg / nrm : 0.f;
    for (int h = hs; h < a.H; h += 4) {
      const float x = w[(int64_t)h * a.ld];
      g[(int64_t)h * a.ld] += a.gscale * (a.l1 * sgn(x) + inv * x);
    }
  }
  __syncthreads();
  publish(a, temporal ? 0.f : pen, temporal ? pen : 0.f);
}

}  // namespace stdadk

using namespace stdadk;

extern "C" int stdadk_sparsity_f32(const stdadk_sparsity_desc *s, const float *W0, float *dW0, int64_t ld,
                                   int32_t w0_t, int32_t H0, int32_t p, int32_t Ks, int32_t Kt,
                                   float grad_scale, float loss_scale, float *loss_sum, float *penalties,
                                   stdadk_stream_t stream) {
  STDADK_REQUIRE(s, STDADK_E_ARG, "sparsity: NULL descriptor");
  STDADK_REQUIRE(s->kind == STDADK_SPARSITY_NONE || s->kind == STDADK_SPARSITY_ELEMENT ||
                     s->kind == STDADK_SPARSITY_GROUP || s->kind == STDADK_SPARSITY_SPARSE_GROUP,
                 STDADK_E_ARG, "Unknown penalty_type: %d", s->kind);
  STDADK_REQUIRE(H0 >= 1 && p >= 0 && Ks >= 0 && Kt >= 0, STDADK_E_ARG, "sparsity: bad sizes H0=%d p=%d Ks=%d Kt=%d",
                 H0, p, Ks, Kt);
  STDADK_REQUIRE(ld >= (w0_t ? (int64_t)H0 : (int64_t)p + Ks + Kt), STDADK_E_ARG,
                 "sparsity: ld=%lld shorter than a row", (long long)ld);
  if (s->kind == STDADK_SPARSITY_NONE || Ks + Kt == 0) return 0;
  STDADK_REQUIRE(W0, STDADK_E_ARG, "sparsity: NULL weights");
  SparsityArgs a;
  a.W = W0; a.dW = dW0; a.ld = ld; a.H = H0; a.row0 = p; a.n_s = Ks; a.n_t = Kt;
  a.apply_s = s->apply_spatial != 0; a.apply_t = s->apply_temporal != 0;
  a.l1 = s->kind == STDADK_SPARSITY_GROUP ? 0.f : s->lambda_l1;
  a.lg = s->kind == STDADK_SPARSITY_ELEMENT ? 0.f : s->lambda_group;
  a.gscale = grad_scale; a.lscale = loss_scale; a.loss_sum = loss_sum; a.pen = penalties;
  if (!penalties && !a.apply_s && !a.apply_t) return 0;
  const int n = Ks + Kt;
  if (w0_t) {
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(n, ROWS_T / 64), 256);
    const bool vec = (H0 & 3) == 0 && H0 <= 1024 && (ld & 3) == 0 && aligned16(W0) && (!dW0 || aligned16(dW0));
    if (vec && H0 <= 256) {
      STDADK_LAUNCH(sparsity_rows_kernel<1>, dim3(grid), dim3(ROWS_T), 0, (hipStream_t)stream, a);
    } else if (vec && H0 <= 512) {
      STDADK_LAUNCH(sparsity_rows_kernel<2>, dim3(grid), dim3(ROWS_T), 0, (hipStream_t)stream, a);
    } else if (vec) {
      STDADK_LAUNCH(sparsity_rows_kernel<4>, dim3(grid), dim3(ROWS_T), 0, (hipStream_t)stream, a);
    } else {
      STDADK_LAUNCH(sparsity_rows_kernel<0>, dim3(grid), dim3(ROWS_T), 0, (hipStream_t)stream, a);
    }
  } else {
    STDADK_LAUNCH(sparsity_cols_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, (hipStream_t)stream, a);
  }
  STDADK_CHECK_LAUNCH("sparsity");
  return 0;
}
