// A6-A8: dense MLP forward / backward around the fp32 MFMA GEMM, plus the fused MSE loss.
//
//   forward  (stnf/models/st_interp.py:656-690,880):  per hidden layer
//       z = a_prev W^T (+b)          split-K MFMA GEMM -> partial slabs
//       LN -> ReLU -> Dropout        one wave per row; sums the slabs, adds the bias, saves xhat/rstd
//   backward (scripts/train_st_interp.py:693): per hidden layer, last to first
//       dz, partial column sums      one wave per row (LN backward, ReLU/dropout mask recomputed)
//       dgamma, dbeta, db            column-sum reduction of the per-workgroup partials
//       dW = dz^T a_prev             split-K (over the batch) MFMA GEMM -> slabs -> sum
//       da_prev = dz W               MFMA GEMM (skipped for the first layer: knots are buffers)
#include "common.h"
#include "gemm_f32.h"

namespace stdadk {

constexpr int ROW_T = 256;        // threads of the row kernels: 4 waves = 4 rows in flight
constexpr int MAX_CPL = 16;       // columns per lane => hidden width <= 1024
constexpr int BWD_ROWS = 16;      // rows per workgroup in the backward row kernel

// ---------------------------------------------------------------------------------------------
// workspace plan
// ---------------------------------------------------------------------------------------------
struct Plan {
  int L;
  int64_t B;
  size_t xhat[STDADK_MAX_HIDDEN], rstd[STDADK_MAX_HIDDEN], act[STDADK_MAX_HIDDEN];  // float offsets
  size_t slab, slab_floats;
  size_t dA, dZ;            // [B][hmax] each
  size_t part, part_floats; // column-sum partials
  size_t total_floats;
};

static void make_plan(const stdadk_mlp_desc *d, int64_t B, Plan *p) {
  p->L = d->n_hidden;
  p->B = B;
  size_t off = 0;
  auto take = [&](size_t n) {
    size_t o = off;
    off += align_up(n, 64);   // 256-byte granules keep every sub-buffer float4-aligned
    return o;
  };
  int hmax = d->out_dim;
  size_t slab = 0;
  int prev = d->in_dim;
  for (int l = 0; l < d->n_hidden; ++l) {
    int h = d->hidden[l];
    p->xhat[l] = take((size_t)B * h);
    p->rstd[l] = take((size_t)B);
    p->act[l] = take((size_t)B * h);
    hmax = h > hmax ? h : hmax;
    size_t s;
    s = gemm_slab_floats((int)B, h, prev); slab = s > slab ? s : slab;          // forward z
    s = gemm_slab_floats(h, prev, (int)B); slab = s > slab ? s : slab;          // dW
    if (l > 0) { s = gemm_slab_floats((int)B, prev, h); slab = s > slab ? s : slab; }  // dA
    prev = h;
  }
  // the output layer (Q > 8) goes through the GEMM as well
  {
    size_t s = gemm_slab_floats((int)B, d->out_dim, prev); slab = s > slab ? s : slab;
    s = gemm_slab_floats(d->out_dim, prev, (int)B); slab = s > slab ? s : slab;
    s = gemm_slab_floats((int)B, prev, d->out_dim); slab = s > slab ? s : slab;
  }
  p->slab_floats = slab;
  p->slab = take(slab);
  p->dA = take((size_t)B * hmax);
  p->dZ = take((size_t)B * hmax);
  int64_t nblk = ceil_div(B, BWD_ROWS);
  p->part_floats = (size_t)nblk * 3 * hmax;
  size_t head = (size_t)nblk * (size_t)d->out_dim * (hmax + 1);
  if (head > p->part_floats) p->part_floats = head;
  p->part = take(p->part_floats);
  p->total_floats = off;
}

// ---------------------------------------------------------------------------------------------
// row kernels
// ---------------------------------------------------------------------------------------------
// z[row][c] = sum_s slab[s][row][c] (+ bias[c]); LayerNorm -> ReLU -> Dropout.  One wave per row.
template <bool LN, int CPL>
__global__ __launch_bounds__(ROW_T) void ln_relu_fwd_kernel(
    const float *__restrict__ zsrc, int splits, int64_t slab_stride, const float *__restrict__ bias,
    const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int64_t B, int h,
    float *__restrict__ xhat, float *__restrict__ rstd_out, float *__restrict__ act, float drop_p,
    uint64_t seed, int layer, const uint8_t *__restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (ROW_T / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  float z[CPL];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    int c = lane + 64 * j;
    z[j] = 0.f;
    if (c < h) {
      float v = bias ? bias[c] : 0.f;
      for (int s = 0; s < splits; ++s) v += zsrc[(int64_t)s * slab_stride + row * h + c];
      z[j] = v;
      sum += v;
    }
  }
  float rs = 1.f, mean = 0.f;
  if (LN) {
    mean = wave_sum(sum) / (float)h;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (lane + 64 * j < h) { float d = z[j] - mean; sq += d * d; }
    float var = wave_sum(sq) / (float)h;
    rs = 1.0f / sqrtf(var + eps);
    if (lane == 0) rstd_out[row] = rs;
  }
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    int c = lane + 64 * j;
    if (c < h) {
      float xh = LN ? (z[j] - mean) * rs : z[j];
      float u = LN ? fmaf(xh, gamma[c], beta[c]) : xh;
      float a = fmaxf(u, 0.f);
      if (drop_p > 0.f) {
        int64_t e = row * h + c;
        bool keep = mask ? (mask[e] != 0) : drop_keep(seed, layer, e, drop_p);
        a = keep ? a * keep_scale : 0.f;
      }
      xhat[row * h + c] = xh;
      act[row * h + c] = a;
    }
  }
}

// Backward of Dropout -> ReLU -> LayerNorm for BWD_ROWS rows per workgroup (4 rows per wave):
//   du = dA * keep/(1-p) * (u > 0);  dgamma += du*xhat;  dbeta += du
//   dz = rstd * (dxh - mean(dxh) - xhat*mean(dxh*xhat)),  dxh = du*gamma;   db += dz
// Column partials go to part[blk][3][h] (dgamma, dbeta, db) and are summed by colsum_kernel.
template <bool LN, int CPL>
__global__ __launch_bounds__(ROW_T) void ln_relu_bwd_kernel(
    const float *__restrict__ dA, const float *__restrict__ xhat, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, int64_t B, int h,
    float *__restrict__ dZ, float *__restrict__ part, float drop_p, uint64_t seed, int layer,
    const uint8_t *__restrict__ mask) {
  __shared__ float red[3][ROW_T / 64][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  float pg[CPL], pb[CPL], pz[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) pg[j] = pb[j] = pz[j] = 0.f;
  const int64_t row0 = (int64_t)blockIdx.x * BWD_ROWS;
  for (int rr = wave; rr < BWD_ROWS; rr += ROW_T / 64) {
    const int64_t row = row0 + rr;
    if (row >= B) break;
    float du[CPL], xh[CPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      int c = lane + 64 * j;
      du[j] = 0.f; xh[j] = 0.f;
      if (c < h) {
        int64_t e = row * h + c;
        float x = xhat[e];
        float u = LN ? fmaf(x, gamma[c], beta[c]) : x;
        float d = dA[e];
        if (drop_p > 0.f) {
          bool keep = mask ? (mask[e] != 0) : drop_keep(seed, layer, e, drop_p);
          d = keep ? d * keep_scale : 0.f;
        }
        d = u > 0.f ? d : 0.f;
        du[j] = d; xh[j] = x;
        if (LN) {
          pg[j] += d * x;
          pb[j] += d;
          float dxh = d * gamma[c];
          s1 += dxh;
          s2 += dxh * x;
        }
      }
    }
    float rs = 1.f, m1 = 0.f, m2 = 0.f;
    if (LN) {
      m1 = wave_sum(s1) / (float)h;
      m2 = wave_sum(s2) / (float)h;
      rs = rstd[row];
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      int c = lane + 64 * j;
      if (c < h) {
        float dz = LN ? rs * (du[j] * gamma[c] - m1 - xh[j] * m2) : du[j];
        dZ[row * h + c] = dz;
        pz[j] += dz;
      }
    }
  }
  // cross-wave reduction of the column partials, 256 columns at a time
  float *pbase = part + (int64_t)blockIdx.x * 3 * h;
#pragma unroll
  for (int j0 = 0; j0 < CPL; j0 += 4) {
    if (64 * j0 < h) {   // workgroup-uniform
      __syncthreads();
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        if (j0 + jj < CPL) {
          red[0][wave][lane + 64 * jj] = pg[j0 + jj];
          red[1][wave][lane + 64 * jj] = pb[j0 + jj];
          red[2][wave][lane + 64 * jj] = pz[j0 + jj];
        }
      }
      __syncthreads();
      int c = 64 * j0 + threadIdx.x;   // 256 threads <-> 256 columns
      if (c < h) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          float s = red[q][0][threadIdx.x] + red[q][1][threadIdx.x] + red[q][2][threadIdx.x] + red[q][3][threadIdx.x];
          pbase[q * h + c] = s;
        }
      }
    }
  }
}

// out[c] = sum_blk part[blk*stride + c]  for c < n  (one thread per column, coalesced over c)
__global__ void colsum_kernel(const float *__restrict__ part, int64_t nblk, int64_t stride, int n,
                              float *__restrict__ out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t b = 0;
  for (; b + 3 < nblk; b += 4) {
    s0 += part[(b + 0) * stride + c];
    s1 += part[(b + 1) * stride + c];
    s2 += part[(b + 2) * stride + c];
    s3 += part[(b + 3) * stride + c];
  }
  for (; b < nblk; ++b) s0 += part[b * stride + c];
  out[c] = (s0 + s1) + (s2 + s3);
}

// Output layer for small Q: y[row][q] = a[row,:] . W[q,:] + b[q].  One wave per row.
constexpr int HEAD_MAXQ = 8;
__global__ __launch_bounds__(ROW_T) void head_fwd_kernel(const float *__restrict__ a, int64_t B, int h,
                                                         const float *__restrict__ W,
                                                         const float *__restrict__ b, int Q,
                                                         float *__restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (ROW_T / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  float acc[HEAD_MAXQ];
#pragma unroll
  for (int q = 0; q < HEAD_MAXQ; ++q) acc[q] = 0.f;
  for (int c = lane; c < h; c += 64) {
    float v = a[row * h + c];
#pragma unroll
    for (int q = 0; q < HEAD_MAXQ; ++q)
      if (q < Q) acc[q] = fmaf(v, W[q * h + c], acc[q]);
  }
#pragma unroll
  for (int q = 0; q < HEAD_MAXQ; ++q) {
    if (q < Q) {
      float s = wave_sum(acc[q]);
      if (lane == 0) y[row * Q + q] = s + b[q];
    }
  }
}

// Backward of the small-Q output layer for BWD_ROWS rows per workgroup:
//   dA[row][c] = sum_q dY[row][q] W[q][c];  partial dW[q][c] = sum_rows dY[row][q] a[row][c];
//   partial db[q] = sum_rows dY[row][q].   part[blk][q][h+1] (last column = db).
__global__ __launch_bounds__(ROW_T) void head_bwd_kernel(const float *__restrict__ a,
                                                         const float *__restrict__ dY, int64_t B,
                                                         int h, const float *__restrict__ W, int Q,
                                                         float *__restrict__ dA,
                                                         float *__restrict__ part) {
  const int64_t row0 = (int64_t)blockIdx.x * BWD_ROWS;
  const int nrow = (int)min((int64_t)BWD_ROWS, B - row0);
  __shared__ float sdy[BWD_ROWS * HEAD_MAXQ];
  for (int i = threadIdx.x; i < nrow * Q; i += ROW_T) sdy[i] = dY[row0 * Q + i];
  __syncthreads();
  float *pbase = part + (int64_t)blockIdx.x * Q * (h + 1);
  for (int c = threadIdx.x; c < h; c += ROW_T) {
    float w[HEAD_MAXQ], pw[HEAD_MAXQ];
#pragma unroll
    for (int q = 0; q < HEAD_MAXQ; ++q) { w[q] = q < Q ? W[q * h + c] : 0.f; pw[q] = 0.f; }
    for (int r = 0; r < nrow; ++r) {
      float av = a[(row0 + r) * h + c];
      float d = 0.f;
#pragma unroll
      for (int q = 0; q < HEAD_MAXQ; ++q)
        if (q < Q) { float g = sdy[r * Q + q]; d = fmaf(g, w[q], d); pw[q] = fmaf(g, av, pw[q]); }
      dA[(row0 + r) * h + c] = d;
    }
#pragma unroll
    for (int q = 0; q < HEAD_MAXQ; ++q)
      if (q < Q) pbase[q * (h + 1) + c] = pw[q];
  }
  if (threadIdx.x < Q) {
    float s = 0.f;
    for (int r = 0; r < nrow; ++r) s += sdy[r * Q + threadIdx.x];
    pbase[threadIdx.x * (h + 1) + h] = s;
  }
}

// split the head partial sums [Q][h+1] into dW[Q][h] and db[Q]
__global__ void head_reduce_kernel(const float *__restrict__ part, int64_t nblk, int Q, int h,
                                   float *__restrict__ dW, float *__restrict__ db) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int n = Q * (h + 1);
  if (i >= n) return;
  float s = 0.f;
  for (int64_t b = 0; b < nblk; ++b) s += part[b * n + i];
  int q = i / (h + 1), c = i - q * (h + 1);
  if (c < h) dW[q * h + c] = s; else db[q] = s;
}

__global__ void mse_kernel(const float *__restrict__ yp, const float *__restrict__ y, int64_t n,
                           float scale, float *__restrict__ dY, float *__restrict__ loss_sum) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float d = yp[i] - y[i];
    acc = fmaf(d, d, acc);
    if (dY) dY[i] = 2.0f * d * scale;
  }
  if (loss_sum) {
    __shared__ float red[4];
    float s = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
  }
}

// ---------------------------------------------------------------------------------------------
// host-side orchestration
// ---------------------------------------------------------------------------------------------
static int check_desc(const stdadk_mlp_desc *d) {
  STDADK_REQUIRE(d != nullptr, STDADK_E_ARG, "mlp: desc is NULL");
  STDADK_REQUIRE(d->n_hidden >= 0 && d->n_hidden <= STDADK_MAX_HIDDEN, STDADK_E_ARG,
                 "mlp: n_hidden %d out of range", d->n_hidden);
  STDADK_REQUIRE(d->in_dim > 0 && d->out_dim > 0, STDADK_E_ARG, "mlp: in_dim/out_dim must be > 0");
  for (int l = 0; l < d->n_hidden; ++l)
    STDADK_REQUIRE(d->hidden[l] > 0 && d->hidden[l] <= 64 * MAX_CPL, STDADK_E_SHAPE,
                   "mlp: hidden[%d]=%d unsupported (1..%d)", l, d->hidden[l], 64 * MAX_CPL);
  STDADK_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, STDADK_E_ARG, "mlp: dropout_p out of range");
  return 0;
}

}  // namespace stdadk

using namespace stdadk;

extern "C" size_t stdadk_mlp_workspace_bytes(const stdadk_mlp_desc *desc, int64_t B) {
  if (check_desc(desc) != 0 || B < 0) return 0;
  Plan p;
  make_plan(desc, B > 0 ? B : 1, &p);
  return p.total_floats * sizeof(float);
}

extern "C" int stdadk_mlp_forward_f32(const stdadk_mlp_desc *d, const stdadk_mlp_tensors *P,
                                      const float *features, int64_t ldf, int64_t B, float *y_pred,
                                      void *workspace, size_t workspace_bytes, int32_t training,
                                      uint64_t drop_seed, const uint8_t *const *drop_mask,
                                      stdadk_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  STDADK_REQUIRE(B >= 0 && B < (1ll << 31), STDADK_E_ARG, "mlp_forward: bad B");
  if (B == 0) return 0;
  STDADK_REQUIRE(P && features && y_pred && workspace, STDADK_E_ARG, "mlp_forward: NULL pointer");
  STDADK_REQUIRE(ldf >= d->in_dim, STDADK_E_SHAPE, "mlp_forward: ldf %lld < in_dim %d", (long long)ldf, d->in_dim);
  STDADK_REQUIRE(aligned16(workspace), STDADK_E_ALIGN, "mlp_forward: workspace must be 16-byte aligned");
  Plan pl;
  make_plan(d, B, &pl);
  STDADK_REQUIRE(workspace_bytes >= pl.total_floats * sizeof(float), STDADK_E_WORKSPACE,
                 "mlp_forward: workspace %zu < %zu bytes", workspace_bytes, pl.total_floats * sizeof(float));
  float *ws = (float *)workspace;
  hipStream_t st = (hipStream_t)stream;
  const float dp = training ? d->dropout_p : 0.f;

  const float *in = features;
  int64_t ld_in = ldf;
  int K = d->in_dim;
  for (int l = 0; l < d->n_hidden; ++l) {
    const int h = d->hidden[l];
    STDADK_REQUIRE(P->W[l] && P->b[l], STDADK_E_ARG, "mlp_forward: layer %d weights NULL", l);
    STDADK_REQUIRE(!d->layernorm || (P->ln_g[l] && P->ln_b[l]), STDADK_E_ARG, "mlp_forward: layer %d LN NULL", l);
    float *xh = ws + pl.xhat[l], *act = ws + pl.act[l];
    int splits = 1;
    // z partials: slabs when split, else straight into the xhat buffer (overwritten in place below)
    rc = gemm_run(in, ld_in, false, P->W[l], K, false, (int)B, h, K, nullptr, xh, h, ws + pl.slab, true, &splits, st);
    if (rc) return rc;
    const float *zsrc = splits > 1 ? ws + pl.slab : xh;
    const uint8_t *mk = (drop_mask && dp > 0.f) ? drop_mask[l] : nullptr;
    dim3 grid((unsigned)ceil_div(B, ROW_T / 64));
#define FWD(LN_, CPL_)                                                                                   \
  hipLaunchKernelGGL((ln_relu_fwd_kernel<LN_, CPL_>), grid, dim3(ROW_T), 0, st, zsrc, splits,            \
                     (int64_t)B * h, P->b[l], LN_ ? P->ln_g[l] : (const float *)nullptr,                 \
                     LN_ ? P->ln_b[l] : (const float *)nullptr, d->ln_eps, B, h, xh, ws + pl.rstd[l],    \
                     act, dp, drop_seed, l, mk)
    if (d->layernorm) { if (h <= 64) FWD(true, 1); else if (h <= 128) FWD(true, 2); else if (h <= 256) FWD(true, 4); else FWD(true, 16); }
    else { if (h <= 64) FWD(false, 1); else if (h <= 128) FWD(false, 2); else if (h <= 256) FWD(false, 4); else FWD(false, 16); }
#undef FWD
    STDADK_CHECK_LAUNCH("ln_relu_fwd");
    in = act; ld_in = h; K = h;
  }
  const int L = d->n_hidden, Q = d->out_dim;
  STDADK_REQUIRE(P->W[L] && P->b[L], STDADK_E_ARG, "mlp_forward: output layer weights NULL");
  if (Q <= HEAD_MAXQ) {
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)ceil_div(B, ROW_T / 64)), dim3(ROW_T), 0, st, in, B,
                       K, P->W[L], P->b[L], Q, y_pred);
    STDADK_CHECK_LAUNCH("head_fwd");
  } else {
    STDADK_REQUIRE(ld_in == K, STDADK_E_SHAPE, "mlp_forward: unexpected ld");
    rc = gemm_run(in, ld_in, false, P->W[L], K, false, (int)B, Q, K, P->b[L], y_pred, Q, ws + pl.slab, false, nullptr, st);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int stdadk_mlp_backward_f32(const stdadk_mlp_desc *d, const stdadk_mlp_tensors *P,
                                       const stdadk_mlp_tensors *G, const float *features,
                                       int64_t ldf, int64_t B, const float *dY, void *workspace,
                                       size_t workspace_bytes, uint64_t drop_seed,
                                       const uint8_t *const *drop_mask, stdadk_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  STDADK_REQUIRE(B > 0 && B < (1ll << 31), STDADK_E_ARG, "mlp_backward: bad B");
  STDADK_REQUIRE(P && G && features && dY && workspace, STDADK_E_ARG, "mlp_backward: NULL pointer");
  STDADK_REQUIRE(ldf >= d->in_dim, STDADK_E_SHAPE, "mlp_backward: ldf < in_dim");
  Plan pl;
  make_plan(d, B, &pl);
  STDADK_REQUIRE(workspace_bytes >= pl.total_floats * sizeof(float), STDADK_E_WORKSPACE,
                 "mlp_backward: workspace too small");
  float *ws = (float *)workspace;
  hipStream_t st = (hipStream_t)stream;
  const int L = d->n_hidden, Q = d->out_dim;
  const float dp = d->dropout_p;
  const int64_t nblk = ceil_div(B, BWD_ROWS);
  float *dA = ws + pl.dA, *dZ = ws + pl.dZ, *part = ws + pl.part, *slab = ws + pl.slab;

  // ---- output layer
  const float *aL = L > 0 ? ws + pl.act[L - 1] : features;
  const int64_t ldaL = L > 0 ? d->hidden[L - 1] : ldf;
  const int hL = L > 0 ? d->hidden[L - 1] : d->in_dim;
  STDADK_REQUIRE(G->W[L] && G->b[L], STDADK_E_ARG, "mlp_backward: output layer grads NULL");
  if (Q <= HEAD_MAXQ && L > 0) {
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)nblk), dim3(ROW_T), 0, st, aL, dY, B, hL, P->W[L], Q, dA, part);
    STDADK_CHECK_LAUNCH("head_bwd");
    int n = Q * (hL + 1);
    hipLaunchKernelGGL(head_reduce_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, part, nblk, Q, hL, G->W[L], G->b[L]);
    STDADK_CHECK_LAUNCH("head_reduce");
  } else {
    // dW_out[Q][hL] = dY^T a ; db = colsum(dY) ; dA = dY W_out
    rc = gemm_run(dY, Q, true, aL, ldaL, true, Q, hL, (int)B, nullptr, G->W[L], hL, slab, false, nullptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ceil_div(Q, 256)), dim3(256), 0, st, dY, B, (int64_t)Q, Q, G->b[L]);
    STDADK_CHECK_LAUNCH("colsum");
    if (L > 0) {
      rc = gemm_run(dY, Q, false, P->W[L], hL, true, (int)B, hL, Q, nullptr, dA, hL, slab, false, nullptr, st);
      if (rc) return rc;
    }
  }

  // ---- hidden layers, last to first
  for (int l = L - 1; l >= 0; --l) {
    const int h = d->hidden[l];
    const uint8_t *mk = (drop_mask && dp > 0.f) ? drop_mask[l] : nullptr;
    STDADK_REQUIRE(G->W[l] && G->b[l], STDADK_E_ARG, "mlp_backward: layer %d grads NULL", l);
#define BWD(LN_, CPL_)                                                                                   \
  hipLaunchKernelGGL((ln_relu_bwd_kernel<LN_, CPL_>), dim3((unsigned)nblk), dim3(ROW_T), 0, st, dA,      \
                     ws + pl.xhat[l], ws + pl.rstd[l], LN_ ? P->ln_g[l] : (const float *)nullptr,        \
                     LN_ ? P->ln_b[l] : (const float *)nullptr, B, h, dZ, part, dp, drop_seed, l, mk)
    if (d->layernorm) { if (h <= 64) BWD(true, 1); else if (h <= 128) BWD(true, 2); else if (h <= 256) BWD(true, 4); else BWD(true, 16); }
    else { if (h <= 64) BWD(false, 1); else if (h <= 128) BWD(false, 2); else if (h <= 256) BWD(false, 4); else BWD(false, 16); }
#undef BWD
    STDADK_CHECK_LAUNCH("ln_relu_bwd");
    dim3 cg((unsigned)ceil_div(h, 256));
    if (d->layernorm) {
      hipLaunchKernelGGL(colsum_kernel, cg, dim3(256), 0, st, part, nblk, (int64_t)3 * h, h, G->ln_g[l]);
      hipLaunchKernelGGL(colsum_kernel, cg, dim3(256), 0, st, part + h, nblk, (int64_t)3 * h, h, G->ln_b[l]);
    }
    hipLaunchKernelGGL(colsum_kernel, cg, dim3(256), 0, st, part + 2 * h, nblk, (int64_t)3 * h, h, G->b[l]);
    STDADK_CHECK_LAUNCH("colsum");
    const float *ain = l > 0 ? ws + pl.act[l - 1] : features;
    const int64_t ldin = l > 0 ? d->hidden[l - 1] : ldf;
    const int kin = l > 0 ? d->hidden[l - 1] : d->in_dim;
    // dW[h][kin] = dZ^T ain   (reduction over the batch)
    rc = gemm_run(dZ, h, true, ain, ldin, true, h, kin, (int)B, nullptr, G->W[l], kin, slab, false, nullptr, st);
    if (rc) return rc;
    if (l > 0) {
      // dA_prev[B][kin] = dZ W   (W stored [h][kin] => K-major B operand)
      rc = gemm_run(dZ, h, false, P->W[l], kin, true, (int)B, kin, h, nullptr, dA, kin, slab, false, nullptr, st);
      if (rc) return rc;
    }
  }
  return 0;
}

extern "C" int stdadk_mse_f32(const float *y_pred, const float *y, int64_t n, float grad_scale,
                              float *dY, float *loss_sum, stdadk_stream_t stream) {
  STDADK_REQUIRE(n >= 0, STDADK_E_ARG, "mse: negative n");
  if (n == 0) return 0;
  STDADK_REQUIRE(y_pred && y, STDADK_E_ARG, "mse: NULL pointer");
  int64_t blocks = ceil_div(n, 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(mse_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y_pred, y, n,
                     grad_scale, dY, loss_sum);
  STDADK_CHECK_LAUNCH("mse");
  return 0;
}
