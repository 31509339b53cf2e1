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
#include "window.h"
#include "tail.h"
#include "optim.h"
#include "bin_body.h"
#include "knots.h"
#include "basis.h"

#include <stdlib.h>

namespace stdadk {

constexpr int ROW_T = 256;        // threads of the row kernels: 4 waves = 4 rows in flight
constexpr int MAX_CPL = 16;       // columns per lane => hidden width <= 1024

// rows per workgroup of the backward row kernels: 16 up to B = 8192, then <= 512 workgroups
static int bwd_rows(int64_t B) {
  if (B <= 8192) return 16;
  return (int)(ceil_div(ceil_div(B, 512), 4) * 4);
}

// ---------------------------------------------------------------------------------------------
// workspace plan (offsets in floats; every sub-buffer starts on a 256-byte boundary)
// ---------------------------------------------------------------------------------------------
enum PlanMode { PLAN_MLP = 0, PLAN_STEP_DENSE = 1, PLAN_STEP_WINDOW = 2 };

// rows from which the merged weight-gradient launch also does the reductions (run_backward, FinArgs): measured
// per step at 8 192 / 16 384 / 32 768 / 65 536 rows: +4 / 0 / -9 / -17 us against the reductions launch
constexpr int64_t FIN_FULL_MIN_ROWS = 49152;

struct Plan {
  int L;
  int64_t B;
  size_t xhat[STDADK_MAX_HIDDEN], rstd[STDADK_MAX_HIDDEN], act[STDADK_MAX_HIDDEN];
  size_t slab, slab_floats;
  size_t dA, dZ;            // [B][hmax] each
  size_t part, part_floats; // column-sum partials
  size_t dZl[STDADK_MAX_HIDDEN], partl[STDADK_MAX_HIDDEN];   // fused tail: per-layer dZ and partials
  // step-level buffers
  size_t feats; int64_t ldf;           // dense: materialised features
  size_t bw_exp, kpart; int kslabs;    // learnable knots: exp(log_bw) [Ks], per-slab knot partials
  size_t halo;                         // learnable knots, window path: per-level partial maxima of the candidate reach
  size_t kcs, kperm, kperm_tmp, reach; // scattered knots, window path: per-level cell lists of the knots (knot_bins)
  size_t psi; int ld_psi;              // window: temporal basis [B][ld_psi]
  size_t ypred, dY;                    // [B*Q]
  size_t keys, hist, cursor, cell_start, perm_tmp, perm, xs, ys, ts, y_s, X_s;
  int G;
  size_t fin_cnt, fin_slots; int fin_cap;   // window path: arrival counters / squared-norm slots of the merged dW launch
  size_t total_floats;
};

static void make_plan(const stdadk_mlp_desc *d, int64_t B, Plan *p, int mode = PLAN_MLP, int p_cov = 0,
                      int Kt = 0, int64_t Ks_learn = 0, int64_t Ks_scattered = 0, int n_levels = 0,
                      int64_t Ks_window = 0) {
  p->L = d->n_hidden;
  p->B = B;
  size_t off = 0;
  auto take = [&](size_t n) {
    size_t o = off;
    off += align_up(n > 0 ? n : 1, 64);
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
    if (!(l == 0 && mode == PLAN_STEP_WINDOW)) {
      s = gemm_slab_floats((int)B, h, prev); slab = s > slab ? s : slab;          // forward z
      s = gemm_slab_floats(h, prev, (int)B); slab = s > slab ? s : slab;          // dW  (M=h)
      s = gemm_slab_floats(prev, h, (int)B); slab = s > slab ? s : slab;          // dW^T (layer 0, W0 transposed)
    }
    if (l > 0 || Ks_learn > 0) { s = gemm_slab_floats((int)B, prev, h); slab = s > slab ? s : slab; }  // dA (layer 0: dFeat, learnable knots)
    prev = h;
  }
  {
    size_t s = gemm_slab_floats((int)B, d->out_dim, prev); slab = s > slab ? s : slab;
    s = gemm_slab_floats(d->out_dim, prev, (int)B); slab = s > slab ? s : slab;
    s = gemm_slab_floats((int)B, prev, d->out_dim); slab = s > slab ? s : slab;
  }
  if (mode == PLAN_STEP_WINDOW && d->n_hidden > 0) {
    size_t s = gemm_slab_floats(Kt, d->hidden[0], (int)B); slab = s > slab ? s : slab;
    if (p_cov > 0) { s = gemm_slab_floats(p_cov, d->hidden[0], (int)B); slab = s > slab ? s : slab; }
  }
  {
    // the grouped dW launch of the fused-tail backward keeps every job's slabs alive at once
    auto job_floats = [&](int M, int N) {
      int kps;
      int sp = gemm_pick_splits(M, N, (int)B, &kps, false);
      return (size_t)sp * M * N;
    };
    size_t grouped = 0;
    for (int l = 1; l < d->n_hidden; ++l) grouped += job_floats(d->hidden[l], d->hidden[l - 1]);
    if (mode == PLAN_STEP_WINDOW && d->n_hidden > 0) {
      grouped += job_floats(Kt, d->hidden[0]);
      if (p_cov > 0) grouped += job_floats(p_cov, d->hidden[0]);
    } else if (d->n_hidden > 0) {
      // materialising path, small D: dW0^T = features^T dZ_0 is a split-K product too and joins the group
      int kps;
      if (gemm_pick_splits(d->in_dim, d->hidden[0], (int)B, &kps, false) > 1) grouped += job_floats(d->in_dim, d->hidden[0]);
    }
    slab = grouped > slab ? grouped : slab;
  }
  p->slab_floats = slab;
  p->slab = take(slab);
  p->dA = take((size_t)B * hmax);
  p->dZ = take((size_t)B * hmax);
  int64_t nblk = ceil_div(B, bwd_rows(B));
  p->part_floats = (size_t)nblk * 3 * hmax;
  size_t head = (size_t)nblk * (size_t)d->out_dim * (hmax + 1);
  if (head > p->part_floats) p->part_floats = head;
  {
    const int64_t nb16 = ceil_div(B, TAIL_MIN_ROWS);
    size_t head16 = (size_t)nb16 * (size_t)d->out_dim * (hmax + 1);
    if (head16 > p->part_floats) p->part_floats = head16;
  }
  p->part = take(p->part_floats);
  for (int l = 0; l < d->n_hidden; ++l) {
    p->dZl[l] = take((size_t)B * d->hidden[l]);
    p->partl[l] = take((size_t)ceil_div(B, TAIL_MIN_ROWS) * 3 * d->hidden[l]);
  }
  p->feats = p->psi = p->ypred = p->dY = 0;
  p->ldf = 0; p->ld_psi = 0; p->G = 0;
  if (mode != PLAN_MLP) {
    p->ypred = take((size_t)B * d->out_dim);
    p->dY = take((size_t)B * d->out_dim);
  }
  if (mode == PLAN_STEP_DENSE) {
    p->ldf = (int64_t)align_up((size_t)d->in_dim, 32);
    p->feats = take((size_t)B * p->ldf);
  }
  p->bw_exp = p->kpart = p->halo = 0; p->kslabs = 0;
  if (mode != PLAN_MLP && Ks_learn > 0) {
    // dense: one partial per row slab; window: the wave that owns a knot produces its whole sum
    p->kslabs = mode == PLAN_STEP_DENSE ? knot_slabs(B) : 1;
    p->bw_exp = take((size_t)Ks_learn);
    p->kpart = take((size_t)p->kslabs * 3 * (size_t)Ks_learn);
    p->halo = take((size_t)STDADK_MAX_LEVELS * HALO_SPLIT);
  }
  p->kcs = p->kperm = p->kperm_tmp = p->reach = 0;
  if (mode == PLAN_STEP_WINDOW && Ks_scattered > 0) {
    p->kcs = take((size_t)n_levels * (KNOT_CELLS * KNOT_CELLS + 1));
    p->kperm = take((size_t)Ks_scattered);
    p->kperm_tmp = take((size_t)Ks_scattered);
    p->reach = take(STDADK_MAX_LEVELS);
    if (Ks_learn == 0) p->bw_exp = take(1);      // (fixed scattered knots: no exp table)
  }
  if (mode == PLAN_STEP_WINDOW) {
    p->G = pick_cell_grid(B);
    p->ld_psi = (int)align_up((size_t)(Kt > 0 ? Kt : 1), 4);
    p->psi = take((size_t)B * p->ld_psi);
    size_t nc = (size_t)p->G * p->G + 1;
    p->keys = take(B); p->hist = take(nc); p->cursor = take(nc); p->cell_start = take(nc);
    p->perm_tmp = take(B); p->perm = take(B);
    p->xs = take(B); p->ys = take(B); p->ts = take(B);
    p->y_s = take((size_t)B * d->out_dim);
    p->X_s = take((size_t)B * (p_cov > 0 ? p_cov : 1));
  }
  p->fin_cnt = p->fin_slots = 0; p->fin_cap = 0;
  if (mode == PLAN_STEP_WINDOW) {
    // output tiles + tall reduce workgroups + knot workgroups (at most one per knot + the padding of the
    // XCD-striped order), see FinArgs
    p->fin_cap = FIN_TILES_MAX + 1024 + (int)Ks_window + 64;
    p->fin_cnt = take(FIN_TILES_MAX);
    p->fin_slots = take((size_t)p->fin_cap);
  }
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
    uint64_t seed, const int *__restrict__ step_dev, int layer, const uint8_t *__restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (ROW_T / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  if (step_dev) seed += (uint64_t)step_dev[0] * 0x9E3779B97F4A7C15ULL;
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
    const float inv_h = 1.0f / (float)h;
    mean = wave_sum(sum) * inv_h;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      if (lane + 64 * j < h) { float d = z[j] - mean; sq += d * d; }
    float var = wave_sum(sq) * inv_h;
    rs = ln_rstd(var, eps);
    if (lane == 0) rstd_out[row] = rs;
  }
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const uint32_t rowkey = drop_rowkey(seed, layer, row), drop_thr = drop_threshold(drop_p);
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    int c = lane + 64 * j;
    if (c < h) {
      float xh = LN ? (z[j] - mean) * rs : z[j];
      float u = LN ? fmaf(xh, gamma[c], beta[c]) : xh;
      float a = fmaxf(u, 0.f);
      if (drop_p > 0.f) {
        int64_t e = row * h + c;
        bool keep = mask ? (mask[e] != 0) : drop_keep(rowkey, c, drop_thr);
        a = keep ? a * keep_scale : 0.f;
      }
      xhat[row * h + c] = xh;
      act[row * h + c] = a;
    }
  }
}

// Backward of Dropout -> ReLU -> LayerNorm for rows_per_wg rows per workgroup:
//   du = dA * keep/(1-p) * (u > 0);  dgamma += du*xhat;  dbeta += du
//   dz = rstd * (dxh - mean(dxh) - xhat*mean(dxh*xhat)),  dxh = du*gamma;   db += dz
// Column partials go to part[blk][3][h] (dgamma, dbeta, db) and are summed by colsum_kernel.
template <bool LN, int CPL>
__global__ __launch_bounds__(ROW_T) void ln_relu_bwd_kernel(
    const float *__restrict__ dA, const float *__restrict__ xhat, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, int64_t B, int h,
    float *__restrict__ dZ, float *__restrict__ part, float drop_p, uint64_t seed,
    const int *__restrict__ step_dev, int layer, const uint8_t *__restrict__ mask, int rows_per_wg) {
  __shared__ float red[3][ROW_T / 64][256];
  if (step_dev) seed += (uint64_t)step_dev[0] * 0x9E3779B97F4A7C15ULL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  float pg[CPL], pb[CPL], pz[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) pg[j] = pb[j] = pz[j] = 0.f;
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_wg;
  for (int rr = wave; rr < rows_per_wg; rr += ROW_T / 64) {
    const int64_t row = row0 + rr;
    if (row >= B) break;
    float du[CPL], xh[CPL];
    float s1 = 0.f, s2 = 0.f;
    const uint32_t rowkey = drop_rowkey(seed, layer, row), drop_thr = drop_threshold(drop_p);
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
          bool keep = mask ? (mask[e] != 0) : drop_keep(rowkey, c, drop_thr);
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
      const float inv_h = 1.0f / (float)h;
      m1 = wave_sum(s1) * inv_h;
      m2 = wave_sum(s2) * inv_h;
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

// Column sums of per-workgroup partials: out_q[c] = sum_blk part[blk*stride + q*seg + c], q < nseg.
// 64 columns x 4 block-quarters per workgroup, LDS reduction over the quarters.
struct ColsumOut { float *o[3]; };
constexpr int CS_G = 16;   // block-groups per workgroup
__global__ __launch_bounds__(64 * CS_G) void colsum_kernel(const float *__restrict__ part, int64_t nblk,
                                                           int64_t stride, int seg, int nseg,
                                                           ColsumOut out) {
  __shared__ float red[CS_G][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int n = seg * nseg;
  float s0 = 0.f, s1 = 0.f;
  if (c < n) {
    int64_t b = ty;
    for (; b + CS_G < nblk; b += 2 * CS_G) {
      s0 += part[b * stride + c];
      s1 += part[(b + CS_G) * stride + c];
    }
    for (; b < nblk; b += CS_G) s0 += part[b * stride + c];
  }
  red[ty][tx] = s0 + s1;
  __syncthreads();
  if (ty == 0 && c < n) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < CS_G; ++g) v += red[g][tx];
    int q = c / seg;
    out.o[q][c - q * seg] = v;
  }
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

// Backward of the small-Q output layer for rows_per_wg rows per workgroup:
//   dA[row][c] = sum_q dY[row][q] W[q][c];  partial dW[q][c] = sum_rows dY[row][q] a[row][c];
//   partial db[q] = sum_rows dY[row][q].   part[blk][q][h+1] (last column = db).
__global__ __launch_bounds__(ROW_T) void head_bwd_kernel(const float *__restrict__ a,
                                                         const float *__restrict__ dY, int64_t B,
                                                         int h, const float *__restrict__ W, int Q,
                                                         float *__restrict__ dA,
                                                         float *__restrict__ part, int rows_per_wg) {
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_wg;
  const int nrow = (int)min((int64_t)rows_per_wg, B - row0);
  extern __shared__ float sdy[];   // [rows_per_wg][Q]
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

// sum the head partials [blk][Q][h+1] and split them into dW[Q][h] and db[Q]
__global__ __launch_bounds__(64 * CS_G) void head_reduce_kernel(const float *__restrict__ part, int64_t nblk,
                                                                int Q, int h, float *__restrict__ dW,
                                                                float *__restrict__ db) {
  __shared__ float red[CS_G][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;
  const int n = Q * (h + 1);
  float s = 0.f;
  if (i < n)
    for (int64_t b = ty; b < nblk; b += CS_G) s += part[b * n + i];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < n) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < CS_G; ++g) v += red[g][tx];
    int q = i / (h + 1), c = i - q * (h + 1);
    if (c < h) dW[q * h + c] = v; else db[q] = v;
  }
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

// out[i][q] = in[perm[i]][q]
__global__ void gather_rows_kernel(const float *__restrict__ in, const int *__restrict__ perm, int B, int Q,
                                   float *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Q) return;
  int r = i / Q, q = i - r * Q;
  out[i] = in[(int64_t)perm[r] * Q + q];
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

struct Ctx;
struct Ctx {
  const stdadk_mlp_desc *d;
  const stdadk_mlp_tensors *P;
  const stdadk_mlp_tensors *G;
  float *ws;
  Plan pl;
  hipStream_t st;
  int64_t B;
  bool w0t;             // W[0] / dW[0] stored (in,out)
  float dp;             // effective dropout probability (0 in eval)
  uint64_t seed;
  const int *step_dev;
  const uint8_t *const *masks;
  // fused MSE (tail path): targets in the row order of the activations; outputs optional
  const float *mse_y = nullptr;
  float mse_scale = 0.f;
  float *mse_dY = nullptr;
  float *mse_loss = nullptr;
  bool mse_done = false;        // set by run_forward when the loss was fused into the tail kernel
  // fused training step: run_forward parks the tail launch here and run_backward issues it together
  // with the backward chain as ONE kernel (the tail of a row tile is row-local through the loss)
  bool fuse_tail = false, pend_valid = false;
  TailFwdArgs pend;
  // whole-step entry (stdadk_train_step_f32): squared-norm partials of the gradient ride along with the
  // reductions launch when every gradient is produced by it or by the per-knot gather
  float *gradsq = nullptr;      // [STDADK_GRADSQ_PARTS] or NULL
  int *step_inc = nullptr;
  bool gradsq_done = false, all_grouped = true;
  const float *gradsq_out = nullptr;   // where the partials went when gradsq_done (gradsq, or the slots of the merged
  int gradsq_n = 0;                    // dW launch) and how many
  bool merge_dw = false, dw_pend = false;   // window path: grouped dW products + per-knot gather as one launch
  int fin = 0;                  // ... 2: which also does the reductions (FinArgs; the tail launch cleared the counters);
                                // 1: which leaves the squared-norm slots of its knot rows, reductions launch behind it
  GemmGroup gg_pend;
  ReduceGroup rg_pend;
  bool l1_pend_valid = false;   // window path, B <= 4096: the layer-0 launch is parked as well
  L1FwdArgs l1_pend;
  int l1_basis = 0;
  // materialising path, small D: layer 0 starts inside the tail launch from the raw observations (TailDense0)
  bool d0 = false;
  const float *d0_coords = nullptr, *d0_t = nullptr, *d0_X = nullptr, *d0_bw = nullptr;
  bool scattered = false;       // STDADK_FLAG_SCATTERED: the knots of a level sit anywhere (window path through knot cell lists)
  bool bf16 = false;            // STDADK_FLAG_BF16: bf16 operands in the fused tail kernels (P->W_bf16 / WT_bf16)
  bool cap32 = false;           // this batch's tail launches use <= 32-row tiles (bf16 + dense layer 0 in the launch)
  bool save = true;             // false in eval mode: the forward keeps nothing for a backward (xhat, rstd, act, psi)
  bool log_bw = false;          // basis->s_bw holds log-bandwidths (learnable knots)
  const int64_t *idx = nullptr; // window path: the batch is rows idx[b] of the resident observation arrays
  bool prebinned = false;       // window path: stdadk_bin_batch_f32 already filled the workspace's bins
  LossDev loss = {STDADK_LOSS_MSE, 0, {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f}, 0.f, 1};   // y_cols 0 = Q
  const float *dz0 = nullptr;   // set by run_backward: dZ of layer 0
  // optional second stream: independent kernels of a step fork onto it (hipGraph-capturable
  // fork/join through events); NULL = everything on `st`
  hipStream_t aux = nullptr;
  int (*fork_after_dz)(Ctx &) = nullptr;   // called by run_backward (fused tail) right after the dZ kernel
  const stdadk_basis_desc *basis = nullptr;
  const stdadk_basis_desc *basis_in = nullptr;   // set by step_common (scattered_bins reads the log-bandwidths)
  bool dw0_forked = false;        // the per-knot gather of dW0^T was forked onto the auxiliary stream
  // extra products C[M][H0] = A^T dZ_0 (reduction over the batch) to run with the dW GEMMs of the
  // fused-tail backward: the temporal / covariate rows of dW0^T on the window path
  int n_extra = 0;
  struct { const float *A; int64_t lda; int M; float *C; } extra[2];
};

// fork/join between the main and the auxiliary stream (events are created once per thread)
static hipEvent_t *fork_events() {
  static thread_local hipEvent_t ev[4];
  static thread_local bool made = false;
  if (!made) {
    for (int i = 0; i < 4; ++i)
      if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    made = true;
  }
  return ev;
}
// `to` waits for everything enqueued on `from` so far
static int stream_depends(hipStream_t to, hipStream_t from, int slot) {
  if (g_dry_run) return 0;
  hipEvent_t *ev = fork_events();
  STDADK_REQUIRE(ev != nullptr, STDADK_E_ARG, "could not create fork/join events");
  hipError_t e = hipEventRecord(ev[slot], from);
  if (e == hipSuccess) e = hipStreamWaitEvent(to, ev[slot], 0);
  if (e != hipSuccess) { set_error("stream fork/join: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

static bool tail_enabled() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("STDADK_NO_FUSED_TAIL");
    v = (e && e[0] == '1') ? 0 : 1;
  }
  return v != 0;
}

static TailLayer tail_layer(const Ctx &c, int l) {
  TailLayer t;
  const stdadk_mlp_desc *d = c.d;
  t.W = c.P->W[l]; t.b = c.P->b[l];
  t.Wbf = c.bf16 ? c.P->W_bf16[l] : nullptr;
  t.WTbf = c.bf16 ? c.P->WT_bf16[l] : nullptr;
  t.g = d->layernorm ? c.P->ln_g[l] : nullptr;
  t.be = d->layernorm ? c.P->ln_b[l] : nullptr;
  t.h = d->hidden[l];
  t.hp = l > 0 ? d->hidden[l - 1] : d->in_dim;
  t.xhat = c.ws + c.pl.xhat[l]; t.rstd = c.ws + c.pl.rstd[l]; t.act = c.ws + c.pl.act[l];
  t.layer_id = l;
  return t;
}

// one hidden layer with the generic kernels: z = in W^T (+b) -> LN -> ReLU -> Dropout
static int generic_layer_forward(Ctx &c, int l, const float *in, int64_t ld_in, int K) {
  const stdadk_mlp_desc *d = c.d;
  const stdadk_mlp_tensors *P = c.P;
  float *ws = c.ws;
  const Plan &pl = c.pl;
  const int64_t B = c.B;
  hipStream_t st = c.st;
  const int h = d->hidden[l];
  STDADK_REQUIRE(P->W[l] && P->b[l], STDADK_E_ARG, "mlp_forward: layer %d weights NULL", l);
  STDADK_REQUIRE(!d->layernorm || (P->ln_g[l] && P->ln_b[l]), STDADK_E_ARG, "mlp_forward: layer %d LN NULL", l);
  float *xh = ws + pl.xhat[l], *act = ws + pl.act[l];
  int splits = 1, rc;
  // z partials: slabs when split, else straight into the xhat buffer (normalised in place below)
  if (l == 0 && c.w0t)
    rc = gemm_run(in, ld_in, false, P->W[l], h, true, (int)B, h, K, nullptr, xh, h, ws + pl.slab, true, &splits, st);
  else
    rc = gemm_run(in, ld_in, false, P->W[l], K, false, (int)B, h, K, nullptr, xh, h, ws + pl.slab, true, &splits, st);
  if (rc) return rc;
  const float *zsrc = splits > 1 ? ws + pl.slab : xh;
  const uint8_t *mk = (c.masks && c.dp > 0.f) ? c.masks[l] : nullptr;
  dim3 grid((unsigned)ceil_div(B, ROW_T / 64));
#define FWD(LN_, CPL_)                                                                                   \
  STDADK_LAUNCH((ln_relu_fwd_kernel<LN_, CPL_>), grid, dim3(ROW_T), 0, st, zsrc, splits,                 \
                (int64_t)B * h, P->b[l], LN_ ? P->ln_g[l] : (const float *)nullptr,                      \
                LN_ ? P->ln_b[l] : (const float *)nullptr, d->ln_eps, B, h, xh, ws + pl.rstd[l],         \
                act, c.dp, c.seed, c.step_dev, l, mk)
  if (d->layernorm) { if (h <= 64) FWD(true, 1); else if (h <= 128) FWD(true, 2); else if (h <= 256) FWD(true, 4); else FWD(true, 16); }
  else { if (h <= 64) FWD(false, 1); else if (h <= 128) FWD(false, 2); else if (h <= 256) FWD(false, 4); else FWD(false, 16); }
#undef FWD
  STDADK_CHECK_LAUNCH("ln_relu_fwd");
  return 0;
}

// hidden layers [l0, L) and the output layer; `in` = input of layer l0.  When the fused tail
// applies, every layer after the first, the output layer and (if c.mse_y is set) the MSE loss run
// in ONE kernel.
static int run_forward(Ctx &c, int l0, const float *in, int64_t ld_in, int K, float *y_pred) {
  const stdadk_mlp_desc *d = c.d;
  const stdadk_mlp_tensors *P = c.P;
  float *ws = c.ws;
  const Plan &pl = c.pl;
  const int64_t B = c.B;
  hipStream_t st = c.st;
  const int L = d->n_hidden, Q = d->out_dim;
  int rc;
  c.mse_done = false;
  STDADK_REQUIRE(P->W[L] && P->b[L], STDADK_E_ARG, "mlp_forward: output layer weights NULL");
  if (tail_enabled() && !c.masks && L >= 1 && tail_supported(d, 1)) {
    int l = l0;
    TailFwdArgs a;
    a.d0.on = 0;
    if (l == 0 && c.d0) {
      // layer 0 inside the tail launch: features of a row tile in LDS, W0^T rows as the B operand
      static const float cals[3] = {1.000000f, 0.223477f, 0.654714f};  // st_interp.py:56-60
      const stdadk_basis_desc *b = c.basis;
      TailDense0 &z = a.d0;
      z.on = 1;
      z.coords = c.d0_coords; z.t = c.d0_t; z.X = c.d0_X;
      z.p = b->p; z.Ks = (int)b->Ks; z.Kt = (int)b->Kt; z.basis = b->basis; z.cal = cals[b->basis];
      z.s_centers = b->s_centers; z.s_bw = c.d0_bw; z.t_centers = b->t_centers; z.t_bw = b->t_bw;
      z.feats = c.save ? ws + pl.feats : nullptr; z.ldf = pl.ldf;
      z.W0T = P->W[0];
      z.L0 = tail_layer(c, 0);
      if (!c.save) z.L0.xhat = z.L0.rstd = z.L0.act = nullptr;
      l = 1;
    } else if (l == 0) {
      rc = generic_layer_forward(c, 0, in, ld_in, K);
      if (rc) return rc;
      l = 1;
    }
    a.n_layers = L - l;
    for (int i = l; i < L; ++i) {
      STDADK_REQUIRE(P->W[i] && P->b[i] && (!d->layernorm || (P->ln_g[i] && P->ln_b[i])), STDADK_E_ARG,
                     "mlp_forward: layer %d parameters NULL", i);
      a.L[i - l] = tail_layer(c, i);
      if (!c.save) a.L[i - l].xhat = a.L[i - l].rstd = a.L[i - l].act = nullptr;
    }
    a.a_in = ws + pl.act[l - 1];
    a.h_in = d->hidden[l - 1];
    a.B = (int)B;
    a.Wo = P->W[L]; a.bo = P->b[L]; a.Q = Q;
    a.y_pred = y_pred;
    a.y = c.mse_y; a.grad_scale = c.mse_scale; a.dY = c.mse_dY; a.loss_sum = c.mse_loss;
    a.loss = c.loss;
    if (a.loss.y_cols == 0) a.loss.y_cols = Q;
    a.layernorm = d->layernorm; a.eps = d->ln_eps; a.drop_p = c.dp; a.seed = c.seed; a.step_dev = c.step_dev;
    a.bf16 = c.bf16 ? 1 : 0;
    c.cap32 = c.bf16 && a.d0.on != 0;
    c.mse_done = c.mse_y != nullptr;
    { const char *e = getenv("STDADK_TAIL_STAMPS"); a.stamps = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }
    if (c.fuse_tail && c.mse_done) { c.pend = a; c.pend_valid = true; return 0; }
    return tail_forward(a, st);
  }
  for (int l = l0; l < L; ++l) {
    rc = generic_layer_forward(c, l, in, ld_in, K);
    if (rc) return rc;
    in = ws + pl.act[l]; ld_in = d->hidden[l]; K = d->hidden[l];
  }
  if (Q <= HEAD_MAXQ && ld_in == K) {
    STDADK_LAUNCH(head_fwd_kernel, dim3((unsigned)ceil_div(B, ROW_T / 64)), dim3(ROW_T), 0, st, in, B,
                  K, P->W[L], P->b[L], Q, y_pred);
    STDADK_CHECK_LAUNCH("head_fwd");
  } else {
    STDADK_REQUIRE(!(L == 0 && c.w0t), STDADK_E_ARG, "mlp_forward: transposed W0 needs a hidden layer");
    rc = gemm_run(in, ld_in, false, P->W[L], K, false, (int)B, Q, K, P->b[L], y_pred, Q, ws + pl.slab, false, nullptr, st);
    if (rc) return rc;
  }
  return 0;
}

// Output layer backward, then hidden layers L-1 .. 0.  With `layer0_dense` false the first layer
// stops after dZ_0 and its bias / LayerNorm gradients (dW0 is produced by the window kernels).
static int run_backward(Ctx &c, const float *dY, const float *features, int64_t ldf, bool layer0_dense) {
  const stdadk_mlp_desc *d = c.d;
  const stdadk_mlp_tensors *P = c.P, *G = c.G;
  float *ws = c.ws;
  const Plan &pl = c.pl;
  const int64_t B = c.B;
  hipStream_t st = c.st;
  const int L = d->n_hidden, Q = d->out_dim;
  const int rows = bwd_rows(B);
  const int64_t nblk = ceil_div(B, rows);
  float *dA = ws + pl.dA, *dZ = ws + pl.dZ, *part = ws + pl.part, *slab = ws + pl.slab;
  int rc;
  c.dz0 = dZ;

  if (tail_enabled() && !c.masks && L >= 1 && tail_supported(d, 1)) {
    // one kernel for the whole activation-gradient path, then reductions and the dW GEMMs
    // (a split backward call cannot know whether its forward ran the dense layer 0 inside the launch: every
    //  bf16 backward on the materialising path with D <= TAIL_D0_MAX takes the 32-row cap the forward took)
    const bool cap32 = c.pend_valid ? c.cap32 : (c.bf16 && layer0_dense && c.w0t && d->in_dim <= TAIL_D0_MAX);
    const int64_t nb16 = ceil_div(B, tail_rows(B, cap32));      // workgroups of the fused backward = partial rows
    const int hL = d->hidden[L - 1];
    TailBwdArgs a;
    a.n_layers = L;
    for (int l = 0; l < L; ++l) {
      STDADK_REQUIRE(G->W[l] && G->b[l] && (!d->layernorm || (G->ln_g[l] && G->ln_b[l])), STDADK_E_ARG,
                     "mlp_backward: layer %d grads NULL", l);
      a.L[l] = tail_layer(c, l);
      a.dZ[l] = ws + pl.dZl[l];
      a.part[l] = ws + pl.partl[l];
    }
    STDADK_REQUIRE(G->W[L] && G->b[L], STDADK_E_ARG, "mlp_backward: output layer grads NULL");
    a.B = (int)B; a.Wo = P->W[L]; a.Q = Q; a.dY = dY;
    a.act_last = ws + pl.act[L - 1]; a.part_head = part;
    a.layernorm = d->layernorm; a.drop_p = c.dp; a.seed = c.seed; a.step_dev = c.step_dev;
    a.bf16 = c.bf16 ? 1 : 0;
    // merged weight-gradient launch: with the reductions inside it (FinArgs; this tail launch clears the arrival
    // counters) for large batches, where the K slices of a tile arrive spread out and the sums hide under the rest
    // of the launch; below that the last slices arrive together at the end of the launch and their sums would
    // extend it by more than the reductions launch costs (measured at 4 096 rows: 45 us against 27.6 + 7.7), so the
    // launch only leaves the squared norms of its knot rows.  Environment STDADK_DW_FIN=0|1|2 forces a mode
    // (0 = the round-2 path: reductions launch that also re-reads the knot rows for their norm).
    c.fin = 0;
    if (c.merge_dw && !layer0_dense && pl.fin_cap > 0) {
      int64_t tiles = 0;
      for (int l = 1; l < L; ++l) tiles += ceil_div(d->hidden[l], 64) * ceil_div(d->hidden[l - 1], 64);
      for (int e = 0; e < c.n_extra; ++e) tiles += ceil_div(c.extra[e].M, 64) * ceil_div(d->hidden[0], 64);
      const char *e = getenv("STDADK_DW_FIN");
      c.fin = e ? atoi(e) : (B >= FIN_FULL_MIN_ROWS ? 2 : 1);
      if (c.fin < 0 || c.fin > 2) c.fin = 1;
      if (c.fin == 2 && tiles > FIN_TILES_MAX) c.fin = 1;
    }
    a.zero_ints = c.fin == 2 ? reinterpret_cast<int *>(ws + pl.fin_cnt) : nullptr;
    a.n_zero = c.fin == 2 ? FIN_TILES_MAX : 0;
    { const char *e = getenv("STDADK_TAIL_BWD_STAMPS"); a.stamps = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }
    if (c.pend_valid) {
      c.pend_valid = false;
      if (c.l1_pend_valid) {
        c.l1_pend_valid = false;
        rc = l1_tail_launch(c.l1_pend, c.l1_basis, d->layernorm != 0, c.pend, a, st);
      } else {
        rc = tail_forward_backward(c.pend, a, st);
      }
    } else {
      rc = tail_backward(a, st, cap32);
    }
    if (rc) return rc;
    c.dz0 = ws + pl.dZl[0];
    if (c.fork_after_dz) {       // dZ of every layer is final: independent consumers may start now
      rc = c.fork_after_dz(c);
      if (rc) return rc;
    }
    // ---- ONE grouped launch for the dW products, ONE grouped launch for every fixed-order sum
    GemmGroup gg;
    ReduceGroup rg;
    size_t slab_off = 0;
    auto add_reduce = [&](const float *src, float *dst, int n, int splits, int64_t stride) -> bool {
      if (rg.n >= REDUCE_GROUP_MAX) return false;
      ReduceJob &j = rg.job[rg.n++];
      j.src = src; j.dst = dst; j.n = n; j.splits = splits; j.stride = stride;
      return true;
    };
    // C[M][N] (contiguous) = A^T Bm, both operands stored [B rows][.]; falls back to a stand-alone GEMM
    // (`bf`: a layer after the first under STDADK_FLAG_BF16 -- operands rounded to bf16 at the LDS boundary)
    auto add_tn = [&](const float *A, int64_t lda, const float *Bm, int64_t ldb, int M, int N, float *C,
                      bool bf = false) -> int {
      if (gg.n < GEMM_GROUP_MAX && rg.n < REDUCE_GROUP_MAX && gemm_tn_groupable(A, lda, Bm, ldb)) {
        GemmArgs &g = gg.job[gg.n++];
        g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = N; g.bias = nullptr;
        g.M = M; g.N = N; g.K = (int)B;
        g.bf16 = bf ? 1 : 0;
        g.splits = gemm_pick_splits(M, N, (int)B, &g.kps, false);
        g.slab = slab + slab_off; g.slab_stride = (int64_t)M * N;
        slab_off += (size_t)g.splits * M * N;
        if (c.fin != 2) add_reduce(g.slab, C, M * N, g.splits, g.slab_stride);   // 2: the last K slice to arrive sums
        return 0;
      }
      c.all_grouped = false;
      return gemm_run(A, lda, true, Bm, ldb, true, M, N, (int)B, nullptr, C, N, slab + slab_off, false, nullptr, st);
    };
    add_reduce(part, G->W[L], Q * hL, (int)nb16, (int64_t)Q * (hL + 1));
    add_reduce(part + Q * hL, G->b[L], Q, (int)nb16, (int64_t)Q * (hL + 1));
    for (int l = L - 1; l >= 0; --l) {
      const int h = d->hidden[l];
      const float *pp = ws + pl.partl[l];
      if (d->layernorm) {
        add_reduce(pp, G->ln_g[l], h, (int)nb16, (int64_t)3 * h);
        add_reduce(pp + h, G->ln_b[l], h, (int)nb16, (int64_t)3 * h);
      }
      add_reduce(pp + 2 * h, G->b[l], h, (int)nb16, (int64_t)3 * h);
      if (l == 0) break;
      // dW_l[h][kin] = dZ_l^T act_{l-1}
      rc = add_tn(ws + pl.dZl[l], h, ws + pl.act[l - 1], d->hidden[l - 1], h, d->hidden[l - 1], G->W[l], c.bf16);
      if (rc) return rc;
    }
    for (int e = 0; e < c.n_extra; ++e) {
      rc = add_tn(c.extra[e].A, c.extra[e].lda, c.dz0, d->hidden[0], c.extra[e].M, d->hidden[0], c.extra[e].C);
      if (rc) return rc;
    }
    c.n_extra = 0;     // consumed
    bool dw0_in_group = false;
    if (layer0_dense && c.w0t) {
      // materialising path with the (in,out) layout: while dW0^T = features^T dZ_0 is a split-K product
      // (few tiles: small D) it joins the grouped launch and its slabs the grouped reduction -- two launches
      // fewer than a stand-alone GEMM + slab sum; a D that fills the chip on its own keeps its direct GEMM
      int kps;
      const int h0 = d->hidden[0];
      const int sp = gemm_pick_splits(d->in_dim, h0, (int)B, &kps, false);
      if (sp > 1 && slab_off + (size_t)sp * d->in_dim * h0 <= pl.slab_floats) {
        rc = add_tn(features, ldf, c.dz0, h0, d->in_dim, h0, G->W[0]);
        if (rc) return rc;
        dw0_in_group = true;
      }
    }
    if (c.merge_dw && !layer0_dense) {
      // window path: the caller issues these products together with the per-knot gather of dW0^T as one
      // launch, then the reductions (step_backward)
      c.gg_pend = gg; c.rg_pend = rg; c.dw_pend = true;
      return 0;
    }
    rc = launch_gemm_tn_grouped(gg, st);
    if (rc) return rc;
    if (c.gradsq && c.all_grouped && layer0_dense && dw0_in_group && reduce_jobs_block_count(rg) > 0 &&
        reduce_jobs_block_count(rg) <= STDADK_GRADSQ_PARTS) {
      // one-call step: every gradient of the step is an output of this reductions launch, which then also
      // leaves the squared-norm partials for the clip (no region beside them)
      rg.sq_parts = c.gradsq; rg.step_inc = c.step_inc;
      c.gradsq_done = true; c.gradsq_out = c.gradsq; c.gradsq_n = reduce_jobs_block_count(rg);
    }
    rc = launch_reduce_jobs(rg, st);
    if (rc) return rc;
    if (layer0_dense && !dw0_in_group) {
      const int h = d->hidden[0];
      if (c.w0t)
        rc = gemm_run(features, ldf, true, c.dz0, h, true, d->in_dim, h, (int)B, nullptr, G->W[0], h, slab, false, nullptr, st);
      else
        rc = gemm_run(c.dz0, h, true, features, ldf, true, h, d->in_dim, (int)B, nullptr, G->W[0], d->in_dim, slab, false, nullptr, st);
      if (rc) return rc;
    }
    return 0;
  }

  const float *aL = L > 0 ? ws + pl.act[L - 1] : features;
  const int64_t ldaL = L > 0 ? d->hidden[L - 1] : ldf;
  const int hL = L > 0 ? d->hidden[L - 1] : d->in_dim;
  STDADK_REQUIRE(G->W[L] && G->b[L], STDADK_E_ARG, "mlp_backward: output layer grads NULL");
  if (Q <= HEAD_MAXQ && L > 0) {
    STDADK_LAUNCH(head_bwd_kernel, dim3((unsigned)nblk), dim3(ROW_T), sizeof(float) * rows * Q, st, aL, dY,
                       B, hL, P->W[L], Q, dA, part, rows);
    STDADK_CHECK_LAUNCH("head_bwd");
    int n = Q * (hL + 1);
    STDADK_LAUNCH(head_reduce_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(64 * CS_G), 0, st, part, nblk, Q, hL,
                       G->W[L], G->b[L]);
    STDADK_CHECK_LAUNCH("head_reduce");
  } else {
    // dW_out[Q][hL] = dY^T a ; db = colsum(dY) ; dA = dY W_out
    rc = gemm_run(dY, Q, true, aL, ldaL, true, Q, hL, (int)B, nullptr, G->W[L], hL, slab, false, nullptr, st);
    if (rc) return rc;
    ColsumOut co; co.o[0] = G->b[L]; co.o[1] = co.o[2] = nullptr;
    STDADK_LAUNCH(colsum_kernel, dim3((unsigned)ceil_div(Q, 64)), dim3(64 * CS_G), 0, st, dY, B, (int64_t)Q, Q, 1, co);
    STDADK_CHECK_LAUNCH("colsum");
    if (L > 0) {
      rc = gemm_run(dY, Q, false, P->W[L], hL, true, (int)B, hL, Q, nullptr, dA, hL, slab, false, nullptr, st);
      if (rc) return rc;
    }
  }

  for (int l = L - 1; l >= 0; --l) {
    const int h = d->hidden[l];
    const uint8_t *mk = (c.masks && c.dp > 0.f) ? c.masks[l] : nullptr;
    STDADK_REQUIRE(G->W[l] && G->b[l], STDADK_E_ARG, "mlp_backward: layer %d grads NULL", l);
#define BWD(LN_, CPL_)                                                                                   \
  STDADK_LAUNCH((ln_relu_bwd_kernel<LN_, CPL_>), dim3((unsigned)nblk), dim3(ROW_T), 0, st, dA,      \
                     ws + pl.xhat[l], ws + pl.rstd[l], LN_ ? P->ln_g[l] : (const float *)nullptr,        \
                     LN_ ? P->ln_b[l] : (const float *)nullptr, B, h, dZ, part, c.dp, c.seed,            \
                     c.step_dev, l, mk, rows)
    if (d->layernorm) { if (h <= 64) BWD(true, 1); else if (h <= 128) BWD(true, 2); else if (h <= 256) BWD(true, 4); else BWD(true, 16); }
    else { if (h <= 64) BWD(false, 1); else if (h <= 128) BWD(false, 2); else if (h <= 256) BWD(false, 4); else BWD(false, 16); }
#undef BWD
    STDADK_CHECK_LAUNCH("ln_relu_bwd");
    ColsumOut co;
    if (d->layernorm) {
      STDADK_REQUIRE(G->ln_g[l] && G->ln_b[l], STDADK_E_ARG, "mlp_backward: layer %d LN grads NULL", l);
      co.o[0] = G->ln_g[l]; co.o[1] = G->ln_b[l]; co.o[2] = G->b[l];
      STDADK_LAUNCH(colsum_kernel, dim3((unsigned)ceil_div(3 * h, 64)), dim3(64 * CS_G), 0, st, part, nblk,
                         (int64_t)3 * h, h, 3, co);
    } else {
      co.o[0] = G->b[l]; co.o[1] = co.o[2] = nullptr;
      STDADK_LAUNCH(colsum_kernel, dim3((unsigned)ceil_div(h, 64)), dim3(64 * CS_G), 0, st, part + 2 * h, nblk,
                         (int64_t)3 * h, h, 1, co);
    }
    STDADK_CHECK_LAUNCH("colsum");
    if (l == 0 && !layer0_dense) break;
    const float *ain = l > 0 ? ws + pl.act[l - 1] : features;
    const int64_t ldin = l > 0 ? d->hidden[l - 1] : ldf;
    const int kin = l > 0 ? d->hidden[l - 1] : d->in_dim;
    if (l == 0 && c.w0t)   // dW0^T[kin][h] = ain^T dZ
      rc = gemm_run(ain, ldin, true, dZ, h, true, kin, h, (int)B, nullptr, G->W[l], h, slab, false, nullptr, st);
    else                   // dW[h][kin] = dZ^T ain   (reduction over the batch)
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

static int launch_mse(const float *yp, const float *y, int64_t n, float scale, float *dY, float *loss_sum,
                      hipStream_t st) {
  int64_t blocks = ceil_div(n, 256);
  if (blocks > 1024) blocks = 1024;
  STDADK_LAUNCH(mse_kernel, dim3((unsigned)blocks), dim3(256), 0, st, yp, y, n, scale, dY, loss_sum);
  STDADK_CHECK_LAUNCH("mse");
  return 0;
}

static int check_basis(const stdadk_basis_desc *b, const stdadk_mlp_desc *d) {
  STDADK_REQUIRE(b != nullptr, STDADK_E_ARG, "basis desc is NULL");
  STDADK_REQUIRE(b->p >= 0 && b->Ks >= 0 && b->Kt >= 0 && b->basis >= 0 && b->basis <= 2, STDADK_E_ARG,
                 "basis desc: bad sizes / kind");
  STDADK_REQUIRE(b->p + b->Ks + b->Kt == d->in_dim, STDADK_E_SHAPE, "basis desc: p+Ks+Kt=%lld != in_dim=%d",
                 (long long)(b->p + b->Ks + b->Kt), d->in_dim);
  STDADK_REQUIRE(b->Ks == 0 || (b->s_centers && b->s_bw), STDADK_E_ARG, "basis desc: spatial knots NULL");
  STDADK_REQUIRE(b->Kt == 0 || (b->t_centers && b->t_bw), STDADK_E_ARG, "basis desc: temporal knots NULL");
  if (b->n_levels > 0) {
    STDADK_REQUIRE(b->n_levels <= STDADK_MAX_LEVELS, STDADK_E_ARG, "basis desc: too many levels");
    int64_t k = 0, kc = 0;
    for (int l = 0; l < b->n_levels; ++l) {
      STDADK_REQUIRE(b->side[l] >= 1, STDADK_E_ARG, "basis desc: side[%d] < 1", l);
      k += (int64_t)b->side[l] * b->side[l];
      kc += b->side[l];
    }
    // uniform grid: side lengths; STDADK_FLAG_SCATTERED: level sizes (the flag is checked by the callers)
    STDADK_REQUIRE(k == b->Ks || kc == b->Ks, STDADK_E_SHAPE,
                   "basis desc: neither sum(side^2)=%lld nor sum(side)=%lld equals Ks=%lld", (long long)k, (long long)kc,
                   (long long)b->Ks);
  }
  return 0;
}

static inline bool is_scattered(const stdadk_basis_desc *b, int flags) {
  return (flags & STDADK_FLAG_SCATTERED) != 0 && b->n_levels > 0;
}
static int check_levels(const stdadk_basis_desc *b, int flags) {
  if (b->n_levels <= 0) return 0;
  int64_t k = 0;
  for (int l = 0; l < b->n_levels; ++l) k += is_scattered(b, flags) ? (int64_t)b->side[l] : (int64_t)b->side[l] * b->side[l];
  STDADK_REQUIRE(k == b->Ks, STDADK_E_SHAPE, "basis desc: the level sizes add up to %lld, Ks = %lld%s", (long long)k,
                 (long long)b->Ks, is_scattered(b, flags) ? " (STDADK_FLAG_SCATTERED: side[] holds knot counts)" : "");
  return 0;
}

static inline int64_t learn_ks(const stdadk_basis_desc *b, int flags) {
  return (flags & STDADK_FLAG_LOG_BW) ? b->Ks : 0;
}

static bool want_window(const stdadk_basis_desc *b, const stdadk_mlp_desc *d, int flags) {
  if (flags & STDADK_FLAG_DENSE) return false;
  if (!(flags & STDADK_FLAG_W0_T)) return false;
  if (d->n_hidden < 1 || b->Ks <= 0 || b->Kt <= 0) return false;
  // small tables: materialising is cheap (D small) and the window path's per-knot gather is serial over
  // the many rows each coarse knot sees (MI355X, B = 4096, 227 knots: 0.34 ms window vs 0.15 ms dense)
  if (!(flags & STDADK_FLAG_WINDOW) && b->Ks < 1024) return false;
  if (is_scattered(b, flags) && b->n_levels > STDADK_MAX_LEVELS) return false;
  return l1_window_supported(b->n_levels, b->basis, d->hidden[0], b->p, (int)b->Kt);
}

static GridView make_grid(const stdadk_basis_desc *b, bool scattered = false) {
  static const float cals[3] = {1.000000f, 0.223477f, 0.654714f};   // st_interp.py:56-60
  GridView g;
  g.n_levels = b->n_levels;
  g.scattered = scattered ? 1 : 0;
  int off = 0;
  for (int l = 0; l < STDADK_MAX_LEVELS; ++l) {
    const int v = l < b->n_levels ? b->side[l] : 0;
    g.side[l] = scattered ? 0 : v;
    g.cnt[l] = scattered ? v : v * v;
    g.off[l] = off;
    off += g.cnt[l];
  }
  g.p = b->p; g.Ks = (int)b->Ks; g.Kt = (int)b->Kt;
  g.cal = cals[b->basis];
  g.centers = b->s_centers; g.bw = b->s_bw; g.t_centers = b->t_centers; g.t_bw = b->t_bw;
  return g;
}

static BinBuffers plan_bins(float *ws, const Plan &pl) {
  BinBuffers bb;
  bb.keys = (int *)(ws + pl.keys); bb.hist = (int *)(ws + pl.hist); bb.cursor = (int *)(ws + pl.cursor);
  bb.cell_start = (int *)(ws + pl.cell_start); bb.perm_tmp = (int *)(ws + pl.perm_tmp);
  bb.perm = (int *)(ws + pl.perm);
  bb.xs = ws + pl.xs; bb.ys = ws + pl.ys; bb.ts = ws + pl.ts; bb.y_s = ws + pl.y_s; bb.X_s = ws + pl.X_s;
  return bb;
}

// scattered knots: bin the knots of every level into cells (knot_bins) and point the forward at the lists
static int scattered_bins(Ctx &c, L1FwdArgs &a) {
  const Plan &pl = c.pl;
  const stdadk_basis_desc *b = c.basis_in;
  int *kcs = (int *)(c.ws + pl.kcs), *kperm = (int *)(c.ws + pl.kperm), *ktmp = (int *)(c.ws + pl.kperm_tmp);
  float *reach = c.ws + pl.reach;
  int rc;
  if (c.log_bw) {
    a.g.bw = c.ws + pl.bw_exp;
    rc = knot_bins(a.g, kcs, kperm, ktmp, reach, c.st, b->s_bw, c.ws + pl.bw_exp);
  } else {
    rc = knot_bins(a.g, kcs, kperm, ktmp, reach, c.st);
  }
  if (rc) return rc;
  a.kcs = kcs; a.kperm = kperm; a.reach = reach; a.Gk = KNOT_CELLS;
  return 0;
}

// feature build + layer 0 for the window path (sorted order); returns with act[0] ready
static int window_layer0_forward(Ctx &c, const stdadk_basis_desc *b, const float *coords, const float *t,
                                 const float *X, const float *y) {
  const Plan &pl = c.pl;
  BinBuffers bb = plan_bins(c.ws, pl);
  int rc = 0;
  if (!c.prebinned) {
    STDADK_REQUIRE(coords && t && (b->p == 0 || X), STDADK_E_ARG, "step: NULL observation pointer");
    rc = bin_obs(coords, t, y, c.loss.y_cols ? c.loss.y_cols : c.d->out_dim, X, b->p, (int)c.B, pl.G, bb, c.st,
                 c.idx);
    if (rc) return rc;
  }
  L1FwdArgs a;
  a.g = make_grid(b, c.scattered);
  a.halo = nullptr;
  if (c.scattered) {
    // knots anywhere: this step's cell lists of the knots (and exp(log_bw) for learnable ones)
    rc = scattered_bins(c, a);
    if (rc) return rc;
  } else if (c.log_bw) {
    // learnable knots: bandwidth = exp(log_bandwidth) (st_interp.py:146-148), and the candidate
    // windows follow wherever the knots are now
    a.g.bw = c.ws + pl.bw_exp;
    rc = knot_halo(a.g, c.ws + pl.halo, c.st, b->s_bw, c.ws + pl.bw_exp);   // also fills bw_exp
    if (rc) return rc;
    a.halo = c.ws + pl.halo;
  }
  a.xs = bb.xs; a.ys = bb.ys; a.ts = bb.ts; a.Xs = b->p > 0 ? bb.X_s : nullptr;
  a.B = (int)c.B; a.H = c.d->hidden[0];
  a.W0T = c.P->W[0]; a.b0 = c.P->b[0];
  a.gamma = c.d->layernorm ? c.P->ln_g[0] : nullptr;
  a.beta = c.d->layernorm ? c.P->ln_b[0] : nullptr;
  a.eps = c.d->ln_eps;
  a.xhat = c.ws + pl.xhat[0]; a.rstd = c.ws + pl.rstd[0]; a.act = c.ws + pl.act[0];
  a.psi = c.ws + pl.psi; a.ld_psi = pl.ld_psi;
  if (!c.save) a.xhat = a.rstd = a.psi = nullptr;      // act stays: it is the input of the next layer
  a.drop_p = c.dp; a.seed = c.seed; a.step_dev = c.step_dev;
  a.rows_per_wg = 0; a.n_wg = 0;
  STDADK_REQUIRE(a.W0T && a.b0, STDADK_E_ARG, "window forward: layer 0 weights NULL");
  // fused training step at <= 16 rows per CU: park this launch too; run_backward issues layer 0, the tail
  // forward, the loss and the tail backward as ONE kernel (every one of them is row-local)
  if (c.fuse_tail && c.mse_y && !c.masks && tail_enabled() && c.d->n_hidden >= 1 && tail_supported(c.d, 1) &&
      tail_rows(c.B) == 16 && l1_tail_supported(c.B, a.H) && getenv("STDADK_NO_L1_TAIL") == nullptr) {
    c.l1_pend = a; c.l1_pend_valid = true; c.l1_basis = b->basis;
    return 0;
  }
  return l1_window_forward(a, b->basis, c.d->layernorm != 0, c.st);
}

}  // namespace stdadk

using namespace stdadk;

extern "C" size_t stdadk_mlp_workspace_bytes(const stdadk_mlp_desc *desc, int64_t B) {
  // (a batch beyond the 32-bit row indices of the kernels is refused here too: the planner casts B to int, and the
  //  host-side sanitizer pass found the division by zero a truncated 2^40 ran into)
  if (check_desc(desc) != 0 || B < 0 || B >= (1ll << 31)) return 0;
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
  Ctx c;
  make_plan(d, B, &c.pl);
  STDADK_REQUIRE(workspace_bytes >= c.pl.total_floats * sizeof(float), STDADK_E_WORKSPACE,
                 "mlp_forward: workspace %zu < %zu bytes", workspace_bytes, c.pl.total_floats * sizeof(float));
  c.d = d; c.P = P; c.G = nullptr; c.ws = (float *)workspace; c.st = (hipStream_t)stream; c.B = B;
  c.w0t = false; c.dp = training ? d->dropout_p : 0.f; c.seed = drop_seed; c.step_dev = nullptr;
  c.masks = drop_mask;
  return run_forward(c, 0, features, ldf, d->in_dim, y_pred);
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
  Ctx c;
  make_plan(d, B, &c.pl);
  STDADK_REQUIRE(workspace_bytes >= c.pl.total_floats * sizeof(float), STDADK_E_WORKSPACE,
                 "mlp_backward: workspace too small");
  c.d = d; c.P = P; c.G = G; c.ws = (float *)workspace; c.st = (hipStream_t)stream; c.B = B;
  c.w0t = false; c.dp = d->dropout_p; c.seed = drop_seed; c.step_dev = nullptr; c.masks = drop_mask;
  return run_backward(c, dY, features, ldf, true);
}

extern "C" int stdadk_mse_f32(const float *y_pred, const float *y, int64_t n, float grad_scale,
                              float *dY, float *loss_sum, stdadk_stream_t stream) {
  STDADK_REQUIRE(n >= 0, STDADK_E_ARG, "mse: negative n");
  if (n == 0) return 0;
  STDADK_REQUIRE(y_pred && y, STDADK_E_ARG, "mse: NULL pointer");
  return launch_mse(y_pred, y, n, grad_scale, dY, loss_sum, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// step-level entry points: observations in, predictions / gradients out
// ---------------------------------------------------------------------------------------------
extern "C" int32_t stdadk_step_uses_window(const stdadk_basis_desc *b, const stdadk_mlp_desc *d, int32_t flags) {
  if (check_desc(d) != 0 || check_basis(b, d) != 0) return 0;
  return want_window(b, d, flags) ? 1 : 0;
}

extern "C" size_t stdadk_step_workspace_bytes(const stdadk_basis_desc *b, const stdadk_mlp_desc *d, int64_t B,
                                              int32_t flags) {
  if (check_desc(d) != 0 || check_basis(b, d) != 0 || B < 0 || B >= (1ll << 31)) return 0;
  Plan p;
  make_plan(d, B > 0 ? B : 1, &p, want_window(b, d, flags) ? PLAN_STEP_WINDOW : PLAN_STEP_DENSE, b->p, (int)b->Kt,
            learn_ks(b, flags), is_scattered(b, flags) ? b->Ks : 0, b->n_levels, b->Ks);
  return p.total_floats * sizeof(float);
}

static int step_common(Ctx &c, const stdadk_basis_desc *b, const stdadk_mlp_desc *d, int64_t B, void *workspace,
                       size_t workspace_bytes, int32_t flags, bool *window) {
  int rc = check_desc(d);
  if (rc) return rc;
  rc = check_basis(b, d);
  if (rc) return rc;
  STDADK_REQUIRE(B > 0 && B < (1ll << 31), STDADK_E_ARG, "step: bad B");
  STDADK_REQUIRE(workspace && aligned16(workspace), STDADK_E_ALIGN, "step: workspace NULL or not 16-byte aligned");
  *window = want_window(b, d, flags);
  rc = check_levels(b, flags);
  if (rc) return rc;
  c.scattered = *window && is_scattered(b, flags);
  make_plan(d, B, &c.pl, *window ? PLAN_STEP_WINDOW : PLAN_STEP_DENSE, b->p, (int)b->Kt, learn_ks(b, flags),
            c.scattered ? b->Ks : 0, b->n_levels, b->Ks);
  c.log_bw = (flags & STDADK_FLAG_LOG_BW) != 0;
  c.bf16 = (flags & STDADK_FLAG_BF16) != 0;
  STDADK_REQUIRE(!c.bf16 || (tail_enabled() && d->n_hidden >= 1 && tail_supported(d, 1)), STDADK_E_ARG,
                 "step: STDADK_FLAG_BF16 needs the fused tail kernels (hidden widths multiples of 16 up to %d, out_dim <= %d)",
                 TAIL_MAX_W, TAIL_MAXQ);
  STDADK_REQUIRE(workspace_bytes >= c.pl.total_floats * sizeof(float), STDADK_E_WORKSPACE,
                 "step: workspace %zu < %zu bytes", workspace_bytes, c.pl.total_floats * sizeof(float));
  c.d = d; c.ws = (float *)workspace; c.B = B;
  c.basis_in = b;
  c.w0t = (flags & STDADK_FLAG_W0_T) != 0;
  c.masks = nullptr;
  c.prebinned = *window && (flags & STDADK_FLAG_PREBINNED) != 0;
  STDADK_REQUIRE(*window || !(flags & STDADK_FLAG_PREBINNED), STDADK_E_ARG,
                 "step: STDADK_FLAG_PREBINNED needs the window path");
  return 0;
}

// dW0^T spatial rows (window path): every knot row by the wave that owns it
static int window_dw0(Ctx &c, hipStream_t st, GemmGroup *with_products = nullptr, const FinArgs *fin = nullptr,
                      ReduceGroup *tall = nullptr, int *n_slots = nullptr, int slot_cap = 0) {
  const stdadk_basis_desc *b = c.basis;
  float *ws = c.ws;
  L1BwdArgs a;
  a.g = make_grid(b, c.scattered);
  a.xs = ws + c.pl.xs; a.ys = ws + c.pl.ys;
  a.cell_start = (const int *)(ws + c.pl.cell_start);
  a.G = c.pl.G; a.B = (int)c.B; a.H = c.d->hidden[0];
  a.dZ = c.dz0; a.dW0T = c.G->W[0];
  a.W0T = nullptr; a.kpart = nullptr;
  if (c.log_bw) {      // learnable knots: the same pass also yields the raw knot gradients
    a.g.bw = ws + c.pl.bw_exp;
    a.W0T = c.P->W[0];
    a.kpart = ws + c.pl.kpart;
  }
  if (with_products && fin) {
    int need = dw_all_knot_blocks(a);
    if (fin->cnt) {
      STDADK_REQUIRE(tall && tall->n > 0, STDADK_E_ARG, "dw_all: finishing work without its reduce table");
      for (int j = 0; j < with_products->n; ++j)
        need += (int)(ceil_div(with_products->job[j].M, 64) * ceil_div(with_products->job[j].N, 64));
      need += reduce_jobs_block_count(*tall);
    }
    STDADK_REQUIRE(need <= slot_cap, STDADK_E_WORKSPACE, "dw_all: %d squared-norm slots, the plan holds %d", need,
                   slot_cap);
    return launch_dw_all(*with_products, a, b->basis, st, fin, tall, n_slots);
  }
  if (with_products) return launch_dw_all(*with_products, a, b->basis, st);
  return l1_window_backward(a, b->basis, st);
}

// forward of one batch; training != 0 keeps everything backward needs in the workspace
// (window path: `y`, when given, is carried into sorted order next to the observations, and the
//  predictions stay in sorted order in the plan's ypred buffer; y_pred NULL skips the un-permute)
static int step_forward(Ctx &c, const stdadk_basis_desc *b, bool window, const float *coords, const float *t,
                        const float *X, const float *y, float *y_pred, stdadk_stream_t stream) {
  const stdadk_mlp_desc *d = c.d;
  float *ws = c.ws;
  int rc;
  if (window) {
    rc = window_layer0_forward(c, b, coords, t, X, y);
    if (rc) return rc;
    rc = run_forward(c, 1, ws + c.pl.act[0], d->hidden[0], d->hidden[0], ws + c.pl.ypred);
    if (rc) return rc;
    if (!y_pred) return 0;
    return unpermute_rows(ws + c.pl.ypred, (const int *)(ws + c.pl.perm), (int)c.B, d->out_dim, y_pred, c.st);
  }
  const float *s_bw = b->s_bw;
  if (c.log_bw && b->Ks > 0) {        // bandwidth = exp(log_bandwidth)  (st_interp.py:146-148)
    rc = launch_exp(b->s_bw, b->Ks, ws + c.pl.bw_exp, c.st);
    if (rc) return rc;
    s_bw = ws + c.pl.bw_exp;
  }
  // small feature width, (in,out) first weight, fused tail available: features and layer 0 inside the tail
  // launch (environment STDADK_NO_DENSE0_TAIL=1: the separate kernels, diagnostic)
  c.d0 = c.w0t && d->in_dim <= TAIL_D0_MAX && d->n_hidden >= 1 && tail_enabled() && !c.masks && tail_supported(d, 1) &&
         c.pl.ldf == ((d->in_dim + 31) & ~31) && getenv("STDADK_NO_DENSE0_TAIL") == nullptr;
  if (c.d0) {
    c.basis = b;
    c.d0_coords = coords; c.d0_t = t; c.d0_X = X; c.d0_bw = s_bw;
    rc = run_forward(c, 0, nullptr, 0, d->in_dim, y_pred);
    c.d0 = false;
    return rc;
  }
  rc = stdadk_rbf_build_f32(coords, t, X, c.B, b->p, b->s_centers, s_bw, b->Ks, b->basis, b->t_centers,
                            b->t_bw, b->Kt, ws + c.pl.feats, c.pl.ldf, stream);
  if (rc) return rc;
  return run_forward(c, 0, ws + c.pl.feats, c.pl.ldf, d->in_dim, y_pred);
}

// backward of the batch whose training forward left its state in the workspace; dY in caller order
// (`dY_sorted`: window path only — dY is already the plan's dY buffer in sorted order)
static int step_backward(Ctx &c, const stdadk_basis_desc *b, bool window, const float *dY, bool dY_sorted) {
  const stdadk_mlp_desc *d = c.d;
  float *ws = c.ws;
  int rc;
  if (!window) return run_backward(c, dY, ws + c.pl.feats, c.pl.ldf, true);
  const int H = d->hidden[0], Q = d->out_dim;
  if (!dY_sorted) {   // dY rows into sorted order
    STDADK_LAUNCH(gather_rows_kernel, dim3((unsigned)ceil_div(c.B * Q, 256)), dim3(256), 0, c.st, dY,
                       (const int *)(ws + c.pl.perm), (int)c.B, Q, ws + c.pl.dY);
    STDADK_CHECK_LAUNCH("gather_rows");
  }
  // dW0^T rows of the temporal / covariate columns: small products that ride along with the dW
  // GEMMs of the fused-tail backward (or run on their own on the generic path)
  c.n_extra = 0;
  c.extra[c.n_extra].A = ws + c.pl.psi; c.extra[c.n_extra].lda = c.pl.ld_psi; c.extra[c.n_extra].M = (int)b->Kt;
  c.extra[c.n_extra].C = c.G->W[0] + (size_t)(b->p + b->Ks) * H;
  ++c.n_extra;
  if (b->p > 0) {
    c.extra[c.n_extra].A = ws + c.pl.X_s; c.extra[c.n_extra].lda = b->p; c.extra[c.n_extra].M = b->p;
    c.extra[c.n_extra].C = c.G->W[0];
    ++c.n_extra;
  }
  STDADK_REQUIRE(c.G->W[0], STDADK_E_ARG, "backward: dW[0] NULL");
  c.basis = b;
  c.dw0_forked = false;
  if (c.aux) {
    // as soon as dZ is final, the knot-row gather of dW0^T runs on the auxiliary stream beside the
    // dW GEMMs and reductions of the other layers
    c.fork_after_dz = [](Ctx &cc) -> int {
      int r = stream_depends(cc.aux, cc.st, 2);
      if (r) return r;
      cc.dw0_forked = true;
      return window_dw0(cc, cc.aux);
    };
  }
  c.merge_dw = !c.aux && getenv("STDADK_NO_DW_ALL") == nullptr;
  c.dw_pend = false;
  rc = run_backward(c, ws + c.pl.dY, nullptr, 0, false);
  c.fork_after_dz = nullptr;
  c.merge_dw = false;
  if (rc) return rc;
  if (c.dw_pend) {
    c.dw_pend = false;
    const bool sq = c.gradsq && c.all_grouped;
    const int fin_mode = c.fin;
    c.fin = 0;
    if (fin_mode == 2) {
      // products, per-knot gather, every fixed-order sum and the squared-norm partials: ONE launch
      FinArgs fin;
      fin.cnt = reinterpret_cast<int *>(ws + c.pl.fin_cnt);
      fin.slots = sq ? ws + c.pl.fin_slots : nullptr;
      fin.step_inc = sq ? c.step_inc : nullptr;
      int n_slots = 0;
      rc = window_dw0(c, c.st, &c.gg_pend, &fin, &c.rg_pend, &n_slots, c.pl.fin_cap);
      if (rc) return rc;
      if (sq) { c.gradsq_done = true; c.gradsq_out = fin.slots; c.gradsq_n = n_slots; }
      return 0;
    }
    const int nrb = reduce_jobs_block_count(c.rg_pend);
    if (fin_mode == 1 && sq && nrb > 0) {
      // the knot workgroups leave the squares of the rows they write (no second pass over dW0^T), the reductions
      // launch adds the partials of what it writes behind them
      FinArgs fin;
      fin.slots = ws + c.pl.fin_slots;
      int n_knot = 0;
      rc = window_dw0(c, c.st, &c.gg_pend, &fin, nullptr, &n_knot, c.pl.fin_cap - nrb);
      if (rc) return rc;
      c.rg_pend.sq_parts = fin.slots + n_knot;
      c.rg_pend.step_inc = c.step_inc;
      c.gradsq_done = true; c.gradsq_out = fin.slots; c.gradsq_n = n_knot + nrb;
      return launch_reduce_jobs(c.rg_pend, c.st);
    }
    rc = window_dw0(c, c.st, &c.gg_pend);
    if (rc) return rc;
    if (sq && nrb > 0 && nrb <= STDADK_GRADSQ_PARTS - 256) {
      // every gradient of the step is either an output of this launch or a spatial row of dW0^T (final
      // after the launch above): their squared norm comes out of the same launch
      c.rg_pend.sq_region_parts = c.gradsq;
      c.rg_pend.sq_parts = c.gradsq + 256;
      c.rg_pend.sq_src = c.G->W[0] + (size_t)b->p * H;
      c.rg_pend.sq_n = (int64_t)b->Ks * H;
      c.rg_pend.step_inc = c.step_inc;
      c.gradsq_done = true; c.gradsq_out = c.gradsq; c.gradsq_n = 256 + nrb;
    }
    return launch_reduce_jobs(c.rg_pend, c.st);
  }
  for (int e = 0; e < c.n_extra; ++e) {     // not consumed by a grouped launch
    rc = gemm_run(c.extra[e].A, c.extra[e].lda, true, c.dz0, H, true, c.extra[e].M, H, (int)c.B, nullptr,
                  c.extra[e].C, H, ws + c.pl.slab, false, nullptr, c.st);
    if (rc) return rc;
  }
  c.n_extra = 0;
  if (c.dw0_forked) return stream_depends(c.st, c.aux, 3);      // join
  return window_dw0(c, c.st);
}

extern "C" int stdadk_forward_f32(const stdadk_basis_desc *b, const stdadk_mlp_desc *d,
                                  const stdadk_mlp_tensors *P, const float *coords, const float *t,
                                  const float *X, int64_t B, float *y_pred, void *workspace,
                                  size_t workspace_bytes, int32_t training, uint64_t drop_seed,
                                  const int32_t *step_dev, int32_t flags, stdadk_stream_t stream) {
  if (B == 0) return 0;
  Ctx c;
  bool window;
  int rc = step_common(c, b, d, B, workspace, workspace_bytes, flags, &window);
  if (rc) return rc;
  STDADK_REQUIRE(P && coords && t && y_pred, STDADK_E_ARG, "forward: NULL pointer");
  STDADK_REQUIRE(b->p == 0 || X, STDADK_E_ARG, "forward: X is NULL with p=%d", b->p);
  c.P = P; c.G = nullptr; c.st = (hipStream_t)stream;
  c.dp = training ? d->dropout_p : 0.f; c.seed = drop_seed; c.step_dev = step_dev;
  c.save = training != 0;
  return step_forward(c, b, window, coords, t, X, nullptr, y_pred, stream);
}

// ---------------------------------------------------------------------------------------------------------
// Site x time prediction grids (A10): layer 0's pre-activation splits into a per-site and a per-time row
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void temporal_partial_kernel(const float *__restrict__ tv, int T, int Kt,
                                                               const float *__restrict__ tc, const float *__restrict__ tbw,
                                                               const float *__restrict__ Wt, int H, float *__restrict__ out) {
  // out[ti][c] = sum_j psi_j(t_ti) Wt[j][c]; one workgroup per time value
  const int ti = blockIdx.x;
  const float t = tv[ti];
  for (int c = threadIdx.x; c < H; c += 256) {
    float acc = 0.f;
    for (int j = 0; j < Kt; ++j) acc = fmaf(psi_eval(t, tc[j], tbw[j]), Wt[(size_t)j * H + c], acc);
    out[(size_t)ti * H + c] = acc;
  }
}

extern "C" int stdadk_spatial_partial_f32(const stdadk_basis_desc *b, const stdadk_mlp_desc *d,
                                          const stdadk_mlp_tensors *P, const float *coords, int64_t S, float *out,
                                          void *workspace, size_t workspace_bytes, int32_t flags,
                                          stdadk_stream_t stream) {
  if (S == 0) return 0;
  Ctx c;
  bool window;
  int rc = step_common(c, b, d, S, workspace, workspace_bytes, flags, &window);
  if (rc) return rc;
  STDADK_REQUIRE(window && !(flags & STDADK_FLAG_LOG_BW), STDADK_E_ARG,
                 "spatial_partial: needs the window path with fixed grid knots (compact-support basis, W0 stored (in,out))");
  STDADK_REQUIRE(b->p == 0, STDADK_E_ARG, "spatial_partial: covariates are per (site, time) row; p must be 0");
  STDADK_REQUIRE(P && P->W[0] && P->b[0] && coords && out, STDADK_E_ARG, "spatial_partial: NULL pointer");
  c.P = P; c.st = (hipStream_t)stream;
  const Plan &pl = c.pl;
  BinBuffers bb = plan_bins(c.ws, pl);
  // the sites' x coordinates stand in for the (unused) time column of the binning
  rc = bin_obs(coords, coords, nullptr, 0, nullptr, 0, (int)S, pl.G, bb, c.st);
  if (rc) return rc;
  L1FwdArgs a;
  a.g = make_grid(b, c.scattered);
  a.halo = nullptr;
  if (c.scattered) {
    rc = scattered_bins(c, a);
    if (rc) return rc;
  }
  a.xs = bb.xs; a.ys = bb.ys; a.ts = bb.ts; a.Xs = nullptr;
  a.B = (int)S; a.H = d->hidden[0];
  a.W0T = P->W[0]; a.b0 = P->b[0]; a.gamma = a.beta = nullptr; a.eps = d->ln_eps;
  a.xhat = a.rstd = a.psi = nullptr; a.ld_psi = pl.ld_psi;
  a.act = c.ws + pl.act[0];
  a.drop_p = 0.f; a.seed = 0; a.step_dev = nullptr;
  a.rows_per_wg = 0; a.n_wg = 0;
  a.raw = 1;
  rc = l1_window_forward(a, b->basis, false, c.st);
  if (rc) return rc;
  return unpermute_rows(c.ws + pl.act[0], (const int *)(c.ws + pl.perm), (int)S, d->hidden[0], out, c.st);
}

extern "C" int stdadk_temporal_partial_f32(const stdadk_basis_desc *b, const stdadk_mlp_desc *d,
                                           const stdadk_mlp_tensors *P, const float *t_values, int64_t T,
                                           float *out, int32_t flags, stdadk_stream_t stream) {
  if (T == 0) return 0;
  int rc = check_desc(d);
  if (rc) return rc;
  STDADK_REQUIRE(b && P && P->W[0] && t_values && out && T < (1ll << 31), STDADK_E_ARG, "temporal_partial: bad argument");
  STDADK_REQUIRE((flags & STDADK_FLAG_W0_T) && d->n_hidden >= 1, STDADK_E_ARG,
                 "temporal_partial: needs the first weight stored (in,out)");
  const int H = d->hidden[0];
  STDADK_LAUNCH(temporal_partial_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, t_values, (int)T,
                (int)b->Kt, b->t_centers, b->t_bw, P->W[0] + (size_t)(b->p + b->Ks) * H, H, out);
  STDADK_CHECK_LAUNCH("temporal_partial");
  return 0;
}

extern "C" int stdadk_forward_parts_f32(const stdadk_mlp_desc *d, const stdadk_mlp_tensors *P, const float *sp,
                                        int64_t S, const float *tp, int64_t T, float *y_pred,
                                        stdadk_stream_t stream) {
  if (S == 0 || T == 0) return 0;
  int rc = check_desc(d);
  if (rc) return rc;
  const int L = d->n_hidden, Q = d->out_dim;
  STDADK_REQUIRE(P && sp && tp && y_pred && S * T < (1ll << 31), STDADK_E_ARG, "forward_parts: bad argument");
  STDADK_REQUIRE(L >= 1 && tail_supported(d, 1) && P->W[L] && P->b[L], STDADK_E_ARG,
                 "forward_parts: hidden widths must be multiples of 16 up to %d, out_dim <= %d", TAIL_MAX_W, TAIL_MAXQ);
  auto layer = [&](int l) {
    TailLayer tl;
    tl.W = P->W[l]; tl.b = P->b[l];
    tl.Wbf = P->W_bf16[l]; tl.WTbf = P->WT_bf16[l];
    tl.g = d->layernorm ? P->ln_g[l] : nullptr; tl.be = d->layernorm ? P->ln_b[l] : nullptr;
    tl.h = d->hidden[l]; tl.hp = l > 0 ? d->hidden[l - 1] : d->in_dim;
    tl.xhat = tl.rstd = tl.act = nullptr;
    tl.layer_id = l;
    return tl;
  };
  TailFwdArgs a;
  a.n_layers = L - 1;
  for (int l = 1; l < L; ++l) {
    STDADK_REQUIRE(P->W[l] && P->b[l] && (!d->layernorm || (P->ln_g[l] && P->ln_b[l])), STDADK_E_ARG,
                   "forward_parts: layer %d parameters NULL", l);
    a.L[l - 1] = layer(l);
  }
  STDADK_REQUIRE(P->b[0] && (!d->layernorm || (P->ln_g[0] && P->ln_b[0])), STDADK_E_ARG, "forward_parts: layer 0 parameters NULL");
  a.d0.on = 2; a.d0.sp = sp; a.d0.tp = tp; a.d0.S = (int)S;
  a.d0.L0 = layer(0);
  a.a_in = nullptr; a.h_in = d->hidden[0];
  a.B = (int)(S * T);
  a.Wo = P->W[L]; a.bo = P->b[L]; a.Q = Q;
  a.y_pred = y_pred;
  a.y = nullptr; a.grad_scale = 0.f; a.dY = nullptr; a.loss_sum = nullptr;
  a.loss = LossDev{STDADK_LOSS_MSE, Q, {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f}, 0.f, 1};
  a.layernorm = d->layernorm; a.eps = d->ln_eps; a.drop_p = 0.f; a.seed = 0; a.step_dev = nullptr;
  a.stamps = nullptr;
  // bf16 operands whenever the caller supplies the copies of every layer after the first
  a.bf16 = L > 1 ? 1 : 0;
  for (int l = 1; l < L; ++l) if (!P->W_bf16[l]) a.bf16 = 0;
  return tail_forward(a, (hipStream_t)stream);
}

extern "C" int stdadk_backward_f32(const stdadk_basis_desc *b, const stdadk_mlp_desc *d,
                                   const stdadk_mlp_tensors *P, const stdadk_mlp_tensors *G, int64_t B,
                                   const float *dY, void *workspace, size_t workspace_bytes,
                                   uint64_t drop_seed, const int32_t *step_dev, int32_t flags,
                                   stdadk_stream_t stream) {
  if (B == 0) return 0;
  Ctx c;
  bool window;
  int rc = step_common(c, b, d, B, workspace, workspace_bytes, flags, &window);
  if (rc) return rc;
  STDADK_REQUIRE(P && G && dY, STDADK_E_ARG, "backward: NULL pointer");
  c.P = P; c.G = G; c.st = (hipStream_t)stream; c.dp = d->dropout_p; c.seed = drop_seed; c.step_dev = step_dev;
  return step_backward(c, b, window, dY, false);
}

extern "C" int stdadk_knot_backward_f32(const stdadk_basis_desc *b, const stdadk_mlp_desc *d,
                                        const stdadk_mlp_tensors *P, const float *coords, int64_t B,
                                        void *workspace, size_t workspace_bytes, int32_t flags,
                                        const stdadk_knot_train *kt, float *d_centers, float *d_log_bw,
                                        float *loss_sum, stdadk_stream_t stream) {
  if (B == 0) return 0;
  Ctx c;
  bool window;
  int rc = step_common(c, b, d, B, workspace, workspace_bytes, flags, &window);
  if (rc) return rc;
  STDADK_REQUIRE((flags & STDADK_FLAG_LOG_BW), STDADK_E_ARG,
                 "knot_backward: needs STDADK_FLAG_LOG_BW (the state of a learnable-knot forward/backward)");
  STDADK_REQUIRE(P && P->W[0] && (coords || window) && d_centers && d_log_bw, STDADK_E_ARG,
                 "knot_backward: NULL pointer");
  STDADK_REQUIRE(b->Ks > 0 && d->n_hidden >= 1, STDADK_E_ARG, "knot_backward: no spatial knots / no hidden layer");
  if (kt) {
    STDADK_REQUIRE(kt->centers_init || (!kt->gradient_damping && !(kt->movement_weight > 0.f)), STDADK_E_ARG,
                   "knot_backward: centers_init is NULL but damping / movement penalty is on");
    STDADK_REQUIRE(kt->domain_weight >= 0.f && kt->movement_weight >= 0.f, STDADK_E_ARG,
                   "knot_backward: negative penalty weight");
  }
  hipStream_t st = (hipStream_t)stream;
  float *ws = c.ws;
  const int H = d->hidden[0];
  KnotFinishArgs fa;
  fa.part = ws + c.pl.kpart; fa.slabs = c.pl.kslabs; fa.Ks = (int)b->Ks; fa.centers = b->s_centers;
  fa.centers_init = kt ? kt->centers_init : nullptr;
  fa.damping = kt ? kt->gradient_damping : 0;
  fa.thr = kt ? kt->damping_threshold : 0.f; fa.strength = kt ? kt->damping_strength : 0.f;
  fa.dom_w = kt ? kt->domain_weight : 0.f; fa.mov_w = kt ? kt->movement_weight : 0.f;
  fa.pen_grad_scale = kt ? kt->penalty_grad_scale : 0.f; fa.pen_loss_scale = kt ? kt->penalty_loss_scale : 0.f;
  fa.d_centers = d_centers; fa.d_log_bw = d_log_bw; fa.loss_sum = loss_sum;
  // window path: the per-knot gather of the backward already left the raw sums in the workspace
  if (window) return launch_knot_finish(fa, st);
  // where run_backward left dZ of layer 0 (fused tail: its per-layer buffer; generic: the shared one)
  const float *dz0 = (tail_enabled() && tail_supported(d, 1)) ? ws + c.pl.dZl[0] : ws + c.pl.dZ;
  // dFeat = dZ0 . W0 over ALL feature columns, into the (now free) feature buffer
  float *dfeat = ws + c.pl.feats;
  if (c.w0t)
    rc = gemm_run(dz0, H, false, P->W[0], H, false, (int)B, d->in_dim, H, nullptr, dfeat, c.pl.ldf,
                  ws + c.pl.slab, false, nullptr, st);
  else
    rc = gemm_run(dz0, H, false, P->W[0], d->in_dim, true, (int)B, d->in_dim, H, nullptr, dfeat, c.pl.ldf,
                  ws + c.pl.slab, false, nullptr, st);
  if (rc) return rc;
  KnotGradArgs ka;
  ka.coords = coords; ka.B = (int)B; ka.dFeat = dfeat; ka.ld = c.pl.ldf; ka.p = b->p;
  ka.centers = b->s_centers; ka.bw = ws + c.pl.bw_exp; ka.Ks = (int)b->Ks; ka.basis = b->basis;
  ka.part = ws + c.pl.kpart; ka.slabs = c.pl.kslabs;
  rc = launch_knot_grad(ka, st);
  if (rc) return rc;
  return launch_knot_finish(fa, st);
}

static int train_fwd_bwd_impl(const stdadk_basis_desc *b, const stdadk_mlp_desc *d,
                              const stdadk_mlp_tensors *P, const stdadk_mlp_tensors *G,
                              const float *coords, const float *t, const float *X,
                              const float *y, const int64_t *idx, int64_t B, float grad_scale,
                              const stdadk_loss_desc *loss, float *loss_sum,
                              float *y_pred, void *workspace, size_t workspace_bytes,
                              uint64_t drop_seed, const int32_t *step_dev, int32_t flags,
                              stdadk_stream_t stream, stdadk_stream_t aux_stream,
                              float *gradsq_parts = nullptr, int32_t *step_inc = nullptr,
                              const float **gradsq_out = nullptr, int *gradsq_n = nullptr) {
  if (B == 0) return 0;
  Ctx c;
  c.gradsq = gradsq_parts; c.step_inc = step_inc;
  struct Done {
    Ctx &c; const float **out; int *n;
    ~Done() { if (out) { *out = c.gradsq_done ? c.gradsq_out : nullptr; *n = c.gradsq_done ? c.gradsq_n : 0; } }
  } done_guard{c, gradsq_out, gradsq_n};
  bool window;
  int rc = step_common(c, b, d, B, workspace, workspace_bytes, flags, &window);
  if (rc) return rc;
  STDADK_REQUIRE(!idx || window, STDADK_E_ARG,
                 "train_fwd_bwd_indexed: only the window path gathers in place; use stdadk_gather_batch_f32 + "
                 "stdadk_train_fwd_bwd_f32 for the materialising path");
  c.idx = idx;
  STDADK_REQUIRE(P && G && (c.prebinned || (coords && t && y)), STDADK_E_ARG, "train_fwd_bwd: NULL pointer");
  c.aux = (aux_stream && aux_stream != stream) ? (hipStream_t)aux_stream : nullptr;
  STDADK_REQUIRE(b->p == 0 || X || c.prebinned, STDADK_E_ARG, "train_fwd_bwd: X is NULL with p=%d", b->p);
  c.P = P; c.G = G; c.st = (hipStream_t)stream; c.dp = d->dropout_p; c.seed = drop_seed; c.step_dev = step_dev;
  rc = make_loss(loss, d->out_dim, &c.loss);
  if (rc) return rc;
  const bool plain = loss_is_plain_mse(c.loss, d->out_dim);
  const int64_t n = B * d->out_dim;
  c.mse_scale = grad_scale; c.mse_dY = c.ws + c.pl.dY; c.mse_loss = loss_sum;
  // one launch for the forward and backward chains of the tail when nothing has to read the
  // predictions in between (window path: the un-permute of y_pred) and the loss is fused
  c.fuse_tail = getenv("STDADK_NO_TAIL_FWD_BWD") == nullptr && !(window && y_pred);
  if (window) {
    // everything between the binning and the weight gradients stays in sorted order
    c.mse_y = c.ws + c.pl.y_s;
    rc = step_forward(c, b, true, coords, t, X, y, y_pred, stream);
    if (rc) return rc;
    if (!c.mse_done) {
      rc = plain ? launch_mse(c.ws + c.pl.ypred, c.ws + c.pl.y_s, n, grad_scale, c.ws + c.pl.dY, loss_sum, c.st)
                 : launch_loss(c.loss, c.ws + c.pl.ypred, c.ws + c.pl.y_s, B, d->out_dim, grad_scale,
                               c.ws + c.pl.dY, loss_sum, c.st);
      if (rc) return rc;
    }
    rc = step_backward(c, b, true, c.ws + c.pl.dY, true);
    STDADK_REQUIRE(rc || (!c.pend_valid && !c.l1_pend_valid), STDADK_E_ARG,
                   "train_fwd_bwd: a parked launch was never issued");
    return rc;
  }
  float *yp = y_pred ? y_pred : c.ws + c.pl.ypred;
  c.mse_y = y;
  rc = step_forward(c, b, false, coords, t, X, nullptr, yp, stream);
  if (rc) return rc;
  if (!c.mse_done) {
    rc = plain ? launch_mse(yp, y, n, grad_scale, c.ws + c.pl.dY, loss_sum, c.st)
               : launch_loss(c.loss, yp, y, B, d->out_dim, grad_scale, c.ws + c.pl.dY, loss_sum, c.st);
    if (rc) return rc;
  }
  rc = step_backward(c, b, false, c.ws + c.pl.dY, false);
  STDADK_REQUIRE(rc || !c.pend_valid, STDADK_E_ARG, "train_fwd_bwd: the parked tail launch was never issued");
  return rc;
}

static int train_step_impl(const stdadk_basis_desc *b, const stdadk_mlp_desc *d,
                           const stdadk_mlp_tensors *P, const stdadk_mlp_tensors *G,
                           const float *coords, const float *t, const float *X, const float *y,
                           const int64_t *idx, int64_t B, float grad_scale,
                           const stdadk_loss_desc *loss, const stdadk_sparsity_desc *sparsity,
                           float *loss_sum, void *workspace, size_t workspace_bytes,
                           uint64_t drop_seed, int32_t flags, const stdadk_optim_desc *o,
                           stdadk_stream_t stream, const int64_t *next_idx, int64_t next_B, int32_t next_y_cols,
                           void *next_workspace, size_t next_workspace_bytes, int32_t *next_binned) {
  if (next_binned) *next_binned = 0;
  STDADK_REQUIRE(o && o->p && o->g && o->m && o->v && o->n > 0 && o->step_dev, STDADK_E_ARG,
                 "train_step: optimiser descriptor incomplete");
  STDADK_REQUIRE(o->max_norm <= 0.f || o->sumsq_parts, STDADK_E_ARG, "train_step: max_norm > 0 needs sumsq_parts");
  if (B == 0) return 0;
  const bool clip = o->max_norm > 0.f;
  const bool sparse = sparsity && sparsity->kind != STDADK_SPARSITY_NONE;
  const float *sq_parts = nullptr;   // where the step's own launches left the squared-norm partials, if they did
  int sq_n = 0;
  // with a sparsity penalty the gradient changes once more after the reductions: the norm is a pass of its own
  const bool fuse_sq = clip && !sparse;
  int rc = train_fwd_bwd_impl(b, d, P, G, coords, t, X, y, idx, B, grad_scale, loss, loss_sum, nullptr, workspace,
                              workspace_bytes, drop_seed, o->step_dev, flags, stream, nullptr,
                              fuse_sq ? o->sumsq_parts : nullptr, fuse_sq ? o->step_dev : nullptr, &sq_parts, &sq_n);
  if (rc) return rc;
  if (sparse) {
    const bool w0_t = (flags & STDADK_FLAG_W0_T) != 0;
    rc = stdadk_sparsity_f32(sparsity, P->W[0], G->W[0], w0_t ? d->hidden[0] : d->in_dim, w0_t, d->hidden[0],
                             b->p, (int32_t)b->Ks, (int32_t)b->Kt, 1.0f, (float)B * (float)d->out_dim, loss_sum,
                             nullptr, stream);
    if (rc) return rc;
  }
  int n_parts = 0;
  const float *parts = o->sumsq_parts;
  if (clip && sq_parts) {
    parts = sq_parts; n_parts = sq_n;              // partials (and the step advance) came out of the step's launches
  } else if (clip) {
    rc = stdadk_sumsq_f32(o->g, o->n, o->sumsq_parts, o->step_dev, stream);
    if (rc) return rc;
    n_parts = STDADK_SUMSQ_PARTS;
  } else {
    rc = stdadk_step_advance(o->step_dev, stream);
    if (rc) return rc;
  }
  if (next_idx && next_B > 0 && next_workspace && next_binned) {
    // the NEXT batch's binning inside this step's optimiser launch (optim.hip: adamw_bin_kernel), when it is the
    // one-launch binning of small batches; the caller then steps on `next_workspace` with STDADK_FLAG_PREBINNED
    Ctx cn;
    bool window_n = false;
    rc = step_common(cn, b, d, next_B, next_workspace, next_workspace_bytes, flags & ~STDADK_FLAG_PREBINNED, &window_n);
    if (rc) return rc;
    if (window_n && bin_small_eligible((int)next_B, cn.pl.G) && coords && t && (b->p == 0 || X)) {
      STDADK_REQUIRE(next_y_cols >= 0 && next_y_cols <= d->out_dim && (next_y_cols == 0 || y), STDADK_E_ARG,
                     "train_step: next_y_cols=%d must be in 0..Q with y given", next_y_cols);
      const BinBuffers bb = plan_bins(cn.ws, cn.pl);
      const BinSmallArgs ba = bin_small_args(coords, t, next_y_cols > 0 ? y : nullptr, next_y_cols, X, b->p, (int)next_B,
                                             cn.pl.G, bb, next_idx);
      rc = adamw_ema_with_binning(o->p, o->g, o->m, o->v, o->ema, o->n, o->lr, o->lr_dev, o->beta1, o->beta2, o->eps,
                                  o->weight_decay, o->step_dev, o->max_norm, clip ? parts : nullptr, n_parts,
                                  o->ema_decay, o->shadow, o->nonfinite_step ? loss_sum : nullptr, o->nonfinite_step,
                                  stream, ba);
      if (rc == 0) *next_binned = 1;
      return rc;
    }
  }
  return stdadk_adamw_ema_f32(o->p, o->g, o->m, o->v, o->ema, o->n, o->lr, o->lr_dev, o->beta1, o->beta2, o->eps,
                              o->weight_decay, 1, o->step_dev, o->max_norm, clip ? parts : nullptr, n_parts,
                              1.0f, o->ema_decay, o->shadow, o->nonfinite_step ? loss_sum : nullptr, o->nonfinite_step,
                              stream);
}

extern "C" int stdadk_train_step_f32(const stdadk_basis_desc *b, const stdadk_mlp_desc *d,
                                     const stdadk_mlp_tensors *P, const stdadk_mlp_tensors *G,
                                     const float *coords, const float *t, const float *X, const float *y,
                                     const int64_t *idx, int64_t B, float grad_scale,
                                     const stdadk_loss_desc *loss, const stdadk_sparsity_desc *sparsity,
                                     float *loss_sum, void *workspace, size_t workspace_bytes,
                                     uint64_t drop_seed, int32_t flags, const stdadk_optim_desc *o,
                                     stdadk_stream_t stream) {
  return train_step_impl(b, d, P, G, coords, t, X, y, idx, B, grad_scale, loss, sparsity, loss_sum, workspace,
                         workspace_bytes, drop_seed, flags, o, stream, nullptr, 0, 0, nullptr, 0, nullptr);
}

extern "C" int stdadk_train_step_next_f32(const stdadk_basis_desc *b, const stdadk_mlp_desc *d,
                                          const stdadk_mlp_tensors *P, const stdadk_mlp_tensors *G,
                                          const float *coords_all, const float *t_all, const float *X_all,
                                          const float *y_all, const int64_t *idx, int64_t B, float grad_scale,
                                          const stdadk_loss_desc *loss, const stdadk_sparsity_desc *sparsity,
                                          float *loss_sum, void *workspace, size_t workspace_bytes,
                                          uint64_t drop_seed, int32_t flags, const stdadk_optim_desc *o,
                                          const int64_t *next_idx, int64_t next_B, int32_t next_y_cols,
                                          void *next_workspace, size_t next_workspace_bytes, int32_t *next_binned,
                                          stdadk_stream_t stream) {
  STDADK_REQUIRE(next_binned, STDADK_E_ARG, "train_step_next: next_binned is NULL");
  return train_step_impl(b, d, P, G, coords_all, t_all, X_all, y_all, idx, B, grad_scale, loss, sparsity, loss_sum,
                         workspace, workspace_bytes, drop_seed, flags, o, stream, next_idx, next_B, next_y_cols,
                         next_workspace, next_workspace_bytes, next_binned);
}

extern "C" int stdadk_bin_batch_f32(const stdadk_basis_desc *b, const stdadk_mlp_desc *d,
                                    const float *coords_all, const float *t_all, const float *X_all,
                                    const float *y_all, const int64_t *idx, int64_t B, int32_t y_cols,
                                    void *workspace, size_t workspace_bytes, int32_t flags,
                                    stdadk_stream_t stream) {
  if (B == 0) return 0;
  Ctx c;
  bool window;
  int rc = step_common(c, b, d, B, workspace, workspace_bytes, flags & ~STDADK_FLAG_PREBINNED, &window);
  if (rc) return rc;
  STDADK_REQUIRE(window, STDADK_E_ARG, "bin_batch: only the window path bins its batches");
  STDADK_REQUIRE(coords_all && t_all && (b->p == 0 || X_all), STDADK_E_ARG, "bin_batch: NULL pointer");
  STDADK_REQUIRE(y_cols >= 0 && y_cols <= d->out_dim && (y_cols == 0 || y_all), STDADK_E_ARG,
                 "bin_batch: y_cols=%d must be in 0..Q with y_all given", y_cols);
  BinBuffers bb = plan_bins(c.ws, c.pl);
  // beside a running step the binning goes as several launches of small workgroups, which share CUs
  // with the step's one-workgroup-per-CU kernels; a single 1024-thread workgroup would take a whole CU
  // away from them for its entire duration
  return bin_obs(coords_all, t_all, y_cols > 0 ? y_all : nullptr, y_cols, X_all, b->p, (int)B, c.pl.G, bb,
                 (hipStream_t)stream, idx, true);
}

extern "C" int stdadk_train_fwd_bwd_f32(const stdadk_basis_desc *b, const stdadk_mlp_desc *d,
                                        const stdadk_mlp_tensors *P, const stdadk_mlp_tensors *G,
                                        const float *coords, const float *t, const float *X,
                                        const float *y, int64_t B, float grad_scale,
                                        const stdadk_loss_desc *loss, float *loss_sum,
                                        float *y_pred, void *workspace, size_t workspace_bytes,
                                        uint64_t drop_seed, const int32_t *step_dev, int32_t flags,
                                        stdadk_stream_t stream, stdadk_stream_t aux_stream) {
  return train_fwd_bwd_impl(b, d, P, G, coords, t, X, y, nullptr, B, grad_scale, loss, loss_sum, y_pred, workspace,
                            workspace_bytes, drop_seed, step_dev, flags, stream, aux_stream);
}

extern "C" int stdadk_train_fwd_bwd_indexed_f32(const stdadk_basis_desc *b, const stdadk_mlp_desc *d,
                                                const stdadk_mlp_tensors *P, const stdadk_mlp_tensors *G,
                                                const float *coords_all, const float *t_all,
                                                const float *X_all, const float *y_all, const int64_t *idx,
                                                int64_t B, float grad_scale, const stdadk_loss_desc *loss,
                                                float *loss_sum, float *y_pred, void *workspace,
                                                size_t workspace_bytes, uint64_t drop_seed,
                                                const int32_t *step_dev, int32_t flags,
                                                stdadk_stream_t stream, stdadk_stream_t aux_stream) {
  STDADK_REQUIRE(idx || B == 0, STDADK_E_ARG, "train_fwd_bwd_indexed: idx is NULL");
  return train_fwd_bwd_impl(b, d, P, G, coords_all, t_all, X_all, y_all, idx, B, grad_scale, loss, loss_sum, y_pred,
                            workspace, workspace_bytes, drop_seed, step_dev, flags, stream, aux_stream);
}
