// Fused MLP tail (see tail.h).  fp32 on the matrix cores with v_mfma_f32_16x16x4_f32: lane l holds
// A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15]; D[row = 4*(l>>4) + reg][col = l&15].
// A lane group q = l>>4 reads 4 consecutive k with one ds_read_b128 and feeds them to 4 MFMAs
// (k = 16j + 4q + e for MFMA e), for both operands, so the k order inside a 16-deep group is free.
//
// Replaces, for the layers after the first: nn.Linear / nn.LayerNorm / nn.ReLU / nn.Dropout forward
// (stnf/models/st_interp.py:656-690,880), nn.MSELoss (scripts/train_st_interp.py:549,621) and the
// activation-gradient half of loss.backward() (:693).
#include "tail.h"

namespace stdadk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TT = 256;                       // threads per workgroup (4 waves)
constexpr int R = TAIL_ROWS;                  // 16 rows
constexpr int ACT_LD = TAIL_MAX_W + 4;        // activation row stride in LDS (floats)
constexpr int WF_LD = 36;                     // forward W chunk: [n][32 k] row stride
constexpr int WK_LD = TAIL_MAX_W + 16;        // backward W chunk: [32 k][n] row stride

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TT) void tail_fwd_kernel(TailFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *act0 = smem, *act1 = smem + R * ACT_LD;
  float *wb = smem + 2 * R * ACT_LD;                 // [2][TAIL_MAX_W * WF_LD]
  __shared__ float red[TT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, c16 = lane & 15;
  const int row0 = blockIdx.x * R;
  for (int i = tid; i < 2 * R * ACT_LD; i += TT) smem[i] = 0.f;   // pads must be finite (x0 later)
  __syncthreads();
  {
    const int v4 = a.h_in >> 2;
    for (int idx = tid; idx < R * v4; idx += TT) {
      const int row = idx / v4, c4 = idx - row * v4;
      if (row0 + row < a.B)
        *reinterpret_cast<float4 *>(act0 + row * ACT_LD + 4 * c4) =
            *reinterpret_cast<const float4 *>(a.a_in + (size_t)(row0 + row) * a.h_in + 4 * c4);
    }
  }
  __syncthreads();
  const uint64_t seed = a.seed + (a.step_dev ? (uint64_t)a.step_dev[0] * 0x9E3779B97F4A7C15ULL : 0ULL);
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  float *cur = act0, *nxt = act1;

  for (int li = 0; li < a.n_layers; ++li) {
    const TailLayer &L = a.L[li];
    const int h = L.h, hp = L.hp;
    const int NT = h >> 4, nchunk = (hp + 31) >> 5;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 wreg[8];
    auto load_chunk = [&](int c) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = tid + TT * i;
        const int n = idx >> 3, k = 32 * c + 4 * (idx & 7);
        wreg[i] = (n < h && k < hp) ? *reinterpret_cast<const float4 *>(L.W + (size_t)n * hp + k)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    load_chunk(0);
    for (int c = 0; c < nchunk; ++c) {
      float *wbuf = wb + (c & 1) * (TAIL_MAX_W * WF_LD);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = tid + TT * i;
        *reinterpret_cast<float4 *>(wbuf + (idx >> 3) * WF_LD + 4 * (idx & 7)) = wreg[i];
      }
      __syncthreads();
      if (c + 1 < nchunk) load_chunk(c + 1);      // next chunk's loads fly under the MFMAs
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float4 av = *reinterpret_cast<const float4 *>(cur + c16 * ACT_LD + 32 * c + 16 * j + 4 * q);
        const float af[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int t = wave + 4 * i;
          if (t < NT) {
            const float4 bv = *reinterpret_cast<const float4 *>(wbuf + (16 * t + c16) * WF_LD + 16 * j + 4 * q);
            const float bf[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i] = mfma16(af[e], bf[e], acc[i]);
          }
        }
      }
    }
    // z = acc + bias into the other activation buffer
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = wave + 4 * i;
      if (t < NT) {
        const int col = 16 * t + c16;
        const float bv = L.b[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) nxt[(4 * q + r) * ACT_LD + col] = acc[i][r] + bv;
      }
    }
    __syncthreads();
    // LayerNorm -> ReLU -> Dropout, wave w owns rows 4w .. 4w+3
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = 4 * wave + rr;
      const int grow = row0 + row;
      float z[4];
      float s = 0.f;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int col = lane + 64 * cc;
        z[cc] = col < h ? nxt[row * ACT_LD + col] : 0.f;
        s += z[cc];
      }
      float mean = 0.f, rs = 1.f;
      if (a.layernorm) {
        mean = wave_sum(s) / (float)h;
        float sq = 0.f;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          if (lane + 64 * cc < h) { float d = z[cc] - mean; sq += d * d; }
        rs = 1.0f / sqrtf(wave_sum(sq) / (float)h + a.eps);
        if (lane == 0 && grow < a.B) L.rstd[grow] = rs;
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int col = lane + 64 * cc;
        if (col < h) {
          const float xh = a.layernorm ? (z[cc] - mean) * rs : z[cc];
          const float u = a.layernorm ? fmaf(xh, L.g[col], L.be[col]) : xh;
          float v = fmaxf(u, 0.f);
          if (a.drop_p > 0.f) {
            const bool keep = drop_keep(seed, L.layer_id, (int64_t)grow * h + col, a.drop_p);
            v = keep ? v * keep_scale : 0.f;
          }
          nxt[row * ACT_LD + col] = v;
          if (grow < a.B) {
            L.xhat[(size_t)grow * h + col] = xh;
            L.act[(size_t)grow * h + col] = v;
          }
        }
      }
    }
    __syncthreads();
    float *tmp = cur; cur = nxt; nxt = tmp;
  }

  // output layer (+ MSE): y[row][qq] = a_last[row,:] . Wo[qq,:] + bo[qq]
  const int hl = a.n_layers ? a.L[a.n_layers - 1].h : a.h_in;
  float lsum = 0.f;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int row = 4 * wave + rr;
    const int grow = row0 + row;
    for (int qq = 0; qq < a.Q; ++qq) {
      float s = 0.f;
      for (int col = lane; col < hl; col += 64) s = fmaf(cur[row * ACT_LD + col], a.Wo[qq * hl + col], s);
      s = wave_sum(s);
      if (lane == 0 && grow < a.B) {
        const float yv = s + a.bo[qq];
        a.y_pred[(size_t)grow * a.Q + qq] = yv;
        if (a.y) {
          const float d = yv - a.y[(size_t)grow * a.Q + qq];
          lsum = fmaf(d, d, lsum);
          if (a.dY) a.dY[(size_t)grow * a.Q + qq] = 2.0f * d * a.grad_scale;
        }
      }
    }
  }
  if (a.y && a.loss_sum) {
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (tid == 0) atomicAdd(a.loss_sum, (red[0] + red[1]) + (red[2] + red[3]));
  }
}

// ---------------------------------------------------------------------------------------------
// backward (data path): dA / dZ of every hidden layer + column partials for dgamma, dbeta, db
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TT) void tail_bwd_kernel(TailBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *d0 = smem, *d1 = smem + R * ACT_LD;
  float *wb = smem + 2 * R * ACT_LD;                 // [2][32 * WK_LD]
  float *red = wb + 2 * 32 * WK_LD;                  // [3][4][256]
  float *sdy = red + 3 * 4 * 256;                    // [R][TAIL_MAXQ]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, c16 = lane & 15;
  const int row0 = blockIdx.x * R;
  for (int i = tid; i < 2 * R * ACT_LD; i += TT) smem[i] = 0.f;
  if (tid < R * TAIL_MAXQ) {
    const int row = tid / a.Q, qq = tid - row * a.Q;
    sdy[tid] = (tid < R * a.Q && row0 + row < a.B) ? a.dY[(size_t)(row0 + row) * a.Q + qq] : 0.f;
  }
  __syncthreads();
  const int hl = a.L[a.n_layers - 1].h;
  for (int idx = tid; idx < R * hl; idx += TT) {
    const int row = idx / hl, col = idx - row * hl;
    float d = 0.f;
    for (int qq = 0; qq < a.Q; ++qq) d = fmaf(sdy[row * a.Q + qq], a.Wo[qq * hl + col], d);
    d0[row * ACT_LD + col] = d;
  }
  // output-layer weight gradient partials of this tile: part_head[blk][qq][hl+1] (last column = db)
  {
    float *ph = a.part_head + (size_t)blockIdx.x * a.Q * (hl + 1);
    const int nrow = min(R, a.B - row0);
    for (int col = tid; col < hl; col += TT) {
      float pw[TAIL_MAXQ];
#pragma unroll
      for (int qq = 0; qq < TAIL_MAXQ; ++qq) pw[qq] = 0.f;
      for (int row = 0; row < nrow; ++row) {
        const float av = a.act_last[(size_t)(row0 + row) * hl + col];
#pragma unroll
        for (int qq = 0; qq < TAIL_MAXQ; ++qq)
          if (qq < a.Q) pw[qq] = fmaf(sdy[row * a.Q + qq], av, pw[qq]);
      }
#pragma unroll
      for (int qq = 0; qq < TAIL_MAXQ; ++qq)
        if (qq < a.Q) ph[qq * (hl + 1) + col] = pw[qq];
    }
    if (tid < a.Q) {
      float sb = 0.f;
      for (int row = 0; row < nrow; ++row) sb += sdy[row * a.Q + tid];
      ph[tid * (hl + 1) + hl] = sb;
    }
  }
  __syncthreads();
  const uint64_t seed = a.seed + (a.step_dev ? (uint64_t)a.step_dev[0] * 0x9E3779B97F4A7C15ULL : 0ULL);
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  float *cur = d0, *nxt = d1;

  for (int li = a.n_layers - 1; li >= 0; --li) {
    const TailLayer &L = a.L[li];
    const int h = L.h;
    // ---- (a) Dropout -> ReLU -> LayerNorm backward, rows 4w .. 4w+3 of this wave
    float pg[4], pb[4], pz[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) pg[cc] = pb[cc] = pz[cc] = 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = 4 * wave + rr;
      const int grow = row0 + row;
      const bool valid = grow < a.B;
      float xh[4], dxh[4];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int col = lane + 64 * cc;
        xh[cc] = 0.f; dxh[cc] = 0.f;
        if (col < h && valid) {
          const float x = L.xhat[(size_t)grow * h + col];
          const float u = a.layernorm ? fmaf(x, L.g[col], L.be[col]) : x;
          float d = cur[row * ACT_LD + col];
          if (a.drop_p > 0.f) {
            const bool keep = drop_keep(seed, L.layer_id, (int64_t)grow * h + col, a.drop_p);
            d = keep ? d * keep_scale : 0.f;
          }
          d = u > 0.f ? d : 0.f;
          xh[cc] = x;
          if (a.layernorm) {
            pg[cc] += d * x;
            pb[cc] += d;
            d *= L.g[col];
            s1 += d;
            s2 += d * x;
          }
          dxh[cc] = d;
        }
      }
      float rs = 1.f, m1 = 0.f, m2 = 0.f;
      if (a.layernorm) {
        m1 = wave_sum(s1) / (float)h;
        m2 = wave_sum(s2) / (float)h;
        rs = valid ? L.rstd[grow] : 0.f;
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int col = lane + 64 * cc;
        if (col < h) {
          const float dz = (a.layernorm && valid) ? rs * (dxh[cc] - m1 - xh[cc] * m2) : dxh[cc];
          cur[row * ACT_LD + col] = dz;
          if (valid) a.dZ[li][(size_t)grow * h + col] = dz;
          pz[cc] += dz;
        }
      }
    }
    // column partials of this workgroup's 16 rows
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      red[(0 * 4 + wave) * 256 + lane + 64 * cc] = pg[cc];
      red[(1 * 4 + wave) * 256 + lane + 64 * cc] = pb[cc];
      red[(2 * 4 + wave) * 256 + lane + 64 * cc] = pz[cc];
    }
    __syncthreads();
    if (tid < h) {
      float *pbase = a.part[li] + (size_t)blockIdx.x * 3 * h;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        pbase[k * h + tid] = (red[(k * 4 + 0) * 256 + tid] + red[(k * 4 + 1) * 256 + tid]) +
                             (red[(k * 4 + 2) * 256 + tid] + red[(k * 4 + 3) * 256 + tid]);
    }
    if (li == 0) break;
    // ---- (b) dA_prev[16 x hp] = dZ[16 x h] W[h x hp]
    const int hp = L.hp;
    const int NT = hp >> 4, nchunk = (h + 31) >> 5, v4 = hp >> 2;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 wreg[8];
    auto load_chunk = [&](int c) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = tid + TT * i;
        const int kk = idx / v4, c4 = idx - kk * v4;
        wreg[i] = (kk < 32 && 32 * c + kk < h)
                      ? *reinterpret_cast<const float4 *>(L.W + (size_t)(32 * c + kk) * hp + 4 * c4)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    load_chunk(0);
    for (int c = 0; c < nchunk; ++c) {
      float *wbuf = wb + (c & 1) * (32 * WK_LD);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = tid + TT * i;
        const int kk = idx / v4, c4 = idx - kk * v4;
        if (kk < 32) *reinterpret_cast<float4 *>(wbuf + kk * WK_LD + 4 * c4) = wreg[i];
      }
      __syncthreads();
      if (c + 1 < nchunk) load_chunk(c + 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float4 av = *reinterpret_cast<const float4 *>(cur + c16 * ACT_LD + 32 * c + 16 * j + 4 * q);
        const float af[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int t = wave + 4 * i;
          if (t < NT) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[i] = mfma16(af[e], wbuf[(16 * j + 4 * q + e) * WK_LD + 16 * t + c16], acc[i]);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = wave + 4 * i;
      if (t < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) nxt[(4 * q + r) * ACT_LD + 16 * t + c16] = acc[i][r];
      }
    }
    __syncthreads();
    float *tmp = cur; cur = nxt; nxt = tmp;
  }
}

// ---------------------------------------------------------------------------------------------
bool tail_supported(const stdadk_mlp_desc *d, int first_layer) {
  if (first_layer < 1 || d->n_hidden < 1) return false;
  if (d->out_dim > TAIL_MAXQ) return false;
  for (int l = first_layer - 1; l < d->n_hidden; ++l)
    if (d->hidden[l] > TAIL_MAX_W || (d->hidden[l] & 15)) return false;
  return true;
}

static size_t fwd_lds() { return (2 * R * ACT_LD + 2 * TAIL_MAX_W * WF_LD) * sizeof(float); }
static size_t bwd_lds() { return (2 * R * ACT_LD + 2 * 32 * WK_LD + 3 * 4 * 256 + R * TAIL_MAXQ) * sizeof(float); }

int tail_forward(const TailFwdArgs &a, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {   // once per process, never inside a stream capture (the first step runs eagerly)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tail_fwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds());
    if (e != hipSuccess) { set_error("tail_forward: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  STDADK_LAUNCH(tail_fwd_kernel, dim3((unsigned)ceil_div(a.B, R)), dim3(TT), fwd_lds(), st, a);
  STDADK_CHECK_LAUNCH("tail_forward");
  return 0;
}

int tail_backward(const TailBwdArgs &a, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {   // once per process, never inside a stream capture (the first step runs eagerly)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tail_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds());
    if (e != hipSuccess) { set_error("tail_backward: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  STDADK_LAUNCH(tail_bwd_kernel, dim3((unsigned)ceil_div(a.B, R)), dim3(TT), bwd_lds(), st, a);
  STDADK_CHECK_LAUNCH("tail_backward");
  return 0;
}

}  // namespace stdadk
