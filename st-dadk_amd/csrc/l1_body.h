// Body of the layer-0 window forward (see window.h / window.hip), shared by its own kernel and by the
// fused training-step kernel (fused_step.hip).
#pragma once
#include "window.h"
#include "basis.h"

namespace stdadk {

// half-width R (grid cells) of the candidate window of a level from knot_halo()'s partial maxima:
// ceil(max + eps), clamped to [1, side]
__device__ __forceinline__ int halo_half_width(const float *__restrict__ halo, int l, int side) {
  float m = 0.f;
#pragma unroll
  for (int s = 0; s < HALO_SPLIT; ++s) m = fmaxf(m, halo[l * HALO_SPLIT + s]);
  const float r = ceilf(m + 1e-3f);
  return r >= (float)side ? side : (r < 1.f ? 1 : (int)r);
}

// XCD-aware chunking: workgroups w and w+8 share an XCD (round-robin dispatch), so XCD x walks the
// x-th contiguous eighth of the sorted observations and its L2 holds that eighth's W0^T rows.
__device__ __forceinline__ int l1_chunk_of(int w, int n_wg) {
  const int per_x = n_wg >> 3;
  return (w & 7) * per_x + (w >> 3);
}

constexpr int FW_T = 1024;   // 16 waves share one LDS copy of the temporal rows (1 workgroup per CU)
constexpr int LIST = 144;   // >= 3*36 + WIN_MAX_P rounded up to 8 (8 levels are chunked below)

template <int CPL>
struct VecT;
template <>
struct VecT<4> { using T = float4; };
template <>
struct VecT<2> { using T = float2; };
template <>
struct VecT<1> { using T = float; };

template <int CPL>
__device__ __forceinline__ void fma_row(float *acc, float s, const float *row) {
  typename VecT<CPL>::T v = *reinterpret_cast<const typename VecT<CPL>::T *>(row);
  const float *f = reinterpret_cast<const float *>(&v);
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = fmaf(s, f[c], acc[c]);
}

// rows [r0, r1) of the sorted batch by this workgroup (one wave per observation)
template <int CPL, bool LN, int BASIS, bool FREE>
__device__ __forceinline__ void l1_window_fwd_body(const L1FwdArgs &a, float *smem, const int r0, const int r1) {
  constexpr int H = 64 * CPL;
  constexpr int NW = FW_T / 64;
  const int Kt = a.g.Kt;
  float *Wt = smem;                                   // [Kt][H] temporal rows of W0^T
  float *lphi = Wt + (size_t)Kt * H;                  // [NW][LIST]
  int *lk = reinterpret_cast<int *>(lphi + NW * LIST); // [NW][LIST]
  float *lpsi = reinterpret_cast<float *>(lk + NW * LIST);  // [NW][Kt_pad]
  const int Kt_pad = (Kt + 3) & ~3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D0 = a.g.p + a.g.Ks;                      // first temporal row of W0^T
  // the wave's first observation and the per-column parameters are requested before the temporal rows
  // are staged, so that all of it shares one memory round trip ahead of the workgroup barrier
  const int rowf = min(r0 + wave, a.B - 1);
  const float xf = a.xs[rowf], yf = a.ys[rowf], tf = a.ts[rowf];
  float bias[CPL], gam[CPL], bet[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    bias[c] = a.b0[CPL * lane + c];
    gam[c] = LN ? a.gamma[CPL * lane + c] : 1.f;
    bet[c] = LN ? a.beta[CPL * lane + c] : 0.f;
  }
  {
    const float4 *src = reinterpret_cast<const float4 *>(a.W0T + (size_t)D0 * H);
    float4 *dst = reinterpret_cast<float4 *>(Wt);
    for (int i = tid; i < Kt * H / 4; i += FW_T) dst[i] = src[i];
  }
  __syncthreads();
  float *my_phi = lphi + wave * LIST;
  int *my_k = lk + wave * LIST;
  float *my_psi = lpsi + wave * Kt_pad;
  const uint64_t seed = a.seed + (a.step_dev ? (uint64_t)a.step_dev[0] * 0x9E3779B97F4A7C15ULL : 0ULL);
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const uint64_t below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));

  for (int row = r0 + wave; row < r1; row += FW_T / 64) {
    const bool first = row == r0 + wave;
    const float x = first ? xf : a.xs[row], y = first ? yf : a.ys[row], t = first ? tf : a.ts[row];
    float acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = bias[c];

    // gather-FMA of the first `cnt` list entries (cnt a multiple of 8): 8 rows of W0^T in flight
    auto consume = [&](int cnt) {
      for (int e0 = 0; e0 < cnt; e0 += 8) {
        float pv[8];
        unsigned ro[8];              // 32-bit element offsets: scalar base + vector offset addressing
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pv[e] = my_phi[e0 + e];
          ro[e] = (unsigned)my_k[e0 + e] * (unsigned)H + (unsigned)(CPL * lane);
        }
        typename VecT<CPL>::T wv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = *reinterpret_cast<const typename VecT<CPL>::T *>(a.W0T + (size_t)ro[e]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float *f = reinterpret_cast<const float *>(&wv[e]);
#pragma unroll
          for (int c = 0; c < CPL; ++c) acc[c] = fmaf(pv[e], f[c], acc[c]);
        }
      }
    };

    // ---- spatial levels through the per-wave candidate list (fixed knots: three levels at a time,
    // 6 x 6 candidates each; free knots: one level at a time, (2R)^2 candidates in passes of 64)
    constexpr int LSTEP = FREE ? 1 : 3;
    for (int l0 = 0; l0 < a.g.n_levels; l0 += LSTEP) {
      int n = 0;
      const int l1 = min(l0 + LSTEP, a.g.n_levels);
      for (int l = l0; FREE && l < l1; ++l) {
        const int side = a.g.side[l];
        {
          const int R = halo_half_width(a.halo, l, side);
          const int win = min(2 * R, side);
          const int hi = side - win;
          const int fx = floor_clamp(x * (float)(side - 1), side), fy = floor_clamp(y * (float)(side - 1), side);
          const int ix0 = min(max(fx - R + 1, 0), hi), iy0 = min(max(fy - R + 1, 0), hi);
          const int ncand = win * win;
          for (int e0 = 0; e0 < ncand; e0 += 64) {
            const int e = e0 + lane;
            const int dx = e / win, dy = e - dx * win;
            float phi = 0.f;
            int k = 0;
            if (e < ncand) {
              k = a.g.off[l] + (ix0 + dx) * side + iy0 + dy;
              phi = phi_eval<BASIS>(x, y, a.g.centers[2 * k], a.g.centers[2 * k + 1],
                                    knot_scale(a.g.bw[k], a.g.cal));
            }
            const uint64_t mask = __ballot(phi != 0.f);
            const int m = __popcll(mask);
            if (n + m > LIST - 8) {        // list full: consume it (zero-padded to 8) and start over
              const int npad = (n + 7) & ~7;
              if (lane < npad - n) { my_phi[n + lane] = 0.f; my_k[n + lane] = 0; }
              __builtin_amdgcn_wave_barrier();
              consume(npad);
              __builtin_amdgcn_wave_barrier();
              n = 0;
            }
            if (phi != 0.f) {
              const int pos = n + __popcll(mask & below);
              my_phi[pos] = phi;
              my_k[pos] = a.g.p + k;
            }
            n += m;
          }
        }
      }
      if (!FREE) {
        // knot table entries of all (up to three) levels of the chunk requested together, from clamped
        // indices and unconditionally, so they share one round trip; evaluated level by level after
        int kk[3];
        bool ok[3];
        float kx[3], ky[3], kb[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          kk[j] = 0; ok[j] = false;
          if (l0 + j < l1) {                      // wave-uniform
            const int l = l0 + j;
            const int side = a.g.side[l];
            const int win = side < WIN ? side : WIN;
            const int ix0 = window_start(x, side, win), iy0 = window_start(y, side, win);
            const int dx = lane / WIN, dy = lane - dx * WIN;
            ok[j] = lane < WIN * WIN && dx < win && dy < win;
            kk[j] = a.g.off[l] + (ok[j] ? (ix0 + dx) * side + iy0 + dy : 0);
          }
          kx[j] = a.g.centers[2 * kk[j]];
          ky[j] = a.g.centers[2 * kk[j] + 1];
          kb[j] = a.g.bw[kk[j]];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (l0 + j < l1) {
            const float phi = ok[j] ? phi_eval<BASIS>(x, y, kx[j], ky[j], knot_scale(kb[j], a.g.cal)) : 0.f;
            const uint64_t mask = __ballot(phi != 0.f);
            if (phi != 0.f) {
              const int pos = n + __popcll(mask & below);
              my_phi[pos] = phi;
              my_k[pos] = a.g.p + kk[j];
            }
            n += __popcll(mask);
          }
        }
      }
      if (l0 == 0 && a.g.p > 0) {            // covariate columns [0, p): dense
        if (lane < a.g.p) {
          my_phi[n + lane] = a.Xs[(size_t)row * a.g.p + lane];
          my_k[n + lane] = lane;
        }
        n += a.g.p;
      }
      const int npad = (n + 7) & ~7;
      if (lane < npad - n) { my_phi[n + lane] = 0.f; my_k[n + lane] = 0; }
      __builtin_amdgcn_wave_barrier();
      consume(npad);
      __builtin_amdgcn_wave_barrier();
    }

    // ---- temporal basis: rows from LDS
    for (int j = lane; j < Kt; j += 64) {
      float v = psi_eval(t, a.g.t_centers[j], a.g.t_bw[j]);
      my_psi[j] = v;
      if (a.psi) a.psi[(size_t)row * a.ld_psi + j] = v;
    }
    if (a.psi)
      for (int j = Kt + lane; j < a.ld_psi; j += 64) a.psi[(size_t)row * a.ld_psi + j] = 0.f;
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < Kt; ++j) fma_row<CPL>(acc, my_psi[j], Wt + (size_t)j * H + CPL * lane);
    __builtin_amdgcn_wave_barrier();

    // ---- LayerNorm -> ReLU -> Dropout (row-local: this wave owns the whole row)
    float mean = 0.f, rs = 1.f;
    if (LN) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) s += acc[c];
      mean = wave_sum(s) / (float)H;
      float sq = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) { float d = acc[c] - mean; sq += d * d; }
      rs = 1.0f / sqrtf(wave_sum(sq) / (float)H + a.eps);
      if (lane == 0 && a.rstd) a.rstd[row] = rs;
    }
    float xh[CPL], av[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      xh[c] = LN ? (acc[c] - mean) * rs : acc[c];
      float u = LN ? fmaf(xh[c], gam[c], bet[c]) : xh[c];
      float v = fmaxf(u, 0.f);
      if (a.drop_p > 0.f) {
        bool keep = drop_keep(seed, 0, (int64_t)row * H + CPL * lane + c, a.drop_p);
        v = keep ? v * keep_scale : 0.f;
      }
      av[c] = v;
    }
    typename VecT<CPL>::T o1, o2;
    float *f1 = reinterpret_cast<float *>(&o1), *f2 = reinterpret_cast<float *>(&o2);
#pragma unroll
    for (int c = 0; c < CPL; ++c) { f1[c] = xh[c]; f2[c] = av[c]; }
    if (a.xhat) *reinterpret_cast<typename VecT<CPL>::T *>(a.xhat + (size_t)row * H + CPL * lane) = o1;
    *reinterpret_cast<typename VecT<CPL>::T *>(a.act + (size_t)row * H + CPL * lane) = o2;
  }
}

}  // namespace stdadk
