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
    // temporal rows of W0^T into LDS: ALL of a thread's pieces requested first, then stored (Kt H 4 B <= 96 KiB =>
    // at most 6 float4 per thread; clamped, unconditional loads).  As a rolled `dst[i] = src[i]` loop this was
    // Kt H / 4096 dependent L2 round trips -- five for the 70 temporal knots -- at the head of every workgroup.
    const float4 *src = reinterpret_cast<const float4 *>(a.W0T + (size_t)D0 * H);
    float4 *dst = reinterpret_cast<float4 *>(Wt);
    const int n4 = Kt * H / 4;
    if (n4 > 0) {                                   // workgroup-uniform (a model without temporal knots)
      float4 tmp[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) tmp[k] = src[min(tid + FW_T * k, n4 - 1)];
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (tid + FW_T * k < n4) dst[tid + FW_T * k] = tmp[k];
    }
  }
  __syncthreads();
  float *my_phi = lphi + wave * LIST;
  int *my_k = lk + wave * LIST;
  float *my_psi = lpsi + wave * Kt_pad;
  const uint64_t seed = a.seed + (a.step_dev ? (uint64_t)a.step_dev[0] * 0x9E3779B97F4A7C15ULL : 0ULL);
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const uint32_t drop_thr = drop_threshold(a.drop_p);
  const uint64_t below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));

  for (int row = r0 + wave; row < r1; row += FW_T / 64) {
    const bool first = row == r0 + wave;
    const float x = first ? xf : a.xs[row], y = first ? yf : a.ys[row], t = first ? tf : a.ts[row];
    float acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = a.raw ? 0.f : bias[c];

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

    // The same gather-FMA over a list zero-padded to a multiple of 16, software-pipelined: the eight rows of group
    // g + 1 are requested before the multiply-adds of group g (two register sets), so the ~8 groups of an observation
    // cost about one L2 round trip plus their FMAs instead of one round trip EACH (the gather was 4-5 us of the 13.6 us
    // layer-0 phase of the fused step kernel).  Same entries in the same order, and fmaf(0, w, acc) == acc for the
    // padding: bit-identical to consume().  The loads are unconditional -- the last iteration re-requests a group it
    // does not use -- so that the loop body is straight-line code and every wait is an exact count.
    auto consume_pipelined = [&](int cnt) {
      auto request = [&](int e0, float *pv, typename VecT<CPL>::T *wv) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pv[e] = my_phi[e0 + e];
          const unsigned ro = (unsigned)my_k[e0 + e] * (unsigned)H + (unsigned)(CPL * lane);
          wv[e] = *reinterpret_cast<const typename VecT<CPL>::T *>(a.W0T + (size_t)ro);
        }
      };
      auto multiply = [&](const float *pv, const typename VecT<CPL>::T *wv) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float *f = reinterpret_cast<const float *>(&wv[e]);
#pragma unroll
          for (int c = 0; c < CPL; ++c) acc[c] = fmaf(pv[e], f[c], acc[c]);
        }
      };
      float pa[8], pb[8];
      typename VecT<CPL>::T wa[8], wb[8];
      if (cnt > 0) request(0, pa, wa);
      for (int e0 = 0; e0 < cnt; e0 += 16) {
        request(e0 + 8, pb, wb);
        multiply(pa, wa);
        request(e0 + 16 < cnt ? e0 + 16 : e0 + 8, pa, wa);
        multiply(pb, wb);
      }
    };

    // ---- spatial levels through the per-wave candidate list (fixed knots: three levels at a time,
    // 6 x 6 candidates each; free knots: one level at a time, (2R)^2 candidates in passes of 64)
    constexpr int LSTEP = FREE ? 1 : 3;
    for (int l0 = 0; l0 < a.g.n_levels; l0 += LSTEP) {
      int n = 0;
      const int l1 = min(l0 + LSTEP, a.g.n_levels);
      for (int l = l0; FREE && l < l1; ++l) {
        // a candidate's value into the per-wave list (non-zeros compacted by ballot + popcount; a full list is
        // consumed, zero-padded to 8, and started over)
        auto push = [&](float phi, int k) {
          const uint64_t mask = __ballot(phi != 0.f);
          const int m = __popcll(mask);
          if (n + m > LIST - 8) {
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
        };
        if (a.kperm) {
          // scattered knots: the level's knots were binned into a Gk x Gk cell grid (knot_bins); the candidates are
          // the knots of the cells within ceil(reach Gk) + 1 of the observation's cell -- every knot whose support
          // can reach it (a knot outside the domain sits in a border cell: never farther in cells than in fact).
          // A row of cells is one contiguous run of kperm: the rows' runs are walked as ONE flat list, 64 candidates
          // per pass (the same segment walk as the per-knot gather of dW0^T).
          const int Gk = a.Gk;
          const float rc = a.reach[l];
          const int rad = (rc < 4.0f) ? (int)ceilf(rc * (float)Gk) + 1 : Gk;        // NaN / huge: the whole level
          const int ocx = floor_clamp(x * (float)Gk, Gk), ocy = floor_clamp(y * (float)Gk, Gk);
          const int cx_lo = max(ocx - rad, 0), cx_hi = min(ocx + rad, Gk - 1);
          const int cy_lo = max(ocy - rad, 0), cy_hi = min(ocy + rad, Gk - 1);
          const int *cs = a.kcs + (size_t)l * (Gk * Gk + 1);
          const int cxl = cx_lo + lane;                                              // Gk <= 64: one lane per cell row
          int seg0 = 0, seg1 = 0;
          if (cxl <= cx_hi) { seg0 = cs[cxl * Gk + cy_lo]; seg1 = cs[cxl * Gk + cy_hi + 1]; }
          int incl = seg1 - seg0;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
          }
          const int total = __shfl(incl, 63, 64);
          for (int f0 = 0; f0 < total; f0 += 64) {
            const int f = f0 + lane;
            int lo = 0;
#pragma unroll
            for (int st = 32; st > 0; st >>= 1) {
              const int probe = __shfl(incl, lo + st - 1, 64);
              if (probe <= f) lo += st;
            }
            const int jl = lo < 63 ? lo : 63;
            const int pin = __shfl(incl, jl, 64);
            const int pl = __shfl(seg1 - seg0, jl, 64);
            const int ps0 = __shfl(seg0, jl, 64);
            float phi = 0.f;
            int k = 0;
            if (f < total) {
              k = a.kperm[ps0 + (f - (pin - pl))];
              phi = phi_eval<BASIS>(x, y, a.g.centers[2 * k], a.g.centers[2 * k + 1], knot_scale(a.g.bw[k], a.g.cal));
            }
            push(phi, k);
          }
          continue;
        }
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
            push(phi, k);
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
      if (FREE) {
        const int npad = (n + 7) & ~7;
        if (lane < npad - n) { my_phi[n + lane] = 0.f; my_k[n + lane] = 0; }
        __builtin_amdgcn_wave_barrier();
        consume(npad);
      } else {
        const int npad = (n + 15) & ~15;          // n <= 3 * 36 + 16 = 124, LIST = 144
        if (lane < npad - n) { my_phi[n + lane] = 0.f; my_k[n + lane] = 0; }
        __builtin_amdgcn_wave_barrier();
        consume_pipelined(npad);
      }
      __builtin_amdgcn_wave_barrier();
    }

    if (a.raw) {                           // wave-uniform: the spatial part alone
      typename VecT<CPL>::T o;
      float *fo = reinterpret_cast<float *>(&o);
#pragma unroll
      for (int c = 0; c < CPL; ++c) fo[c] = acc[c];
      *reinterpret_cast<typename VecT<CPL>::T *>(a.act + (size_t)row * H + CPL * lane) = o;
      continue;
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
    {
      // eight rows of the LDS image in flight (a row at a time is a chain of Kt dependent LDS round trips); same
      // order of the multiply-adds
      int j = 0;
      for (; j + 8 <= Kt; j += 8) {
        float pv[8];
        typename VecT<CPL>::T wv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pv[e] = my_psi[j + e];
          wv[e] = *reinterpret_cast<const typename VecT<CPL>::T *>(Wt + (size_t)(j + e) * H + CPL * lane);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float *f = reinterpret_cast<const float *>(&wv[e]);
#pragma unroll
          for (int c = 0; c < CPL; ++c) acc[c] = fmaf(pv[e], f[c], acc[c]);
        }
      }
      for (; j < Kt; ++j) fma_row<CPL>(acc, my_psi[j], Wt + (size_t)j * H + CPL * lane);
    }
    __builtin_amdgcn_wave_barrier();

    // ---- LayerNorm -> ReLU -> Dropout (row-local: this wave owns the whole row)
    float mean = 0.f, rs = 1.f;
    if (LN) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) s += acc[c];
      mean = wave_sum(s) * (1.0f / (float)H);
      float sq = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) { float d = acc[c] - mean; sq += d * d; }
      rs = ln_rstd(wave_sum(sq) * (1.0f / (float)H), a.eps);
      if (lane == 0 && a.rstd) a.rstd[row] = rs;
    }
    float xh[CPL], av[CPL];
    const uint32_t rowkey = drop_rowkey(seed, 0, row);
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      xh[c] = LN ? (acc[c] - mean) * rs : acc[c];
      float u = LN ? fmaf(xh[c], gam[c], bet[c]) : xh[c];
      float v = fmaxf(u, 0.f);
      if (a.drop_p > 0.f) {
        bool keep = drop_keep(rowkey, CPL * lane + c, drop_thr);
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

// ---------------------------------------------------------------------------------------------------------
// R consecutive observations per wave (fixed grid knots).  The observations are cell-sorted, so neighbours in
// the sorted order see almost the same knots: the candidates of a level are the bounding box of the R 6 x 6
// windows (at most 8 x 8 = one per lane; a knot outside an observation's own window evaluates to exactly 0 for
// it), a row of W0^T is fetched ONCE and feeds R accumulators, and so does a temporal row from LDS.  The L2
// row-gather traffic per observation -- what the one-observation-per-wave body is bound by -- drops to the
// size of the union over R.  Every observation still sums its own non-zero knots in the same order (level by
// level, ix-major), and fmaf(0, w, acc) == acc, so the result is bit-identical to R = 1.  When the windows of a
// group are too far apart for one 8 x 8 box (wrap-around of the cell order), the chunk of levels is done one
// observation at a time.
template <int CPL, bool LN, int BASIS, int R>
__device__ __forceinline__ void l1_window_fwd_multi_body(const L1FwdArgs &a, float *smem, const int r0, const int r1) {
  constexpr int H = 64 * CPL;
  constexpr int NW = FW_T / 64;
  const int Kt = a.g.Kt;
  const int Kt_pad = (Kt + 3) & ~3;
  float *Wt = smem;                                          // [Kt][H] temporal rows of W0^T
  float *lphi = Wt + (size_t)Kt * H;                         // [NW][LIST][R]
  int *lk = reinterpret_cast<int *>(lphi + NW * LIST * R);   // [NW][LIST]
  float *lpsi = reinterpret_cast<float *>(lk + NW * LIST);   // [NW][Kt_pad][R]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D0 = a.g.p + a.g.Ks;
  const int rowf = r0 + R * wave;
  {
    // temporal rows of W0^T into LDS: ALL of a thread's pieces requested first, then stored (Kt H 4 B <= 96 KiB =>
    // at most 6 float4 per thread; clamped, unconditional loads).  As a rolled `dst[i] = src[i]` loop this was
    // Kt H / 4096 dependent L2 round trips -- five for the 70 temporal knots -- at the head of every workgroup.
    const float4 *src = reinterpret_cast<const float4 *>(a.W0T + (size_t)D0 * H);
    float4 *dst = reinterpret_cast<float4 *>(Wt);
    const int n4 = Kt * H / 4;
    if (n4 > 0) {                                   // workgroup-uniform (a model without temporal knots)
      float4 tmp[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) tmp[k] = src[min(tid + FW_T * k, n4 - 1)];
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (tid + FW_T * k < n4) dst[tid + FW_T * k] = tmp[k];
    }
  }
  __syncthreads();
  float *my_phi = lphi + wave * LIST * R;
  int *my_k = lk + wave * LIST;
  float *my_psi = lpsi + wave * Kt_pad * R;
  const uint64_t seed = a.seed + (a.step_dev ? (uint64_t)a.step_dev[0] * 0x9E3779B97F4A7C15ULL : 0ULL);
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const uint32_t drop_thr = drop_threshold(a.drop_p);
  const uint64_t below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
  const int bdx = lane >> 3, bdy = lane & 7;              // this lane's place in an 8 x 8 candidate box

  for (int row = rowf; row < r1; row += R * NW) {
    const int nv = min(R, r1 - row);                      // observations of this group (the rest repeat the last)
    float x[R], y[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = row + min(r, nv - 1);
      x[r] = a.xs[i]; y[r] = a.ys[i];
    }
    float acc[R][CPL];
    {
      const typename VecT<CPL>::T bv = *reinterpret_cast<const typename VecT<CPL>::T *>(a.b0 + CPL * lane);
      const float *bf = reinterpret_cast<const float *>(&bv);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[r][c] = a.raw ? 0.f : bf[c];
    }

    // gather-FMA of the first `cnt` list entries (cnt a multiple of 8): NF rows of W0^T in flight, R uses of each
    constexpr int NF = 8;
    auto consume = [&](int cnt) {
      for (int e0 = 0; e0 < cnt; e0 += NF) {
        unsigned ro[NF];
#pragma unroll
        for (int e = 0; e < NF; ++e) ro[e] = (unsigned)my_k[e0 + e] * (unsigned)H + (unsigned)(CPL * lane);
        typename VecT<CPL>::T wv[NF];
#pragma unroll
        for (int e = 0; e < NF; ++e) wv[e] = *reinterpret_cast<const typename VecT<CPL>::T *>(a.W0T + (size_t)ro[e]);
#pragma unroll
        for (int e = 0; e < NF; ++e) {
          const float *f = reinterpret_cast<const float *>(&wv[e]);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float pv = my_phi[(e0 + e) * R + r];
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[r][c] = fmaf(pv, f[c], acc[r][c]);
          }
        }
      }
    };
    auto flush = [&](int n) {                             // zero-pad to a multiple of 8 and consume
      const int npad = (n + 7) & ~7;
      if (lane < npad - n) {
#pragma unroll
        for (int r = 0; r < R; ++r) my_phi[(n + lane) * R + r] = 0.f;
        my_k[n + lane] = 0;
      }
      __builtin_amdgcn_wave_barrier();
      consume(npad);
      __builtin_amdgcn_wave_barrier();
    };

    for (int l0 = 0; l0 < a.g.n_levels; l0 += 3) {
      const int l1 = min(l0 + 3, a.g.n_levels);
      // bounding box of the R windows per level (wave-uniform)
      int bx0[3], by0[3], sx[3], sy[3];
      bool far = false;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        bx0[j] = by0[j] = 0; sx[j] = sy[j] = 0;
        if (l0 + j < l1) {
          const int side = a.g.side[l0 + j];
          const int win = side < WIN ? side : WIN;
          int mnx = 1 << 30, mny = 1 << 30, mxx = 0, mxy = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int ix0 = window_start(x[r], side, win), iy0 = window_start(y[r], side, win);
            mnx = min(mnx, ix0); mxx = max(mxx, ix0); mny = min(mny, iy0); mxy = max(mxy, iy0);
          }
          bx0[j] = mnx; by0[j] = mny; sx[j] = mxx - mnx + win; sy[j] = mxy - mny + win;
          far = far || sx[j] > 8 || sy[j] > 8;
        }
      }
      const int npass = far ? nv : 1;
      for (int pass = 0; pass < npass; ++pass) {
        int n = 0;
        int kk[3];
        bool ok[3];
        float kx[3], ky[3], kb[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          kk[j] = 0; ok[j] = false;
          if (l0 + j < l1) {
            const int l = l0 + j;
            const int side = a.g.side[l];
            const int win = side < WIN ? side : WIN;
            int ix0 = bx0[j], iy0 = by0[j], spx = sx[j], spy = sy[j];
            if (far) {
              float xp = x[0], yp = y[0];
#pragma unroll
              for (int r = 1; r < R; ++r) if (r == pass) { xp = x[r]; yp = y[r]; }
              ix0 = window_start(xp, side, win); iy0 = window_start(yp, side, win); spx = spy = win;
            }
            ok[j] = bdx < spx && bdy < spy;
            kk[j] = a.g.off[l] + (ok[j] ? (ix0 + bdx) * side + iy0 + bdy : 0);
          }
          kx[j] = a.g.centers[2 * kk[j]];
          ky[j] = a.g.centers[2 * kk[j] + 1];
          kb[j] = a.g.bw[kk[j]];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (l0 + j < l1) {
            float phi[R];
            bool any = false;
            const float sc = knot_scale(kb[j], a.g.cal);
#pragma unroll
            for (int r = 0; r < R; ++r) {
              phi[r] = (ok[j] && (!far || r == pass)) ? phi_eval<BASIS>(x[r], y[r], kx[j], ky[j], sc) : 0.f;
              any = any || phi[r] != 0.f;
            }
            const uint64_t mask = __ballot(any);
            const int m = __popcll(mask);
            if (n + m > LIST - 8 - WIN_MAX_P) { flush(n); n = 0; }     // wave-uniform
            if (any) {
              const int pos = n + __popcll(mask & below);
#pragma unroll
              for (int r = 0; r < R; ++r) my_phi[pos * R + r] = phi[r];
              my_k[pos] = a.g.p + kk[j];
            }
            n += m;
          }
        }
        if (l0 == 0 && a.g.p > 0) {            // covariate columns [0, p): dense, per observation
          if (lane < a.g.p) {
#pragma unroll
            for (int r = 0; r < R; ++r)
              my_phi[(n + lane) * R + r] =
                  (!far || r == pass) ? a.Xs[(size_t)(row + min(r, nv - 1)) * a.g.p + lane] : 0.f;
            my_k[n + lane] = lane;
          }
          n += a.g.p;
        }
        flush(n);
      }
    }

    if (a.raw) {                           // wave-uniform: the spatial part alone
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r < nv) {
          typename VecT<CPL>::T o;
          float *fo = reinterpret_cast<float *>(&o);
#pragma unroll
          for (int c = 0; c < CPL; ++c) fo[c] = acc[r][c];
          *reinterpret_cast<typename VecT<CPL>::T *>(a.act + (size_t)(row + r) * H + CPL * lane) = o;
        }
      }
      continue;
    }
    // ---- temporal basis: each LDS row feeds the R accumulators
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float tr = a.ts[row + min(r, nv - 1)];
      for (int j = lane; j < Kt; j += 64) {
        const float v = psi_eval(tr, a.g.t_centers[j], a.g.t_bw[j]);
        my_psi[j * R + r] = v;
        if (a.psi && r < nv) a.psi[(size_t)(row + r) * a.ld_psi + j] = v;
      }
      if (a.psi && r < nv)
        for (int j = Kt + lane; j < a.ld_psi; j += 64) a.psi[(size_t)(row + r) * a.ld_psi + j] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    {
      // four rows of the LDS image (and their R factors) in flight; same order of the multiply-adds
      int j = 0;
      for (; j + 4 <= Kt; j += 4) {
        typename VecT<CPL>::T wv[4];
        float sv[4][R];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          wv[e] = *reinterpret_cast<const typename VecT<CPL>::T *>(Wt + (size_t)(j + e) * H + CPL * lane);
#pragma unroll
          for (int r = 0; r < R; ++r) sv[e][r] = my_psi[(j + e) * R + r];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float *f = reinterpret_cast<const float *>(&wv[e]);
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[r][c] = fmaf(sv[e][r], f[c], acc[r][c]);
        }
      }
      for (; j < Kt; ++j) {
        typename VecT<CPL>::T v = *reinterpret_cast<const typename VecT<CPL>::T *>(Wt + (size_t)j * H + CPL * lane);
        const float *f = reinterpret_cast<const float *>(&v);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float s = my_psi[j * R + r];
#pragma unroll
          for (int c = 0; c < CPL; ++c) acc[r][c] = fmaf(s, f[c], acc[r][c]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- LayerNorm -> ReLU -> Dropout, row by row
    float gam[CPL], bet[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      gam[c] = LN ? a.gamma[CPL * lane + c] : 1.f;
      bet[c] = LN ? a.beta[CPL * lane + c] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < nv) {
        const int orow = row + r;
        float mean = 0.f, rs = 1.f;
        if (LN) {
          float s = 0.f;
#pragma unroll
          for (int c = 0; c < CPL; ++c) s += acc[r][c];
          mean = wave_sum(s) * (1.0f / (float)H);
          float sq = 0.f;
#pragma unroll
          for (int c = 0; c < CPL; ++c) { float d = acc[r][c] - mean; sq += d * d; }
          rs = ln_rstd(wave_sum(sq) * (1.0f / (float)H), a.eps);
          if (lane == 0 && a.rstd) a.rstd[orow] = rs;
        }
        typename VecT<CPL>::T o1, o2;
        float *f1 = reinterpret_cast<float *>(&o1), *f2 = reinterpret_cast<float *>(&o2);
        const uint32_t rowkey = drop_rowkey(seed, 0, orow);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const float xh = LN ? (acc[r][c] - mean) * rs : acc[r][c];
          const float u = LN ? fmaf(xh, gam[c], bet[c]) : xh;
          float v = fmaxf(u, 0.f);
          if (a.drop_p > 0.f) {
            const bool keep = drop_keep(rowkey, CPL * lane + c, drop_thr);
            v = keep ? v * keep_scale : 0.f;
          }
          f1[c] = xh; f2[c] = v;
        }
        if (a.xhat) *reinterpret_cast<typename VecT<CPL>::T *>(a.xhat + (size_t)orow * H + CPL * lane) = o1;
        *reinterpret_cast<typename VecT<CPL>::T *>(a.act + (size_t)orow * H + CPL * lane) = o2;
      }
    }
  }
}

}  // namespace stdadk
