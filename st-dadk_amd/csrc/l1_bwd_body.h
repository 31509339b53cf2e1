// Body of the per-knot gather of dW0^T (see window.h / window.hip), shared by its own kernel and by the
// merged weight-gradient kernel (dw_all.hip).  `block` = index of the group of BW_T/64 knots.
#pragma once
#include "window.h"
#include "basis.h"
#include "l1_body.h"

namespace stdadk {

constexpr int BW_T = 256;        // 4 waves = 4 knots per workgroup
constexpr int BW_LIST = 64;      // compacted candidates per flush

// d phi / d r as autograd differentiates the forward formulas (compact-support bases of this path)
template <int BASIS>
__device__ __forceinline__ float basis_prime_cs(float r) {
  if (BASIS == STDADK_BASIS_WENDLAND) {
    if (!(r < 1.0f)) return 0.f;
    const float om = 1.0f - r, om2 = om * om;
    return (-56.0f / 3.0f) * r * (om2 * om2 * om) * fmaf(5.0f, r, 1.0f);
  }
  return r <= 1.0f ? -1.0f : 0.f;      // triangular
}

// KNOTS (learnable knots): the wave also accumulates sum_b q_b dZ[b,:] for q = the three per-pair
// factors of d cx, d cy, d log_bw; one dot product with its W0^T row at the end turns them into the
// knot's gradient — sum_b (dZ[b,:] . W0^T[k,:]) q_b without a reduction per pair.
// Returns this lane's share of the squares of the row of dW0^T its wave wrote (0 for a wave without a knot).
template <int CPL, int BASIS, bool KNOTS>
__device__ __forceinline__ float l1_window_bwd_body(const L1BwdArgs &a, const int block) {
  constexpr int H = 64 * CPL;
  constexpr int NQ = KNOTS ? 3 : 0;
  __shared__ float lphi[BW_T / 64][BW_LIST + 8];
  __shared__ int lidx[BW_T / 64][BW_LIST + 8];
  __shared__ float lq[KNOTS ? 3 : 1][BW_T / 64][KNOTS ? BW_LIST + 8 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int k;
  if (a.xcd_slots > 0) {
    // XCD-striped order (see l1_window_bwd_multi_body): XCD block & 7 owns the grid rows [x side/8, (x+1) side/8)
    // of every level
    const int x = block & 7;
    int q = (block >> 3) * (BW_T / 64) + wave, l = 0, r0 = 0;
    for (; l < a.g.n_levels; ++l) {
      const int side = a.g.side[l];
      r0 = (x * side) >> 3;
      const int np = ((((x + 1) * side) >> 3) - r0) * side;
      if (q < np) break;
      q -= np;
    }
    if (l >= a.g.n_levels) return 0.f;
    k = a.g.off[l] + r0 * a.g.side[l] + q;            // q = (ix - r0) * side + iy
  } else {
    k = block * (BW_T / 64) + wave;                   // knots in table order: level 0 (coarsest) first
    if (k >= a.g.Ks) return 0.f;
  }
  float *my_phi = lphi[wave];
  int *my_idx = lidx[wave];
  const float kcx = a.g.centers[2 * k], kcy = a.g.centers[2 * k + 1];
  const float kbw = a.g.bw[k];
  const float ksc = knot_scale(kbw, a.g.cal);
  const float r = kbw * a.g.cal;                      // support radius
  const int G = a.G;
  // cells overlapping the support square, one cell of margin against rounding
  const int cx_lo = max(floor_clamp((kcx - r) * (float)G, G) - 1, 0);
  const int cx_hi = min(floor_clamp((kcx + r) * (float)G, G) + 1, G - 1);
  const int cy_lo = max(floor_clamp((kcy - r) * (float)G, G) - 1, 0);
  const int cy_hi = min(floor_clamp((kcy + r) * (float)G, G) + 1, G - 1);
  const uint64_t below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));

  float acc[CPL];
  float qacc[KNOTS ? 3 : 1][CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    acc[c] = 0.f;
#pragma unroll
    for (int j = 0; j < (KNOTS ? 3 : 1); ++j) qacc[j][c] = 0.f;
  }
  int n = 0;   // entries waiting in the list (wave-uniform)

  auto flush = [&](int cnt) {     // cnt is a multiple of 8
    for (int e0 = 0; e0 < cnt; e0 += 8) {
      float pv[8];
      typename VecT<CPL>::T wv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        pv[e] = my_phi[e0 + e];
        wv[e] = *reinterpret_cast<const typename VecT<CPL>::T *>(
            a.dZ + (size_t)((unsigned)my_idx[e0 + e] * (unsigned)H + (unsigned)(CPL * lane)));
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float *f = reinterpret_cast<const float *>(&wv[e]);
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] = fmaf(pv[e], f[c], acc[c]);
        if (KNOTS) {
#pragma unroll
          for (int j = 0; j < NQ; ++j) {
            const float qv = lq[j][wave][e0 + e];
#pragma unroll
            for (int c = 0; c < CPL; ++c) qacc[j][c] = fmaf(qv, f[c], qacc[j][c]);
          }
        }
      }
    }
  };

  // The sorted-observation segments of up to 64 cell columns are fetched by 64 lanes at once and
  // walked as ONE flat candidate list (same order as column by column), so a knot's ~40 candidates
  // cost two dependent memory round trips instead of two per column.
  for (int cxb = cx_lo; cxb <= cx_hi; cxb += 64) {
    const int cxl = cxb + lane;
    int seg0 = 0, seg1 = 0;
    if (cxl <= cx_hi) { seg0 = a.cell_start[cxl * G + cy_lo]; seg1 = a.cell_start[cxl * G + cy_hi + 1]; }
    int incl = seg1 - seg0;                       // inclusive prefix of the segment lengths
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    const int total = __shfl(incl, 63, 64);
    for (int f0 = 0; f0 < total; f0 += 64) {
      const int f = f0 + lane;
      // segment j with incl[j-1] <= f < incl[j]: first lane whose inclusive prefix exceeds f
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
      const int i = ps0 + (f - (pin - pl));
      const int s1 = f < total ? i + 1 : i;       // keeps the `i < s1` form of the validity test below
      float phi = 0.f;
      float q0 = 0.f, q1 = 0.f, q2 = 0.f;
      if (i < s1) {
        if (KNOTS) {
          const float dx = a.xs[i] - kcx, dy = a.ys[i] - kcy;
          const float d = __builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy));
          const float rr = d * ksc;
          phi = basis_eval<BASIS>(rr);
          const float gp = basis_prime_cs<BASIS>(rr);
          const float qd = d > 0.f ? gp * ksc / d : 0.f;      // cdist's backward: no pull at zero distance
          q0 = -qd * dx; q1 = -qd * dy; q2 = -gp * rr;
        } else {
          phi = phi_eval<BASIS>(a.xs[i], a.ys[i], kcx, kcy, ksc);
        }
      }
      const uint64_t mask = __ballot(phi != 0.f);
      const int m = __popcll(mask);
      if (n + m > BW_LIST) {      // not enough room: flush the full groups of 8, keep the remainder
        __builtin_amdgcn_wave_barrier();
        const int full = n & ~7;
        flush(full);
        __builtin_amdgcn_wave_barrier();
        const int rem = n - full;
        float tp = 0.f; int ti = 0;
        float tq[3] = {0.f, 0.f, 0.f};
        if (lane < rem) {
          tp = my_phi[full + lane]; ti = my_idx[full + lane];
          if (KNOTS) { tq[0] = lq[0][wave][full + lane]; tq[1] = lq[1][wave][full + lane]; tq[2] = lq[2][wave][full + lane]; }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < rem) {
          my_phi[lane] = tp; my_idx[lane] = ti;
          if (KNOTS) { lq[0][wave][lane] = tq[0]; lq[1][wave][lane] = tq[1]; lq[2][wave][lane] = tq[2]; }
        }
        n = rem;
      }
      if (phi != 0.f) {
        const int pos = n + __popcll(mask & below);
        my_phi[pos] = phi;
        my_idx[pos] = i;
        if (KNOTS) { lq[0][wave][pos] = q0; lq[1][wave][pos] = q1; lq[2][wave][pos] = q2; }
      }
      n += m;
    }
  }
  // final flush, zero-padded to a multiple of 8 (row 0 of dZ is a valid address)
  const int npad = (n + 7) & ~7;
  if (lane < npad - n) {
    my_phi[n + lane] = 0.f; my_idx[n + lane] = 0;
    if (KNOTS) { lq[0][wave][n + lane] = 0.f; lq[1][wave][n + lane] = 0.f; lq[2][wave][n + lane] = 0.f; }
  }
  __builtin_amdgcn_wave_barrier();
  flush(npad);
  if (KNOTS) {
    const typename VecT<CPL>::T wk = *reinterpret_cast<const typename VecT<CPL>::T *>(a.W0T + (size_t)(a.g.p + k) * H + CPL * lane);
    const float *wf = reinterpret_cast<const float *>(&wk);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) s = fmaf(qacc[j][c], wf[c], s);
      s = wave_sum(s);
      if (lane == 0) a.kpart[(size_t)j * a.g.Ks + k] = s;
    }
  }

  typename VecT<CPL>::T o;
  float *f = reinterpret_cast<float *>(&o);
#pragma unroll
  for (int c = 0; c < CPL; ++c) f[c] = acc[c];
  *reinterpret_cast<typename VecT<CPL>::T *>(a.dW0T + (size_t)(a.g.p + k) * H + CPL * lane) = o;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < CPL; ++c) sq = fmaf(acc[c], acc[c], sq);
  return sq;
}

// ---------------------------------------------------------------------------------------------------------
// NK = 2 or 4 neighbouring knots per wave (fixed grid knots): knots (ix, iy), (ix, iy + 1) [and (ix + 1, iy),
// (ix + 1, iy + 1)] of a level see almost the same observations (supports of 5 spacings, one spacing apart), so
// the candidates are walked once over the union of the supports, a dZ row is fetched once and feeds all NK
// accumulators.  Every knot still sums its own non-zero observations in sorted order and fmaf(0, dz, acc) == acc,
// so the rows of dW0^T are bit-identical to the one-knot-per-wave body.  Groups: level by level, ceil(side / 2)
// pairs per grid row (NK = 2) or ceil(side / 2)^2 blocks of 2 x 2 (NK = 4); knots past the edge of an odd grid
// are absent.  knot_group_count() in window.hip counts the groups for the launch.
// Returns this lane's share of the squares of the rows its wave wrote, as l1_window_bwd_body.
template <int CPL, int BASIS, int NK>
__device__ __forceinline__ float l1_window_bwd_multi_body(const L1BwdArgs &a, const int block) {
  constexpr int H = 64 * CPL;
  constexpr int NX = NK / 2;                          // knots along ix
  __shared__ float lphi[BW_T / 64][NK][BW_LIST + 8];
  __shared__ int lidx[BW_T / 64][BW_LIST + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int l = 0, ix, iy;
  if (NX == 1 && a.xcd_slots > 0) {
    // XCD-aware order: workgroup `block` runs on XCD block & 7 (round-robin dispatch; the launch pads the
    // blocks before the knot groups to a multiple of 8).  XCD x owns the grid rows [x side/8, (x+1) side/8) of
    // every level: its knots see the observations of one stripe of the domain (+ halo), a contiguous eighth of
    // the cell-sorted dZ rows, so each XCD's L2 fetches about 1/5 of dZ_0 instead of all of it.
    const int x = block & 7;
    int q = (block >> 3) * (BW_T / 64) + wave;        // group index inside XCD x's list, level by level
    int r0 = 0;
    for (; l < a.g.n_levels; ++l) {
      const int side = a.g.side[l], hp = (side + 1) >> 1;
      r0 = (x * side) >> 3;
      const int np = ((((x + 1) * side) >> 3) - r0) * hp;
      if (q < np) break;
      q -= np;
    }
    if (l >= a.g.n_levels) return 0.f;
    const int hp = (a.g.side[l] + 1) >> 1;
    const int gx = q / hp;
    ix = r0 + gx; iy = 2 * (q - gx * hp);
  } else {
    int q = block * (BW_T / 64) + wave;               // group index, level by level
    for (; l < a.g.n_levels; ++l) {
      const int hp = (a.g.side[l] + 1) >> 1;
      const int np = (NX == 2 ? hp : a.g.side[l]) * hp;
      if (q < np) break;
      q -= np;
    }
    if (l >= a.g.n_levels) return 0.f;
    const int hp = (a.g.side[l] + 1) >> 1;
    const int gx = q / hp;
    ix = NX * gx; iy = 2 * (q - gx * hp);
  }
  const int side = a.g.side[l];
  int kk[NK];
  bool has[NK];
  float kx[NK], ky[NK], ksc[NK];
  float xlo = 3.0e38f, xhi = -3.0e38f, ylo = 3.0e38f, yhi = -3.0e38f;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const int dx = j >> 1, dy = j & 1;
    has[j] = ix + dx < side && iy + dy < side;
    kk[j] = a.g.off[l] + (has[j] ? (ix + dx) * side + iy + dy : ix * side + iy);
    kx[j] = a.g.centers[2 * kk[j]]; ky[j] = a.g.centers[2 * kk[j] + 1];
    const float bw = a.g.bw[kk[j]];
    ksc[j] = knot_scale(bw, a.g.cal);
    const float r = bw * a.g.cal;                     // support radius
    xlo = fminf(xlo, kx[j] - r); xhi = fmaxf(xhi, kx[j] + r);
    ylo = fminf(ylo, ky[j] - r); yhi = fmaxf(yhi, ky[j] + r);
  }
  int *my_idx = lidx[wave];
  const int G = a.G;
  // cells overlapping any of the support squares, one cell of margin against rounding
  const int cx_lo = max(floor_clamp(xlo * (float)G, G) - 1, 0);
  const int cx_hi = min(floor_clamp(xhi * (float)G, G) + 1, G - 1);
  const int cy_lo = max(floor_clamp(ylo * (float)G, G) - 1, 0);
  const int cy_hi = min(floor_clamp(yhi * (float)G, G) + 1, G - 1);
  const uint64_t below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));

  float acc[NK][CPL];
#pragma unroll
  for (int j = 0; j < NK; ++j)
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[j][c] = 0.f;
  int n = 0;

  auto flush = [&](int cnt) {     // cnt is a multiple of 8
    for (int e0 = 0; e0 < cnt; e0 += 8) {
      typename VecT<CPL>::T wv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        wv[e] = *reinterpret_cast<const typename VecT<CPL>::T *>(
            a.dZ + (size_t)((unsigned)my_idx[e0 + e] * (unsigned)H + (unsigned)(CPL * lane)));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float *f = reinterpret_cast<const float *>(&wv[e]);
#pragma unroll
        for (int j = 0; j < NK; ++j) {
          const float pv = lphi[wave][j][e0 + e];
#pragma unroll
          for (int c = 0; c < CPL; ++c) acc[j][c] = fmaf(pv, f[c], acc[j][c]);
        }
      }
    }
  };

  for (int cxb = cx_lo; cxb <= cx_hi; cxb += 64) {
    const int cxl = cxb + lane;
    int seg0 = 0, seg1 = 0;
    if (cxl <= cx_hi) { seg0 = a.cell_start[cxl * G + cy_lo]; seg1 = a.cell_start[cxl * G + cy_hi + 1]; }
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
      const int i = ps0 + (f - (pin - pl));
      float pv[NK];
      bool any = false;
#pragma unroll
      for (int j = 0; j < NK; ++j) pv[j] = 0.f;
      if (f < total) {
        const float x = a.xs[i], y = a.ys[i];
#pragma unroll
        for (int j = 0; j < NK; ++j) {
          pv[j] = has[j] ? phi_eval<BASIS>(x, y, kx[j], ky[j], ksc[j]) : 0.f;
          any = any || pv[j] != 0.f;
        }
      }
      const uint64_t mask = __ballot(any);
      const int m = __popcll(mask);
      if (n + m > BW_LIST) {      // not enough room: flush the full groups of 8, keep the remainder
        __builtin_amdgcn_wave_barrier();
        const int full = n & ~7;
        flush(full);
        __builtin_amdgcn_wave_barrier();
        const int rem = n - full;
        float tp[NK]; int ti = 0;
#pragma unroll
        for (int j = 0; j < NK; ++j) tp[j] = 0.f;
        if (lane < rem) {
#pragma unroll
          for (int j = 0; j < NK; ++j) tp[j] = lphi[wave][j][full + lane];
          ti = my_idx[full + lane];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < rem) {
#pragma unroll
          for (int j = 0; j < NK; ++j) lphi[wave][j][lane] = tp[j];
          my_idx[lane] = ti;
        }
        n = rem;
      }
      if (any) {
        const int pos = n + __popcll(mask & below);
#pragma unroll
        for (int j = 0; j < NK; ++j) lphi[wave][j][pos] = pv[j];
        my_idx[pos] = i;
      }
      n += m;
    }
  }
  const int npad = (n + 7) & ~7;
  if (lane < npad - n) {
#pragma unroll
    for (int j = 0; j < NK; ++j) lphi[wave][j][n + lane] = 0.f;
    my_idx[n + lane] = 0;
  }
  __builtin_amdgcn_wave_barrier();
  flush(npad);

  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    if (has[j]) {
      typename VecT<CPL>::T o;
      float *fo = reinterpret_cast<float *>(&o);
#pragma unroll
      for (int c = 0; c < CPL; ++c) { fo[c] = acc[j][c]; sq = fmaf(acc[j][c], acc[j][c], sq); }
      *reinterpret_cast<typename VecT<CPL>::T *>(a.dW0T + (size_t)(a.g.p + kk[j]) * H + CPL * lane) = o;
    }
  }
  return sq;
}

}  // namespace stdadk
