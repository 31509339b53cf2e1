// Whole binning of a small batch by SMALL_WG independent 1024-thread workgroups (see bin_small_kernel in window.hip),
// as a body that another launch can carry: the optimiser launch of a step bins the NEXT batch beside its own work
// (optim.hip: adamw_bin_kernel), so that the batch preparation needs neither a launch of its own on the step's critical
// path nor a side stream with its two cross-stream packets per step.
#pragma once
#include "common.h"
#include "basis.h"

namespace stdadk {

constexpr int SMALL_B = 8192, SMALL_G = 64, SMALL_WG = 8;
// dynamic LDS of a binning workgroup for a batch of B rows, in ints: counters + scan of the cells, the two permutations
// (sized by the batch: 68 KiB at 4 096 rows -- two workgroups per CU -- 100 KiB at 8 192), 1024 scan partials
__host__ __device__ constexpr int bin_small_cap(int B) { return (B + 1023) & ~1023; }
// up to 4 096 rows the row numbers idx[b] are kept in LDS too (int64: + 32 KiB), so that the emission does not fetch
// them again behind the sort -- one dependent global round trip less, which counts when the binning runs beside the
// optimiser's HBM stream (adamw_bin_kernel)
__host__ __device__ constexpr bool bin_small_stages_rows(int B) { return bin_small_cap(B) <= 4096; }
__host__ __device__ constexpr int bin_small_lds_ints(int B) {
  return 2 * SMALL_G * SMALL_G + 4 + 2 * bin_small_cap(B) + 1024 + (bin_small_stages_rows(B) ? 2 * bin_small_cap(B) : 0);
}
constexpr int BIN_SMALL_LDS_INTS = 2 * SMALL_G * SMALL_G + 4 + 2 * SMALL_B + 1024;      // the most (B = SMALL_B: 100 KiB)
static_assert(bin_small_lds_ints(4096) <= BIN_SMALL_LDS_INTS, "the staged row numbers must fit the largest image");

struct BinSmallArgs {
  const int64_t *idx;               // rows of the resident arrays, or NULL (rows 0..B-1)
  const float *coords, *t, *y, *X;  // resident arrays ([.,2], [.], [.,Q], [.,p]); t / y / X may be NULL
  int Q, p, B, G;
  int *keys, *cell_start, *perm;    // outputs (BinBuffers of window.h)
  float *xs, *ys, *ts, *y_s, *X_s;  // y_s / X_s NULL: not carried
};

__device__ __forceinline__ int bin_cell_of(float x, float y, int G) {
  int cx = floor_clamp(x * (float)G, G);
  int cy = floor_clamp(y * (float)G, G);
  return cx * G + cy;
}

// workgroup `block` of `nblocks` (1024 threads); smem: BIN_SMALL_LDS_INTS ints
__device__ __forceinline__ void bin_small_body(const BinSmallArgs &a, const int block, const int nblocks, int *smem) {
  int *hist = smem;                                   // [SMALL_G^2] counts, then running cursors
  int *start = hist + SMALL_G * SMALL_G;              // [SMALL_G^2 + 1] (+ 3 of padding)
  int *ptmp = start + SMALL_G * SMALL_G + 4;          // [cap(B)] unordered permutation
  int *pfin = ptmp + bin_small_cap(a.B);              // [cap(B)] ordered permutation
  int *part = pfin + bin_small_cap(a.B);              // [1024]
  const bool stage = bin_small_stages_rows(a.B) && a.idx != nullptr;
  long long *rows = reinterpret_cast<long long *>(part + 1024);   // [cap(B)] idx[b], when staged
  const int64_t *__restrict__ idx = a.idx;
  const float *__restrict__ coords = a.coords, *__restrict__ t = a.t, *__restrict__ y = a.y, *__restrict__ X = a.X;
  const int Q = a.Q, p = a.p, B = a.B, G = a.G;
  int *__restrict__ keys = a.keys, *__restrict__ cell_start = a.cell_start, *__restrict__ perm = a.perm;
  float *__restrict__ xs = a.xs, *__restrict__ ys = a.ys, *__restrict__ ts = a.ts, *__restrict__ y_s = a.y_s,
        *__restrict__ X_s = a.X_s;
  const int tid = threadIdx.x;
  const int ncell = G * G;
  constexpr int PER_T = SMALL_B / 1024;        // observations per thread
  // this workgroup's slice [lo, hi) of the batch positions (keys) and of the sorted positions (everything else)
  const int per_wg = ((B + nblocks - 1) / nblocks + 63) & ~63;
  const int lo = min(block * per_wg, B), hi = min(lo + per_wg, B);
  for (int c = tid; c < ncell; c += 1024) hist[c] = 0;
  __syncthreads();
  int kk[PER_T];
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    const int b = tid + 1024 * i;
    const int bc = min(b, B - 1);
    const int64_t r = idx ? idx[bc] : bc;
    if (stage && b < B) rows[b] = r;
    kk[i] = bin_cell_of(coords[2 * r], coords[2 * r + 1], G);       // unconditional, clamped
  }
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    const int b = tid + 1024 * i;
    if (b < B) {
      if (b >= lo && b < hi) keys[b] = kk[i];
      atomicAdd(&hist[kk[i]], 1);
    }
  }
  __syncthreads();
  // exclusive scan: thread tid owns cells [tid*per, tid*per+per); wave-level shuffles, two barriers
  const int per = (ncell + 1023) / 1024;
  const int i0 = tid * per, i1 = min(i0 + per, ncell);
  int s = 0;
  for (int i = i0; i < i1; ++i) s += hist[i];
  const int lane = tid & 63, wv = tid >> 6;
  int incl = s;                                  // inclusive scan inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) part[wv] = incl;               // wave totals
  __syncthreads();
  if (tid < 16) {                                // 16 waves: scan their totals in one wave
    int tot = part[tid];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int v = __shfl_up(tot, o, 64);
      if (tid >= o) tot += v;
    }
    part[16 + tid] = tot;                        // inclusive totals
  }
  __syncthreads();
  int run = incl - s + (wv > 0 ? part[16 + wv - 1] : 0);
  for (int i = i0; i < i1; ++i) {
    const int cnt = hist[i];
    start[i] = run;
    if (block == 0) cell_start[i] = run;
    hist[i] = run;            // cursor for the scatter
    run += cnt;
  }
  if (tid == 1023) {
    start[ncell] = part[31];
    if (block == 0) cell_start[ncell] = part[31];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    const int b = tid + 1024 * i;
    if (b < B) ptmp[atomicAdd(&hist[kk[i]], 1)] = b;
  }
  __syncthreads();
  // order every cell that reaches into [lo, hi) by original index (rank by counting): LDS only
  for (int c = tid; c < ncell; c += 1024) {
    const int s0 = start[c], s1 = start[c + 1];
    if (s1 <= lo || s0 >= hi) continue;
    for (int i = s0; i < s1; ++i) {
      const int b = ptmp[i];
      int rank = 0;
      for (int j = s0; j < s1; ++j) rank += ptmp[j] < b;
      pfin[s0 + rank] = b;
    }
  }
  __syncthreads();
  // emit the sorted arrays: one position per thread and pass, independent loads
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    const int pos = lo + tid + 1024 * i;
    if (lo + 1024 * i >= hi) break;              // workgroup-uniform
    const int b = pfin[min(pos, hi - 1)];
    const int64_t r = stage ? (int64_t)rows[b] : (idx ? idx[b] : b);
    const float cx = coords[2 * r], cy = coords[2 * r + 1];
    const float tv = t ? t[r] : 0.f;
    if (pos < hi) {
      perm[pos] = b;
      xs[pos] = cx;
      ys[pos] = cy;
      if (t) ts[pos] = tv;
      if (y_s)
        for (int q = 0; q < Q; ++q) y_s[(int64_t)pos * Q + q] = y[r * Q + q];
      if (X_s)
        for (int q = 0; q < p; ++q) X_s[(int64_t)pos * p + q] = X[r * p + q];
    }
  }
}

}  // namespace stdadk
