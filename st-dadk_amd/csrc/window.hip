// Index-window path of the first layer (compact-support bases on uniform knot grids).
//
// A Wendland / triangular knot with bandwidth 2.5 spacings is non-zero for at most ~21 knots per
// level around an observation (st_interp.py:152-185 grid, :470-471 support r < 1), so
// z0 = [X | phi | psi] W0 touches ~63 + Kt rows of W0^T instead of D.  Observations are binned
// into a G x G cell grid (counting sort, row-major cells) so that
//   forward : one wave per observation gathers its W0^T rows (1 KiB coalesced each), neighbours in
//             the sorted order re-use rows from L2; LayerNorm/ReLU/Dropout fused behind it;
//   backward: every knot row of dW0^T is OWNED by one wave that walks the observations of the
//             cells overlapping the knot's support square and gathers their dZ rows — no atomics,
//             fixed summation order.
// Integer bookkeeping (cell keys, sorted permutation, window origins) follows the bit-exact contract
// written in include/stdadk.h (pinned by tests); phi uses the same arithmetic as rbf_build.hip.
#include "window.h"
#include "bin_body.h"

#include "basis.h"
#include "l1_body.h"
#include "l1_bwd_body.h"

namespace stdadk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// observation binning
// ---------------------------------------------------------------------------------------------
int pick_cell_grid(int64_t B) {
  int G = 8;
  while (G < 256 && (int64_t)G * G < B) G <<= 1;   // ~1 observation per cell
  return G;
}

__device__ __forceinline__ int cell_of(float x, float y, int G) {
  int cx = floor_clamp(x * (float)G, G);
  int cy = floor_clamp(y * (float)G, G);
  return cx * G + cy;
}

// `idx` (optional): the batch is rows idx[b] of the resident arrays; perm / keys stay batch positions
__global__ void cell_hist_kernel(const float *__restrict__ coords, const int64_t *__restrict__ idx, int B,
                                 int G, int *__restrict__ keys, int *__restrict__ hist) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int64_t r = idx ? idx[b] : b;
  int c = cell_of(coords[2 * r], coords[2 * r + 1], G);
  keys[b] = c;
  atomicAdd(&hist[c], 1);
}

// exclusive scan of n counters by one 1024-thread workgroup: tiles of 4096 counters, every thread
// one int4 (coalesced), wave-level shuffles + a 16-entry scan of the wave totals, running carry
__global__ __launch_bounds__(1024) void cell_scan_kernel(const int *__restrict__ hist, int n,
                                                         int *__restrict__ cell_start,
                                                         int *__restrict__ cursor, int vec) {
  __shared__ int part[2][32];
  const bool scalar_only = !vec;                   // buffers not 16-byte aligned: scalar accesses only
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int carry = 0;
  const int ntile = (n + 4095) / 4096;
  int i = 4 * tid;
  int4 v = make_int4(0, 0, 0, 0);
  if (i + 3 < n && !scalar_only) v = *reinterpret_cast<const int4 *>(hist + i);
  else if (i + 3 < n) v = make_int4(hist[i], hist[i + 1], hist[i + 2], hist[i + 3]);
  else { if (i < n) v.x = hist[i]; if (i + 1 < n) v.y = hist[i + 1]; if (i + 2 < n) v.z = hist[i + 2]; }
  for (int tile = 0; tile < ntile; ++tile) {
    // next tile's counters in flight while this one is scanned
    const int in = 4096 * (tile + 1) + 4 * tid;
    int4 vn = make_int4(0, 0, 0, 0);
    if (tile + 1 < ntile) {
      if (in + 3 < n && !scalar_only) vn = *reinterpret_cast<const int4 *>(hist + in);
      else if (in + 3 < n) vn = make_int4(hist[in], hist[in + 1], hist[in + 2], hist[in + 3]);
      else { if (in < n) vn.x = hist[in]; if (in + 1 < n) vn.y = hist[in + 1]; if (in + 2 < n) vn.z = hist[in + 2]; }
    }
    const int s = v.x + v.y + v.z + v.w;
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    int *pp = part[tile & 1];                      // double-buffered: one barrier per tile
    if (lane == 63) pp[wv] = incl;
    __syncthreads();
    int tot = pp[lane & 15];                       // every wave scans the 16 wave totals itself
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int u = __shfl_up(tot, o, 64);
      if ((lane & 15) >= o) tot += u;
    }
    const int before = wv > 0 ? __shfl(tot, wv - 1, 64) : 0;
    const int total = __shfl(tot, 15, 64);
    int run = carry + before + incl - s;
    i = 4096 * tile + 4 * tid;
    const int4 o4 = make_int4(run, run + v.x, run + v.x + v.y, run + v.x + v.y + v.z);
    if (i + 3 < n && !scalar_only) {
      *reinterpret_cast<int4 *>(cell_start + i) = o4;
      *reinterpret_cast<int4 *>(cursor + i) = o4;
    } else if (i + 3 < n) {
      cell_start[i] = o4.x; cell_start[i + 1] = o4.y; cell_start[i + 2] = o4.z; cell_start[i + 3] = o4.w;
      cursor[i] = o4.x; cursor[i + 1] = o4.y; cursor[i + 2] = o4.z; cursor[i + 3] = o4.w;
    } else {
      if (i < n) { cell_start[i] = o4.x; cursor[i] = o4.x; }
      if (i + 1 < n) { cell_start[i + 1] = o4.y; cursor[i + 1] = o4.y; }
      if (i + 2 < n) { cell_start[i + 2] = o4.z; cursor[i + 2] = o4.z; }
    }
    carry += total;
    v = vn;
  }
  if (tid == 0) cell_start[n] = carry;
}

__global__ void cell_scatter_kernel(const int *__restrict__ keys, int B, int *__restrict__ cursor,
                                    int *__restrict__ perm_tmp) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int pos = atomicAdd(&cursor[keys[b]], 1);
  perm_tmp[pos] = b;
}

// One thread per cell: order the cell's observations by original index (rank by counting, which
// makes the permutation independent of the atomics' arrival order) and emit the sorted arrays.
__global__ void cell_order_kernel(const int *__restrict__ cell_start, int ncell,
                                  const int *__restrict__ perm_tmp, int *__restrict__ perm,
                                  const int64_t *__restrict__ idx,
                                  const float *__restrict__ coords, const float *__restrict__ t,
                                  const float *__restrict__ y, int Q, const float *__restrict__ X, int p,
                                  float *__restrict__ xs, float *__restrict__ ys, float *__restrict__ ts,
                                  float *__restrict__ y_s, float *__restrict__ X_s) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const int s0 = cell_start[c], s1 = cell_start[c + 1];
  for (int i = s0; i < s1; ++i) {
    const int b = perm_tmp[i];
    int rank = 0;
    for (int j = s0; j < s1; ++j) rank += perm_tmp[j] < b;
    const int pos = s0 + rank;
    const int64_t r = idx ? idx[b] : b;
    perm[pos] = b;
    xs[pos] = coords[2 * r];
    ys[pos] = coords[2 * r + 1];
    if (t) ts[pos] = t[r];
    if (y_s)
      for (int q = 0; q < Q; ++q) y_s[(int64_t)pos * Q + q] = y[r * Q + q];
    if (X_s)
      for (int q = 0; q < p; ++q) X_s[(int64_t)pos * p + q] = X[r * p + q];
  }
}

__global__ void zero_ints_kernel(int *__restrict__ p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

// Whole binning in ONE launch for small batches (B <= SMALL_B, G <= SMALL_G): histogram, scan, scatter and
// in-cell ordering with the counters and the unordered permutation in LDS.  Same outputs, bit for bit, as
// the multi-kernel path.  SMALL_WG workgroups share the work without talking to each other: every one builds
// the complete histogram / scan / unordered permutation in its OWN LDS (8 192 keys are 64 KB of L2 reads), then
// orders only the cells that reach into its slice of the sorted positions and emits only that slice (the
// in-cell order is by original index, so the workgroups' different atomic arrival orders do not show).  What is
// split -- the ordering loops and the dependent row gathers of the emission -- is the longer half of the
// single-workgroup kernel's latency chain (MI355X, B = 4096: 21 us with one workgroup).
__global__ __launch_bounds__(1024) void bin_small_kernel(BinSmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) int bin_smem[];
  bin_small_body(a, (int)blockIdx.x, (int)gridDim.x, bin_smem);
}

BinSmallArgs bin_small_args(const float *coords, const float *t, const float *y, int Q, const float *X, int p, int B,
                            int G, const BinBuffers &bb, const int64_t *idx) {
  BinSmallArgs a;
  a.idx = idx; a.coords = coords; a.t = t; a.y = y; a.X = X; a.Q = Q; a.p = p; a.B = B; a.G = G;
  a.keys = bb.keys; a.cell_start = bb.cell_start; a.perm = bb.perm;
  a.xs = bb.xs; a.ys = bb.ys; a.ts = bb.ts;
  a.y_s = y ? bb.y_s : nullptr;
  a.X_s = (X && p > 0) ? bb.X_s : nullptr;
  return a;
}
bool bin_small_eligible(int B, int G) { return B <= SMALL_B && G <= SMALL_G; }

int bin_obs(const float *coords, const float *t, const float *y, int Q, const float *X, int p, int B,
            int G, const BinBuffers &bb, hipStream_t st, const int64_t *idx, bool many_small) {
  const int ncell = G * G;
  if (B <= SMALL_B && G <= SMALL_G && !many_small) {
    // (a few hundred rows are not worth splitting: one workgroup)
    BinSmallArgs ba = bin_small_args(coords, t, y, Q, X, p, B, G, bb, idx);
    static bool attr = false;       // > 64 KiB of dynamic LDS: raised on the first (eager) call
    if (!attr) {
      hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void *>(bin_small_kernel), BIN_SMALL_LDS_INTS * (int)sizeof(int));
      if (e != hipSuccess) { set_error("bin_obs: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
      attr = true;
    }
    STDADK_LAUNCH(bin_small_kernel, dim3(B >= 1024 ? SMALL_WG : 1), dim3(1024), BIN_SMALL_LDS_INTS * sizeof(int), st, ba);
    STDADK_CHECK_LAUNCH("bin_obs");
    return 0;
  }
  // a kernel, not hipMemsetAsync: the step must stay a pure chain of kernel nodes under capture
  STDADK_LAUNCH(zero_ints_kernel, dim3((unsigned)ceil_div(ncell, 256)), dim3(256), 0, st, bb.hist, ncell);
  const unsigned nb = (unsigned)ceil_div(B, 256);
  STDADK_LAUNCH(cell_hist_kernel, dim3(nb), dim3(256), 0, st, coords, idx, B, G, bb.keys, bb.hist);
  STDADK_LAUNCH(cell_scan_kernel, dim3(1), dim3(1024), 0, st, bb.hist, ncell, bb.cell_start, bb.cursor,
                (aligned16(bb.hist) && aligned16(bb.cell_start) && aligned16(bb.cursor)) ? 1 : 0);
  STDADK_LAUNCH(cell_scatter_kernel, dim3(nb), dim3(256), 0, st, bb.keys, B, bb.cursor, bb.perm_tmp);
  STDADK_LAUNCH(cell_order_kernel, dim3((unsigned)ceil_div(ncell, 128)), dim3(128), 0, st, bb.cell_start,
                ncell, bb.perm_tmp, bb.perm, idx, coords, t, y, Q, X, p, bb.xs, bb.ys, bb.ts,
                y ? bb.y_s : (float *)nullptr, (X && p > 0) ? bb.X_s : (float *)nullptr);
  STDADK_CHECK_LAUNCH("bin_obs");
  return 0;
}

// One launch instead of a handful of framework gathers: rows idx[b] of the resident observation
// arrays into contiguous batch buffers (the batch producer of train_st_interp.py:413-460,609-612).
__global__ void gather_batch_kernel(const float *__restrict__ coords, const float *__restrict__ t,
                                    const float *__restrict__ y, const float *__restrict__ X,
                                    const int64_t *__restrict__ idx, int B, int Q, int p,
                                    float *__restrict__ coords_o, float *__restrict__ t_o,
                                    float *__restrict__ y_o, float *__restrict__ X_o) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int64_t r = idx[b];
  coords_o[2 * b] = coords[2 * r];
  coords_o[2 * b + 1] = coords[2 * r + 1];
  t_o[b] = t[r];
  for (int q = 0; q < Q; ++q) y_o[(int64_t)b * Q + q] = y[r * Q + q];
  for (int q = 0; q < p; ++q) X_o[(int64_t)b * p + q] = X[r * p + q];
}

__global__ void unpermute_kernel(const float *__restrict__ in, const int *__restrict__ perm, int B, int Q,
                                 float *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Q) return;
  int r = i / Q, q = i - r * Q;
  out[(int64_t)perm[r] * Q + q] = in[i];
}

int unpermute_rows(const float *in, const int *perm, int B, int Q, float *out, hipStream_t st) {
  STDADK_LAUNCH(unpermute_kernel, dim3((unsigned)ceil_div((int64_t)B * Q, 256)), dim3(256), 0, st, in,
                     perm, B, Q, out);
  STDADK_CHECK_LAUNCH("unpermute");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// forward:  z0 -> LayerNorm -> ReLU -> Dropout, one wave per observation
// ---------------------------------------------------------------------------------------------
// R: observations per wave (1: l1_window_fwd_body; 2: l1_window_fwd_multi_body, fixed knots only)
template <int CPL, bool LN, int BASIS, bool FREE, int R>
__global__ __launch_bounds__(FW_T) void l1_window_fwd_kernel(L1FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int r0 = l1_chunk_of(blockIdx.x, a.n_wg) * a.rows_per_wg;
  if constexpr (R == 1) l1_window_fwd_body<CPL, LN, BASIS, FREE>(a, smem, r0, min(r0 + a.rows_per_wg, a.B));
  else l1_window_fwd_multi_body<CPL, LN, BASIS, R>(a, smem, r0, min(r0 + a.rows_per_wg, a.B));
}

// HALO_SPLIT workgroups per level: each takes a slice of the level's knots and writes the slice's
// max_k (s_k + |c_k - grid_k|_inf) (side-1) to halo[level][slice]; the consumer (halo_half_width in
// l1_body.h) takes the max of a level's HALO_SPLIT values and rounds it up to the half-width R.
// log_bw (optional): the bandwidths are given as logs; the kernel then also writes exp(log_bw) to bw_out
// (the table the window kernels read), saving the separate exp launch
__global__ __launch_bounds__(256) void knot_halo_kernel(GridView g, float *__restrict__ halo,
                                                        const float *__restrict__ log_bw, float *__restrict__ bw_out) {
  const int l = blockIdx.x, slice = blockIdx.y;
  const int side = g.side[l], off = g.off[l];
  const int nk = side * side;
  const int per = (nk + HALO_SPLIT - 1) / HALO_SPLIT;
  const int j0 = slice * per, j1 = min(j0 + per, nk);
  const float sm1 = (float)(side > 1 ? side - 1 : 1);
  float m = 0.f;
  for (int j = j0 + threadIdx.x; j < j1; j += 256) {
    const int k = off + j;
    const int ix = j / side, iy = j - ix * side;
    const float gx = side > 1 ? (float)ix / sm1 : 0.f, gy = side > 1 ? (float)iy / sm1 : 0.f;
    const float mv = fmaxf(fabsf(g.centers[2 * k] - gx), fabsf(g.centers[2 * k + 1] - gy));
    float bwk;
    if (log_bw) { bwk = expf(log_bw[k]); bw_out[k] = bwk; } else { bwk = g.bw[k]; }
    const float v = (bwk * g.cal + mv) * sm1;
    m = fmaxf(m, (v == v) ? v : 3.0e38f);          // NaN knots widen the window to the whole level
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) halo[l * HALO_SPLIT + slice] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

int knot_halo(const GridView &g, float *halo, hipStream_t st, const float *log_bw, float *bw_out) {
  STDADK_REQUIRE(!log_bw || bw_out, STDADK_E_ARG, "knot_halo: log_bw needs bw_out");
  STDADK_LAUNCH(knot_halo_kernel, dim3((unsigned)g.n_levels, HALO_SPLIT), dim3(256), 0, st, g, halo, log_bw, bw_out);
  STDADK_CHECK_LAUNCH("knot_halo");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// scattered knots: per-level cell lists of the KNOTS (see knot_bins in window.h)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void knot_bin_kernel(GridView g, int *__restrict__ kcs, int *__restrict__ kperm,
                                                        int *__restrict__ kperm_tmp, float *__restrict__ reach,
                                                        const float *__restrict__ log_bw, float *__restrict__ bw_out) {
  constexpr int Gk = KNOT_CELLS, NC = Gk * Gk;
  static_assert(NC == 1024, "one thread per cell");
  __shared__ int hist[NC];          // counts, then running cursors
  __shared__ int start[NC + 1];
  __shared__ int part[32];
  __shared__ float fred[16];
  const int l = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int off = g.off[l], n = g.cnt[l];
  hist[tid] = 0;
  __syncthreads();
  float m = 0.f;
  for (int j = tid; j < n; j += 1024) {
    const int k = off + j;
    float bwk;
    if (log_bw) { bwk = expf(log_bw[k]); bw_out[k] = bwk; } else { bwk = g.bw[k]; }
    const float v = bwk * g.cal;
    m = fmaxf(m, (v == v) ? v : 3.0e38f);                  // a NaN bandwidth reaches everywhere
    atomicAdd(&hist[cell_of(g.centers[2 * k], g.centers[2 * k + 1], Gk)], 1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) fred[wv] = m;
  __syncthreads();
  if (tid == 0) {
    float r = 0.f;
    for (int w = 0; w < 16; ++w) r = fmaxf(r, fred[w]);
    reach[l] = r;
  }
  // exclusive scan of the 1024 cell counts: one per thread
  const int cnt = hist[tid];
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  if (lane == 63) part[wv] = incl;
  __syncthreads();
  if (tid < 16) {
    int tot = part[tid];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int u = __shfl_up(tot, o, 64);
      if (tid >= o) tot += u;
    }
    part[16 + tid] = tot;
  }
  __syncthreads();
  const int s0 = incl - cnt + (wv > 0 ? part[16 + wv - 1] : 0);
  start[tid] = s0;
  if (tid == 1023) start[NC] = s0 + cnt;
  hist[tid] = s0;                                           // cursor
  int *cs = kcs + (size_t)l * (NC + 1);
  cs[tid] = off + s0;
  if (tid == 1023) cs[NC] = off + s0 + cnt;
  __syncthreads();
  for (int j = tid; j < n; j += 1024) {
    const int k = off + j;
    const int pos = atomicAdd(&hist[cell_of(g.centers[2 * k], g.centers[2 * k + 1], Gk)], 1);
    kperm_tmp[off + pos] = k;
  }
  __syncthreads();            // vmcnt(0) + barrier: the workgroup's own global stores are visible to it
  // one thread per cell: rank its knots by id (the scatter's arrival order must not show)
  {
    const int a0 = start[tid], a1 = start[tid + 1];
    for (int i = a0; i < a1; ++i) {
      const int k = kperm_tmp[off + i];
      int rank = 0;
      for (int j = a0; j < a1; ++j) rank += kperm_tmp[off + j] < k;
      kperm[off + a0 + rank] = k;
    }
  }
}

int knot_bins(const GridView &g, int *kcs, int *kperm, int *kperm_tmp, float *reach, hipStream_t st,
              const float *log_bw, float *bw_out) {
  STDADK_REQUIRE(g.scattered && g.n_levels > 0 && kcs && kperm && kperm_tmp && reach, STDADK_E_ARG, "knot_bins: bad arguments");
  STDADK_REQUIRE(!log_bw || bw_out, STDADK_E_ARG, "knot_bins: log_bw needs bw_out");
  STDADK_LAUNCH(knot_bin_kernel, dim3((unsigned)g.n_levels), dim3(1024), 0, st, g, kcs, kperm, kperm_tmp, reach, log_bw, bw_out);
  STDADK_CHECK_LAUNCH("knot_bins");
  return 0;
}

bool l1_window_supported(int n_levels, int basis, int H, int p, int Kt) {
  if (n_levels <= 0 || n_levels > STDADK_MAX_LEVELS) return false;
  if (basis != STDADK_BASIS_WENDLAND && basis != STDADK_BASIS_TRIANGULAR) return false;  // compact support
  if (H != 128 && H != 256) return false;
  if (p > WIN_MAX_P) return false;
  if ((size_t)Kt * H * 4 > 96 * 1024) return false;   // temporal rows of W0^T live in LDS
  return true;
}

template <int CPL, bool LN, int BASIS, bool FREE, int R>
static int launch_fwd_t(const L1FwdArgs &a, hipStream_t st) {
  const int Kt_pad = (a.g.Kt + 3) & ~3;
  size_t lds = ((size_t)a.g.Kt * 64 * CPL + (FW_T / 64) * (LIST * (R + 1) + Kt_pad * R)) * sizeof(float);
  auto kern = l1_window_fwd_kernel<CPL, LN, BASIS, FREE, R>;
  // raise the dynamic-LDS cap when a launch needs more than any before it (never inside a stream
  // capture: the engine runs its first step eagerly)
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), (int)lds);
    if (e != hipSuccess) { set_error("l1_window_forward: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_lds = lds;
  }
  STDADK_LAUNCH_NAMED("l1_window_fwd_kernel", kern, dim3((unsigned)a.n_wg), dim3(FW_T), lds, st, a);
  STDADK_CHECK_LAUNCH("l1_window_forward");
  return 0;
}

// observations per wave: 2 once every wave of a workgroup still gets a full group (measured on MI355X, C2
// model: 4 per wave is no faster than 2 -- the union of four windows needs the registers that keep 8 rows in
// flight -- and 2 per wave is +14..17 % on forward-only calls of 65 536+ rows, +2..5 % on train steps of
// 8 192+).  Environment STDADK_L1_GROUP = 1 | 2 overrides (diagnostic).
static int l1_group(const L1FwdArgs &a) {
  if (a.halo) return 1;
  int r = a.rows_per_wg >= 2 * (FW_T / 64) ? 2 : 1;
  if (const char *e = getenv("STDADK_L1_GROUP")) {
    const int v = atoi(e);
    if (v == 1 || v == 2) r = v;
  }
  return r;
}

template <int CPL, bool LN, int BASIS>
static int launch_fwd(const L1FwdArgs &a, hipStream_t st) {
  if (a.halo || a.kperm) return launch_fwd_t<CPL, LN, BASIS, true, 1>(a, st);
  return l1_group(a) == 2 ? launch_fwd_t<CPL, LN, BASIS, false, 2>(a, st)
                          : launch_fwd_t<CPL, LN, BASIS, false, 1>(a, st);
}

int l1_window_forward(const L1FwdArgs &a_in, int basis, bool ln, hipStream_t st) {
  L1FwdArgs a = a_in;
  STDADK_REQUIRE((int64_t)(a.g.p + a.g.Ks + a.g.Kt) * a.H < (1ll << 32), STDADK_E_ARG,
                 "l1_window_forward: D*H exceeds 32-bit offsets");
  // one 16-wave workgroup per CU (~95 KiB LDS), multiple of 8 for the XCD mapping
  int n_wg = 256;
  while (n_wg > 8 && (int64_t)(n_wg / 2) * (FW_T / 64) >= a.B) n_wg >>= 1;
  a.n_wg = n_wg;
  a.rows_per_wg = (int)ceil_div(a.B, n_wg);
#define GO(CPL_) \
  (basis == STDADK_BASIS_WENDLAND ? (ln ? launch_fwd<CPL_, true, 0>(a, st) : launch_fwd<CPL_, false, 0>(a, st)) \
                                  : (ln ? launch_fwd<CPL_, true, 2>(a, st) : launch_fwd<CPL_, false, 2>(a, st)))
  if (a.H == 256) return GO(4);
  if (a.H == 128) return GO(2);
#undef GO
  set_error("l1_window_forward: H=%d unsupported", a.H);
  return STDADK_E_SHAPE;
}

// ---------------------------------------------------------------------------------------------
// backward: dW0T[p+k, :] = sum_b phi[b,k] dZ[b,:]   — one wave OWNS one knot row
// ---------------------------------------------------------------------------------------------
// The mirror image of the forward gather: a knot's support square overlaps a few cell rows of the
// binning grid; each cell row is one contiguous range of sorted observations.  64 lanes evaluate
// phi for 64 candidate observations at once, the non-zero ones are compacted into a per-wave list
// and their dZ rows (1 KiB, coalesced) are accumulated 8 loads at a time.  No atomics, summation in
// sorted-observation order => bitwise reproducible.  Coarse levels (most observations per knot) are
// scheduled first.
template <int CPL, int BASIS, bool KNOTS>
__global__ __launch_bounds__(BW_T) void l1_window_bwd_kernel(L1BwdArgs a) {
  l1_window_bwd_body<CPL, BASIS, KNOTS>(a, (int)blockIdx.x);
}

template <int CPL, int BASIS, int NK>
__global__ __launch_bounds__(BW_T) void l1_window_bwd_multi_kernel(L1BwdArgs a) {
  l1_window_bwd_multi_body<CPL, BASIS, NK>(a, (int)blockIdx.x);
}

// groups of the per-knot gather with nk knots per wave: level by level, ceil(side / 2) pairs per grid row
// (nk = 2) or ceil(side / 2)^2 blocks of 2 x 2 (nk = 4)
int knot_group_count(const GridView &g, int nk) {
  if (nk <= 1) return g.Ks;
  int n = 0;
  for (int l = 0; l < g.n_levels; ++l) {
    const int hp = (g.side[l] + 1) / 2;
    n += (nk == 4 ? hp : g.side[l]) * hp;
  }
  return n;
}

// XCD-striped order of the knot groups (nk = 1 or 2 knots per wave): workgroups per XCD = the longest of the eight lists (equal when every
// side is a multiple of 8); environment STDADK_KNOT_XCD=0 switches the striping off (diagnostic)
int knot_xcd_slots(const GridView &g, int nk) {
  const char *e = getenv("STDADK_KNOT_XCD");
  if ((e && e[0] == '0') || (nk != 1 && nk != 2) || g.scattered) return 0;
  int most = 0;
  for (int x = 0; x < 8; ++x) {
    int n = 0;
    for (int l = 0; l < g.n_levels; ++l) {
      const int side = g.side[l];
      n += ((((x + 1) * side) >> 3) - ((x * side) >> 3)) * (nk == 2 ? (side + 1) / 2 : side);
    }
    most = n > most ? n : most;
  }
  return (int)ceil_div(most, BW_T / 64);
}

// knots per wave of the per-knot gather: learnable knots 1; fixed grid knots 2 (environment
// STDADK_KNOTS_PER_WAVE = 1 | 2 overrides, diagnostic).  Measured on MI355X, C2 model: 2 per wave is
// -2.5 us on the merged weight-gradient launch at B = 4096 and -14 % on it at B = 65 536; blocks of 2 x 2
// (the body supports NK = 4) are slower than pairs up to B = 16 384 and no faster at 65 536.
int knots_per_wave(const L1BwdArgs &a) {
  if (a.kpart || a.g.scattered) return 1;
  int nk = 2;
  if (const char *e = getenv("STDADK_KNOTS_PER_WAVE")) {
    const int v = atoi(e);
    if (v == 1 || v == 2) nk = v;
  }
  return nk;
}

int l1_window_backward(L1BwdArgs a, int basis, hipStream_t st) {
  STDADK_REQUIRE(a.G <= 256, STDADK_E_ARG, "l1_window_backward: G too large");
  STDADK_REQUIRE((int64_t)a.B * a.H < (1ll << 32), STDADK_E_ARG, "l1_window_backward: B*H exceeds 32-bit offsets");
  const int nk = knots_per_wave(a);
  a.xcd_slots = knot_xcd_slots(a.g, nk);
  const unsigned grid = a.xcd_slots > 0 ? 8u * (unsigned)a.xcd_slots
                                        : (unsigned)ceil_div(knot_group_count(a.g, nk), BW_T / 64);
  STDADK_REQUIRE(!a.kpart || a.W0T, STDADK_E_ARG, "l1_window_backward: knot sums need W0^T");
#define GO(CPL_, BS_)                                                                                        \
  do {                                                                                                       \
    if (a.kpart) STDADK_LAUNCH_NAMED("l1_window_bwd_kernel<knots>", (l1_window_bwd_kernel<CPL_, BS_, true>), \
                                     dim3(grid), dim3(BW_T), 0, st, a);                                      \
    else if (nk == 2) STDADK_LAUNCH_NAMED("l1_window_bwd_kernel", (l1_window_bwd_multi_kernel<CPL_, BS_, 2>), \
                                          dim3(grid), dim3(BW_T), 0, st, a);                                 \
    else STDADK_LAUNCH_NAMED("l1_window_bwd_kernel", (l1_window_bwd_kernel<CPL_, BS_, false>), dim3(grid),   \
                             dim3(BW_T), 0, st, a);                                                          \
  } while (0)
  if (a.H == 256) { if (basis == STDADK_BASIS_WENDLAND) GO(4, 0); else GO(4, 2); }
  else if (a.H == 128) { if (basis == STDADK_BASIS_WENDLAND) GO(2, 0); else GO(2, 2); }
  else { set_error("l1_window_backward: H=%d unsupported", a.H); return STDADK_E_SHAPE; }
#undef GO
  STDADK_CHECK_LAUNCH("l1_window_backward");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// diagnostic: window origins (bit-exact contract of include/stdadk.h)
// ---------------------------------------------------------------------------------------------
struct LevelTable { int side[STDADK_MAX_LEVELS]; int off[STDADK_MAX_LEVELS]; };

__global__ void knot_windows_kernel(const float *__restrict__ coords, int B, int L, LevelTable lt, int p,
                                    int *__restrict__ ix0, int *__restrict__ iy0, int *__restrict__ col0) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * L) return;
  int b = i / L, l = i - b * L;
  int side = lt.side[l], off = lt.off[l];
  int win = side < WIN ? side : WIN;
  int x0 = window_start(coords[2 * b], side, win), y0 = window_start(coords[2 * b + 1], side, win);
  ix0[i] = x0;
  iy0[i] = y0;
  col0[i] = p + off + x0 * side + y0;
}

}  // namespace stdadk

using namespace stdadk;

extern "C" int stdadk_knot_windows_i32(const float *coords, int64_t B, const int32_t *sides_host,
                                       int32_t n_levels, int32_t p, int32_t *ix0, int32_t *iy0,
                                       int32_t *col0, stdadk_stream_t stream) {
  STDADK_REQUIRE(B >= 0 && n_levels > 0 && n_levels <= STDADK_MAX_LEVELS, STDADK_E_ARG, "knot_windows: bad sizes");
  if (B == 0) return 0;
  STDADK_REQUIRE(coords && sides_host && ix0 && iy0 && col0, STDADK_E_ARG, "knot_windows: NULL pointer");
  LevelTable lt;
  int off = 0;
  for (int l = 0; l < n_levels; ++l) {
    STDADK_REQUIRE(sides_host[l] >= 1, STDADK_E_ARG, "knot_windows: side < 1");
    lt.side[l] = sides_host[l];
    lt.off[l] = off;
    off += sides_host[l] * sides_host[l];
  }
  STDADK_LAUNCH(knot_windows_kernel, dim3((unsigned)ceil_div(B * n_levels, 256)), dim3(256), 0,
                     (hipStream_t)stream, coords, (int)B, (int)n_levels, lt, p, ix0, iy0, col0);
  STDADK_CHECK_LAUNCH("knot_windows");
  return 0;
}

extern "C" int stdadk_gather_batch_f32(const float *coords, const float *t, const float *y, const float *X,
                                       const int64_t *idx, int64_t B, int32_t Q, int32_t p, float *coords_out,
                                       float *t_out, float *y_out, float *X_out, stdadk_stream_t stream) {
  STDADK_REQUIRE(B >= 0 && B < (1ll << 31) && Q >= 0 && p >= 0, STDADK_E_ARG, "gather_batch: bad sizes");
  if (B == 0) return 0;
  STDADK_REQUIRE(coords && t && idx && coords_out && t_out && (Q == 0 || (y && y_out)) && (p == 0 || (X && X_out)),
                 STDADK_E_ARG, "gather_batch: NULL pointer");
  STDADK_LAUNCH(gather_batch_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, coords, t, y,
                X, idx, (int)B, Q, p, coords_out, t_out, y_out, X_out);
  STDADK_CHECK_LAUNCH("gather_batch");
  return 0;
}

extern "C" size_t stdadk_bin_workspace_bytes(int64_t B, int32_t G) {
  if (B < 0 || G < 1 || G > 256) return 0;
  return (size_t)(2 * align_up((size_t)G * G + 1, 64) + 2 * align_up((size_t)B, 64)) * sizeof(int) +
         3 * align_up((size_t)B, 64) * sizeof(float);
}

extern "C" int stdadk_bin_obs_f32(const float *coords, int64_t B, int32_t G, int32_t *keys,
                                  int32_t *cell_start, int32_t *perm, void *workspace,
                                  size_t workspace_bytes, stdadk_stream_t stream) {
  STDADK_REQUIRE(B >= 0 && B < (1ll << 31), STDADK_E_ARG, "bin_obs: bad B");
  STDADK_REQUIRE(G >= 1 && G <= 256, STDADK_E_ARG, "bin_obs: G must be in [1,256]");
  if (B == 0) return 0;
  STDADK_REQUIRE(coords && keys && cell_start && perm && workspace, STDADK_E_ARG, "bin_obs: NULL pointer");
  STDADK_REQUIRE(workspace_bytes >= stdadk_bin_workspace_bytes(B, G), STDADK_E_WORKSPACE, "bin_obs: workspace too small");
  int *w = (int *)workspace;
  BinBuffers bb;
  size_t nc = align_up((size_t)G * G + 1, 64), nb = align_up((size_t)B, 64);
  bb.keys = keys; bb.cell_start = cell_start; bb.perm = perm;
  bb.hist = w; w += nc;
  bb.cursor = w; w += nc;
  bb.perm_tmp = w; w += nb;
  w += nb;
  bb.xs = (float *)w; bb.ys = bb.xs + nb; bb.ts = bb.ys + nb;
  bb.y_s = nullptr; bb.X_s = nullptr;
  return bin_obs(coords, nullptr, nullptr, 0, nullptr, 0, (int)B, G, bb, (hipStream_t)stream);
}
