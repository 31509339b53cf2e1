// Tile body of the fp32 MFMA GEMM (see gemm_f32.h / gemm_f32.hip), shared by the GEMM kernels and by
// the merged weight-gradient kernel (dw_all.hip).
#pragma once
#include "gemm_f32.h"

namespace stdadk {

// Slab tiles that a workgroup on ANOTHER XCD reads later in the same launch (FinArgs): every XCD has its own L2,
// and a device-scope fence (__threadfence) on this part writes back / invalidates the WHOLE L2 -- measured: the
// merged weight-gradient launch 27 -> 153 us with one fence pair per workgroup.  Relaxed device-scope atomic
// stores / loads instead carry the scope on the access itself (sc1: written through, read past the non-coherent
// lines), so the hand-over costs a wait for the stores' acknowledgements and nothing else.
__device__ __forceinline__ void slab_store(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 slab_load2(const float *p) {      // 8-byte aligned
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
}


typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int GT = 256;  // threads

template <int ROWS, bool KM>
struct TileGeom {
  static constexpr int STRIDE = KM ? ROWS : (BK + 4);
  static constexpr int SIZE = KM ? BK * ROWS : ROWS * (BK + 4);
  static constexpr int ELEMS = ROWS * BK / GT;  // floats per thread per stage
};

// Global -> registers for one operand tile.  `r0` first row (m or n) of the tile, `k0` first k.
// VEC = contiguous floats per load along the operand's contiguous dimension.
// Every load is UNCONDITIONAL from a clamped, always-valid address (the tile's first row / k when out of range) and
// stays RAW in its registers: the masks for rows / k beyond the operand are applied by store_tile, when the values go
// to LDS one K step later.  A conditional load gets its own branch and a vmcnt(0); a value masked right behind its
// load makes the wave wait for the load BEFORE the MFMAs of the current step -- either way loads and matrix work
// would take turns instead of overlapping (round-2 ablation, profiles/r02_grouped_gemm_ablation.txt).
template <int ROWS, bool KM, int VEC>
__device__ __forceinline__ void load_tile(const float *__restrict__ P, int64_t ld, int r0, int nrows,
                                          int k0, int kend, float *reg) {
  const int tid = threadIdx.x;
  constexpr int N_VEC = TileGeom<ROWS, KM>::ELEMS / VEC;
  if (!KM) {
    constexpr int VPR = BK / VEC;       // vectors per row
    constexpr int RPP = GT / VPR;       // rows per pass
#pragma unroll
    for (int i = 0; i < N_VEC; ++i) {
      const int row = tid / VPR + i * RPP;
      const int k = k0 + (tid % VPR) * VEC;
      const bool rok = (r0 + row) < nrows;
      const float *src = P + (int64_t)(rok ? r0 + row : r0) * ld + (k < kend ? k : k0);
      if (VEC == 4) {
        const float4 v = *reinterpret_cast<const float4 *>(src);
        reg[i * 4 + 0] = v.x; reg[i * 4 + 1] = v.y; reg[i * 4 + 2] = v.z; reg[i * 4 + 3] = v.w;
      } else if (VEC == 2) {
        const float2 v = *reinterpret_cast<const float2 *>(src);
        reg[i * 2 + 0] = v.x; reg[i * 2 + 1] = v.y;
      } else {
        reg[i] = *src;
      }
    }
  } else {
    constexpr int VPR = ROWS / VEC;     // vectors per k-row
    constexpr int KPP = GT / VPR;       // k-rows per pass
#pragma unroll
    for (int i = 0; i < N_VEC; ++i) {
      const int kk = tid / VPR + i * KPP;
      const int r = r0 + (tid % VPR) * VEC;
      const bool kok = (k0 + kk) < kend;
      const float *src = P + (int64_t)(kok ? k0 + kk : k0) * ld + (r < nrows ? r : r0);
      if (VEC == 4) {
        const float4 v = *reinterpret_cast<const float4 *>(src);
        reg[i * 4 + 0] = v.x; reg[i * 4 + 1] = v.y; reg[i * 4 + 2] = v.z; reg[i * 4 + 3] = v.w;
      } else if (VEC == 2) {
        const float2 v = *reinterpret_cast<const float2 *>(src);
        reg[i * 2 + 0] = v.x; reg[i * 2 + 1] = v.y;
      } else {
        reg[i] = *src;
      }
    }
  }
}

// Registers -> LDS image, with the masks of the tile (the same (r0, nrows, k0, kend) the values were loaded with):
// element (row, k) is kept when row < nrows and k < kend, else 0.
template <int ROWS, bool KM, int VEC>
__device__ __forceinline__ void store_tile(float *lds, const float *reg, int r0, int nrows, int k0, int kend) {
  const int tid = threadIdx.x;
  constexpr int N_VEC = TileGeom<ROWS, KM>::ELEMS / VEC;
  constexpr int STRIDE = TileGeom<ROWS, KM>::STRIDE;
  float m[VEC];
  if (!KM) {
    constexpr int VPR = BK / VEC;
    constexpr int RPP = GT / VPR;
#pragma unroll
    for (int i = 0; i < N_VEC; ++i) {
      const int row = tid / VPR + i * RPP;
      const int kl = (tid % VPR) * VEC;
      const bool rok = (r0 + row) < nrows;
#pragma unroll
      for (int e = 0; e < VEC; ++e) m[e] = (rok && k0 + kl + e < kend) ? reg[i * VEC + e] : 0.f;
      float *dst = lds + row * STRIDE + kl;
      if (VEC == 4) *reinterpret_cast<float4 *>(dst) = make_float4(m[0], m[1], m[2], m[3]);
      else if (VEC == 2) *reinterpret_cast<float2 *>(dst) = make_float2(m[0], m[1]);
      else *dst = m[0];
    }
  } else {
    constexpr int VPR = ROWS / VEC;
    constexpr int KPP = GT / VPR;
#pragma unroll
    for (int i = 0; i < N_VEC; ++i) {
      const int kk = tid / VPR + i * KPP;
      const int rl = (tid % VPR) * VEC;
      const bool kok = (k0 + kk) < kend;
#pragma unroll
      for (int e = 0; e < VEC; ++e) m[e] = (kok && r0 + rl + e < nrows) ? reg[i * VEC + e] : 0.f;
      float *dst = lds + kk * STRIDE + rl;
      if (VEC == 4) *reinterpret_cast<float4 *>(dst) = make_float4(m[0], m[1], m[2], m[3]);
      else if (VEC == 2) *reinterpret_cast<float2 *>(dst) = make_float2(m[0], m[1]);
      else *dst = m[0];
    }
  }
}

// fragment of sub-step s for the 32 rows starting at `row` of an LDS operand image
template <int ROWS, bool KM>
__device__ __forceinline__ void read_frag(const float *lds, int row, int s, int lane, float *f) {
  constexpr int STRIDE = TileGeom<ROWS, KM>::STRIDE;
  const int r = row + (lane & 31), h = lane >> 5;
  if (!KM) {
    float4 v = *reinterpret_cast<const float4 *>(lds + r * STRIDE + 8 * s + 4 * h);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = lds[(8 * s + 4 * h + e) * STRIDE + r];
  }
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains every global access in flight
__device__ __forceinline__ void gemm_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int BM, int BN, bool A_KM, bool B_KM, int VA, int VB, int DEPTH = 1>
__device__ __forceinline__ void gemm_tile_body(const GemmArgs &g, int bx, int by, int bz, float *lds) {
  constexpr int TM = BM / 64, TN = BN / 64;
  using GA = TileGeom<BM, A_KM>;
  using GB = TileGeom<BN, B_KM>;
  float *As = lds, *Bs = lds + GA::SIZE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = by * BM, n0 = bx * BN;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
  int kbeg = 0, kend = g.K;
  if (g.splits > 1) {
    kbeg = bz * g.kps;
    kend = min(g.K, kbeg + g.kps);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto mfma_step = [&]() {
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      float fa[TM][4], fb[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) read_frag<BM, A_KM>(As, wm + i * 32, s, lane, fa[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) read_frag<BN, B_KM>(Bs, wn + j * 32, s, lane, fb[j]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
  };
  if constexpr (DEPTH == 1) {
    float ra[GA::ELEMS], rb[GB::ELEMS];
    if (kbeg < kend) {
      load_tile<BM, A_KM, VA>(g.A, g.lda, m0, g.M, kbeg, kend, ra);
      load_tile<BN, B_KM, VB>(g.B, g.ldb, n0, g.N, kbeg, kend, rb);
    }
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      store_tile<BM, A_KM, VA>(As, ra, m0, g.M, k0, kend);
      store_tile<BN, B_KM, VB>(Bs, rb, n0, g.N, k0, kend);
      gemm_lds_barrier();
      if (k0 + BK < kend) {
        load_tile<BM, A_KM, VA>(g.A, g.lda, m0, g.M, k0 + BK, kend, ra);
        load_tile<BN, B_KM, VB>(g.B, g.ldb, n0, g.N, k0 + BK, kend, rb);
      }
      mfma_step();
      gemm_lds_barrier();
    }
  } else if (kbeg < kend) {
    // DEPTH K steps of operands in flight (ring of register stages, the step loop unrolled over the ring): a short
    // K slice is a chain of load -> LDS -> MFMA steps in which one step of matrix work (0.45 us) cannot hide an L2
    // miss.  The loads are unconditional; a stage beyond the slice re-reads the slice's first step and is never stored.
    float ra[DEPTH][GA::ELEMS], rb[DEPTH][GB::ELEMS];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int k = kbeg + d * BK, kc = k < kend ? k : kbeg;
      load_tile<BM, A_KM, VA>(g.A, g.lda, m0, g.M, kc, kend, ra[d]);
      load_tile<BN, B_KM, VB>(g.B, g.ldb, n0, g.N, kc, kend, rb[d]);
    }
    for (int k0 = kbeg; k0 < kend; k0 += DEPTH * BK) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int k = k0 + d * BK;
        if (k >= kend) break;
        store_tile<BM, A_KM, VA>(As, ra[d], m0, g.M, k, kend);
        store_tile<BN, B_KM, VB>(Bs, rb[d], n0, g.N, k, kend);
        gemm_lds_barrier();
        const int kn = k + DEPTH * BK, kc = kn < kend ? kn : kbeg;
        load_tile<BM, A_KM, VA>(g.A, g.lda, m0, g.M, kc, kend, ra[d]);
        load_tile<BN, B_KM, VB>(g.B, g.ldb, n0, g.N, kc, kend, rb[d]);
        mfma_step();
        gemm_lds_barrier();
      }
    }
  }

  // epilogue: acc reg r of lane l is C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31] of its 32x32 tile
  float *Cbase;
  int64_t ldc;
  if (g.splits > 1 || g.always_slab) {
    Cbase = g.slab + (int64_t)bz * g.slab_stride;
    ldc = g.N;
  } else {
    Cbase = g.C;
    ldc = g.ldc;
  }
  const int h = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int col = n0 + wn + j * 32 + cl;
      if (col >= g.N) continue;
      float bv = (g.splits == 1 && !g.always_slab && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < g.M) {
          if (g.coherent_slab) slab_store(Cbase + (int64_t)row * ldc + col, acc[i][j][r]);
          else Cbase[(int64_t)row * ldc + col] = acc[i][j][r] + bv;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 operands for the TN products of a step (STDADK_FLAG_BF16): C[64 x 64] = A^T B over a K slice, both operands
// stored K-major in fp32 ([k][rows]).  The tiles are rounded to bf16 as they are stored to LDS (the operand
// boundary), [64 k][64 rows] images with a 96-element row stride, and the K-contiguous fragments the MFMA wants
// come from the transposing LDS read: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a
// 4 (k) x 16 (rows) block.  v_mfma_f32_32x32x16_bf16: lane l holds A[row l&31][k = 8(l>>5) + j], B[k][col l&31].
// Stride 96 (192 B = 48 banks): the four k-rows of the two blocks a 32-lane half reads fall on eight disjoint
// 8-bank runs -- conflict-free.
// ---------------------------------------------------------------------------------------------
typedef unsigned short hu16;
typedef short hs16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
constexpr int BKH = 64;            // k per stage
constexpr int HSTRIDE = 96;        // bf16 elements per k-row of an LDS image
constexpr int GROUP_LDS_FLOATS = (2 * BKH * HSTRIDE * 2) / 4 > TileGeom<64, true>::SIZE * 2 ? (2 * BKH * HSTRIDE * 2) / 4
                                                                                             : TileGeom<64, true>::SIZE * 2;

__device__ __forceinline__ uint32_t h_pack(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}

// rows r0.. of k-rows k0.. of a K-major fp32 operand into registers: 4 float4 per thread, unconditional RAW loads from
// clamped addresses; the masks (rows / k beyond the operand) are applied when the values are rounded and written to
// LDS one stage later (see load_tile: a value touched right behind its load makes the wave wait for it before the
// MFMAs of the current stage)
__device__ __forceinline__ void h_load(const float *__restrict__ P, int64_t ld, int r0, int nrows, int k0, int kend,
                                       float4 *reg) {
  const int tid = threadIdx.x;
  const int r = r0 + (tid & 15) * 4;
  const int rc = r < nrows ? r : r0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + (tid >> 4) + 16 * i;
    reg[i] = *reinterpret_cast<const float4 *>(P + (int64_t)(k < kend ? k : k0) * ld + rc);
  }
}
__device__ __forceinline__ void h_store(hu16 *lds, const float4 *reg, int r0, int nrows, int k0, int kend) {
  const int tid = threadIdx.x;
  const int r = r0 + (tid & 15) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kk = (tid >> 4) + 16 * i;
    const bool kok = k0 + kk < kend;
    const float x = (kok && r + 0 < nrows) ? reg[i].x : 0.f, y = (kok && r + 1 < nrows) ? reg[i].y : 0.f;
    const float z = (kok && r + 2 < nrows) ? reg[i].z : 0.f, w = (kok && r + 3 < nrows) ? reg[i].w : 0.f;
    *reinterpret_cast<uint2 *>(lds + kk * HSTRIDE + (tid & 15) * 4) = make_uint2(h_pack(x, y), h_pack(z, w));
  }
}
// fragment of the 16-deep k-step s for the 32 rows starting at `row`: two transposing reads (k = 8h + 0..3, + 4..7)
__device__ __forceinline__ hbf16x8 h_frag(const hu16 *lds, int row, int s, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const hu16 *p = lds + (16 * s + 8 * (g >> 1) + (i >> 2)) * HSTRIDE + row + 16 * (g & 1) + 4 * (i & 3);
  typedef __attribute__((address_space(3))) hs16x4 lds_s16x4;
  const hs16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p);
  const hs16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + 4 * HSTRIDE));
  typedef short hs16x8 __attribute__((ext_vector_type(8)));
  const hs16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(hbf16x8, v);
}

__device__ __forceinline__ void gemm_tn_tile_body_h(const GemmArgs &g, int bx, int by, int bz, float *lds_f) {
  hu16 *As = reinterpret_cast<hu16 *>(lds_f), *Bs = As + BKH * HSTRIDE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = by * 64, n0 = bx * 64;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  int kbeg = 0, kend = g.K;
  if (g.splits > 1) {
    kbeg = bz * g.kps;
    kend = min(g.K, kbeg + g.kps);
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 ra[4], rb[4];
  if (kbeg < kend) {
    h_load(g.A, g.lda, m0, g.M, kbeg, kend, ra);
    h_load(g.B, g.ldb, n0, g.N, kbeg, kend, rb);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BKH) {
    h_store(As, ra, m0, g.M, k0, kend);
    h_store(Bs, rb, n0, g.N, k0, kend);
    gemm_lds_barrier();
    if (k0 + BKH < kend) {
      h_load(g.A, g.lda, m0, g.M, k0 + BKH, kend, ra);
      h_load(g.B, g.ldb, n0, g.N, k0 + BKH, kend, rb);
    }
#pragma unroll
    for (int s = 0; s < BKH / 16; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h_frag(As, wm, s, lane), h_frag(Bs, wn, s, lane), acc, 0, 0, 0);
    gemm_lds_barrier();
  }
  // epilogue as the fp32 tiles: always a slab (grouped jobs)
  float *Cbase = g.slab + (int64_t)bz * g.slab_stride;
  const int h = lane >> 5, cl = lane & 31;
  const int col = n0 + wn + cl;
  if (col < g.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < g.M) {
        if (g.coherent_slab) slab_store(Cbase + (int64_t)row * g.N + col, acc[r]);
        else Cbase[(int64_t)row * g.N + col] = acc[r];
      }
    }
  }
}

// sum of four per-wave values of a 256-thread workgroup through `red` (>= 4 floats of LDS, free to overwrite);
// the result is valid in thread 0
__device__ __forceinline__ float block4_sum(float v, float *red) {
  const float s = wave_sum(v);
  __syncthreads();                                   // earlier readers of `red`
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

constexpr int REDUCE_TALL_COLS = 16;
// One workgroup of a table of fixed-order sums dst_j[i] = sum_s src_j[s*stride_j + i] (see reduce_jobs_kernel);
// returns (every thread) the squares of what this thread wrote.  smem: >= 4 * 64 floats.
__device__ __forceinline__ float reduce_job_block(const ReduceGroup &grp, int block, float *smem) {
  int j = 0;
  while (j + 1 < grp.n && block >= grp.first_block[j + 1]) ++j;
  const ReduceJob &jb = grp.job[j];
  const int blk = block - grp.first_block[j];
  float sq = 0.f;
  if (jb.wide) {
    const int c = (blk * 256 + threadIdx.x) * 4;
    if (c < jb.n) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int s = 0;
      for (; s + 7 < jb.splits; s += 8) {           // 8 independent loads in flight (same serial order of the adds)
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(jb.src + (int64_t)(s + u) * jb.stride + c);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      for (; s + 3 < jb.splits; s += 4) {           // 4 independent loads in flight
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(jb.src + (int64_t)(s + u) * jb.stride + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      for (; s < jb.splits; ++s) {
        const float4 v = *reinterpret_cast<const float4 *>(jb.src + (int64_t)s * jb.stride + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      *reinterpret_cast<float4 *>(jb.dst + c) = acc;
      sq = (acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w);
    }
  } else {
    // tall job: REDUCE_TALL_COLS columns x 16 groups of partials per workgroup (a thread walks every 16th partial
    // with 8 independent loads in flight: 256 partials = two round trips; with 64 columns x 4 groups it was eight)
    float(*red)[REDUCE_TALL_COLS] = reinterpret_cast<float(*)[REDUCE_TALL_COLS]>(smem);
    constexpr int SG = 256 / REDUCE_TALL_COLS;
    const int tx = threadIdx.x % REDUCE_TALL_COLS, ty = threadIdx.x / REDUCE_TALL_COLS;
    const int c = blk * REDUCE_TALL_COLS + tx;
    float s0 = 0.f, s1 = 0.f;
    if (c < jb.n) {
      int s = ty;
      for (; s + 7 * SG < jb.splits; s += 8 * SG) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = jb.src[(int64_t)(s + SG * u) * jb.stride + c];
        s0 += (v[0] + v[2]) + (v[4] + v[6]);
        s1 += (v[1] + v[3]) + (v[5] + v[7]);
      }
      for (; s < jb.splits; s += SG) s0 += jb.src[(int64_t)s * jb.stride + c];
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && c < jb.n) {
      float o = 0.f;
#pragma unroll
      for (int q = 0; q < SG; q += 4) o += (red[q][tx] + red[q + 1][tx]) + (red[q + 2][tx] + red[q + 3][tx]);
      jb.dst[c] = o;
      sq = o * o;
    }
  }
  return sq;
}

// Finishing step of one K slice of a grouped job's output tile (FinArgs): once the slice's slab stores
// (slab_store: device scope) are acknowledged the tile's arrival counter is advanced, and the workgroup that finds
// the other splits - 1 slices already there sums the tile over the slabs in slice order s = 0, 1, ... (the wide
// reduce job's order), writes C and the tile's squared-norm slot.
__device__ __forceinline__ void gemm_tile_finish(const GemmArgs &g, int bx, int by, int *cnt, float *slot, float *lds) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's slab stores have been acknowledged
  __syncthreads();
  int *flag = reinterpret_cast<int *>(lds);
  if (threadIdx.x == 0) flag[0] = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (flag[0] != g.splits - 1) return;                // workgroup-uniform
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // thread -> two neighbouring columns of rows r, r + 8, ..., r + 56 of the tile: a wave's load covers two whole
  // 256-byte tile rows.  The loads are UNCONDITIONAL from clamped addresses (a load under a branch makes the
  // compiler wait for it where the paths meet: every one of the splits x 8 loads would pay its full latency --
  // measured 60 us for this sum), the masks apply to the stores.
  float sq = 0.f;
  const int r = threadIdx.x >> 5, c = bx * 64 + 2 * (threadIdx.x & 31);
  const bool pair_ok = (g.N & 1) == 0 && (g.slab_stride & 1) == 0 && (reinterpret_cast<uintptr_t>(g.slab) & 7) == 0 &&
                       (reinterpret_cast<uintptr_t>(g.C) & 7) == 0;
  if (pair_ok) {                                      // workgroup-uniform
    const bool cok = c < g.N;
    const int cc = cok ? c : bx * 64;
#pragma unroll 1
    for (int hp = 0; hp < 2; ++hp) {                  // rows r + 8 p, p = 4 hp .. 4 hp + 3
      const float *src[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) src[p] = g.slab + (int64_t)min(by * 64 + 8 * (4 * hp + p) + r, g.M - 1) * g.N + cc;
      float2 acc[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[p] = make_float2(0.f, 0.f);
      int s = 0;
      for (; s + 3 < g.splits; s += 4) {              // 16 independent loads in flight
        float2 v[4][4];
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int p = 0; p < 4; ++p)
            v[w][p] = slab_load2(src[p] + (int64_t)(s + w) * g.slab_stride);
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int p = 0; p < 4; ++p) { acc[p].x += v[w][p].x; acc[p].y += v[w][p].y; }
      }
      for (; s < g.splits; ++s) {
        float2 v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = slab_load2(src[p] + (int64_t)s * g.slab_stride);
#pragma unroll
        for (int p = 0; p < 4; ++p) { acc[p].x += v[p].x; acc[p].y += v[p].y; }
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = by * 64 + 8 * (4 * hp + p) + r;
        if (cok && row < g.M) {
          *reinterpret_cast<float2 *>(g.C + (int64_t)row * g.N + c) = acc[p];
          sq += acc[p].x * acc[p].x + acc[p].y * acc[p].y;
        }
      }
    }
  } else {
    for (int p = 0; p < 8; ++p) {
      const int row = by * 64 + 8 * p + r;
      for (int u = 0; u < 2; ++u) {
        if (row >= g.M || c + u >= g.N) continue;
        float acc = 0.f;
        for (int s = 0; s < g.splits; ++s)
          acc += __hip_atomic_load(g.slab + (int64_t)s * g.slab_stride + (int64_t)row * g.N + c + u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        g.C[(int64_t)row * g.N + c + u] = acc;
        sq = fmaf(acc, acc, sq);
      }
    }
  }
  if (slot) {                                         // workgroup-uniform
    const float t = block4_sum(sq, lds + 8);
    if (threadIdx.x == 0) slot[0] = t;
  }
}

// K steps of operands in flight in the grouped TN tiles (gemm_tile_body's DEPTH).  -DSTDADK_GROUP_DEPTH=3 measured
// against 1 on one box (round 3): dw_all_kernel 31.0 vs 30.4 us at 4 096 rows, 83.0 vs 80.8 at 16 384, 329-337 vs
// 317-319 at 65 536 -- the tiles are not waiting for their operands; 1 stays
#ifndef STDADK_GROUP_DEPTH
#define STDADK_GROUP_DEPTH 1
#endif
constexpr int GROUP_DEPTH = STDADK_GROUP_DEPTH;
// one workgroup of a grouped TN launch: block -> (job, tile, split) through the prefix table
__device__ __forceinline__ void gemm_tn_grouped_block(const GemmGroup &grp, int block, float *lds,
                                                      const FinArgs *fin = nullptr) {
  int j = 0;
  while (j + 1 < grp.n && block >= grp.first_block[j + 1]) ++j;
  const GemmArgs &g = grp.job[j];
  int b = block - grp.first_block[j];
  const int tn = (g.N + 63) >> 6, tm = (g.M + 63) >> 6;
  if (b >= tn * tm * g.splits) return;        // padding in front of the next job's aligned first block
  int bz;
  if (g.xcd_split) {
    // XCD-aware order (a job's first block is a multiple of 8 and workgroups go to the XCDs round-robin, so b & 7 is
    // the XCD): XCD x takes the x-th contiguous eighth of the K slices of EVERY output tile, i.e. one eighth of the
    // rows of both operands -- each XCD's L2 then fetches an eighth of dZ / activations instead of all of them (the
    // operands were written by the previous launch on other XCDs: every first touch is a fabric read).  The sorted
    // batch is cell-ordered, so that eighth is also the stripe of the domain whose dZ_0 rows the knot groups of the
    // same XCD gather (l1_window_bwd_multi_body).
    const int x = b & 7, i = b >> 3, spx = g.splits >> 3;      // splits is a multiple of 8
    bz = x * spx + i % spx;
    b = i / spx;
  } else {
    bz = b / (tn * tm);
    b -= bz * tn * tm;
  }
  if (g.bf16) gemm_tn_tile_body_h(g, b % tn, b / tn, bz, lds);      // workgroup-uniform
  else gemm_tile_body<64, 64, true, true, 4, 4, GROUP_DEPTH>(g, b % tn, b / tn, bz, lds);
  if (fin && fin->cnt)
    gemm_tile_finish(g, b % tn, b / tn, fin->cnt + fin->tile0[j] + b, fin->slots ? fin->slots + fin->tile0[j] + b : nullptr,
                     lds);
}

}  // namespace stdadk
