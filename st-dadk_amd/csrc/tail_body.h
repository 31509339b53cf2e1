// Bodies of the fused MLP-tail kernels (see tail.h / tail.hip), shared by their own kernels and by the
// fused training-step kernel (fused_step.hip).  `tile` = index of the row tile (R rows) a workgroup carries.
//
// Everything that depends on the workgroup shape lives in Tail<NW> (NW waves of 64 lanes):
//   Tail<16>: 1024 threads, ONE workgroup per CU, one 16-column N tile per wave; 16 / 32 / 64-row tiles.  The shape of
//             every launch of the library and of the fused step kernel (fused_step.hip) -- at 16 rows per CU the
//             phases are latency-bound and more waves = more loads in flight.
//   Tail<8>:  512 threads, up to two N tiles per wave, 32-row tiles whose LDS (<= 76 KiB) and registers (<= 128) let
//             TWO workgroups share a CU, so that one's row-local phases (LayerNorm, stores, barriers) could run beside
//             the other's GEMM phases.  Built, parity-tested and measured in round 2: no faster at any batch size
//             (DESIGN.md section 8), so the library does not instantiate it; tools/diag/occupancy_tail8.hip does.
#pragma once
#include "tail.h"
#include "basis.h"

namespace stdadk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ACT_LD = TAIL_MAX_W + 4;        // activation row stride in LDS (floats)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding
// GLOBAL store (vmcnt(0)); nothing in these kernels reads global data written by another wave.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int NI>
struct BFrag { float4 v[2 * NI]; };     // [2*i + j]: chunk fragment of this lane, N tile i, 16-deep half j

// ---------------------------------------------------------------------------------------------
// bf16 operands (STDADK_FLAG_BF16): v_mfma_f32_16x16x32_bf16, fp32 accumulate
// ---------------------------------------------------------------------------------------------
// Lane l = 16 q + c16 holds A[row c16][k = 8q + j] and B[k = 8q + j][col c16], j = 0..7 (one 16-byte piece
// each); D as for the fp32 form.  A 64-deep chunk is TWO MFMAs: lane group q owns the 16 consecutive k
// 64c + 16q .. +15 of its row (A: the bf16 activation image in LDS; B: row n of the [N][K] bf16 weights), the
// first MFMA takes its low 8, the second its high 8 -- any assignment of k to (MFMA, q, j) sums the same
// products as long as A and B agree on it, and this one makes each weight row's four pieces one full 128-byte line.
typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int ABF_LD = TAIL_MAX_W + 8;        // bf16 activation row stride in LDS (elements; 528 B, 16-byte aligned)

__device__ __forceinline__ f32x4 mfma16h(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 v = {(__bf16)a, (__bf16)b};       // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ u16 to_bf16(float a) { return (u16)(pack_bf16(a, 0.f) & 0xffffu); }

template <int NI>
struct BFragH { uint4 v[2 * NI]; };     // [2*i + half]: this lane's 16 k of weight row n_i in chunk c

#ifdef STDADK_DIAG   // diagnostic build only: in-kernel wall-clock stamps of the phases
#define STAMP(i) do { if (a.stamps && tid == 0) a.stamps[tile * 16 + (i)] = wall_clock64(); } while (0)
// per-WAVE stamps of the first GEMM phase of the 64-row forward (MT = 4): the stamp buffer of tools/stamp_tail.py holds
// B/16 x 16 slots, the 64-row tiles use the first quarter, [tile][wave][3] goes behind it
#define WSTAMP(i) do { if (MT == 4 && a.stamps && lane == 0) \
  a.stamps[(size_t)((a.B + 63) / 64) * 16 + ((size_t)tile * 16 + wave) * 3 + (i)] = wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#define WSTAMP(i) do { } while (0)
#endif

template <int NW_>
struct Tail {
static constexpr int NW = NW_;                       // waves per workgroup
static constexpr int TT = 64 * NW;                   // threads per workgroup
static constexpr int MAX_NI = TAIL_MAX_W / 16 / NW;  // N tiles per wave: tile i of a wave is column tile wave + NW*i
static_assert(MAX_NI == 1 || MAX_NI == 2, "Tail<NW>: 8 or 16 waves");

// N tiles per wave a GEMM of width N runs with (scalar).  A tile index beyond the layer's N/16 tiles (widths that
// are not a multiple of 16*NW) is clamped for the loads and its accumulator is never stored.
static __device__ __forceinline__ int tiles_of(int N) { return ((N >> 4) + NW - 1) / NW; }

// Loads are UNCONDITIONAL from clamped (always valid) addresses and never masked in registers: a
// conditional load gets its own branch + vmcnt(0), and a select on the loaded value drags the wait
// in front of the MFMAs of the previous chunk.  The number of N tiles of this wave (NI) is a
// template parameter and the K-half test is scalar, so the MFMA stream has no exec-masked branches.
// KN = false: W is [N][K] (K contiguous): one dwordx4 per tile and 16-deep half.
// KN = true:  W is [K][N] row-major (N contiguous): nn.Linear's own (out,in) weight read as the B operand of
//             dA = dZ . W -- four dword loads (64-byte pieces per 16 lanes) instead of one dwordx4, and no
//             transposed copy of the weights is needed.  Component e of v[..] is k = 32c + 16j + 4q + e in both
//             forms, the same k order as the A fragments.
template <int NI, bool KN>
static __device__ __forceinline__ void load_bfrag(BFrag<MAX_NI> &f, const float *__restrict__ W, int N, int K, int c,
                                                  int wave, int c16, int q) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = 32 * c + 16 * j + 4 * q;
    const int kc = k < K ? k : 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = min(16 * (wave + NW * i), N - 16) + c16;
      if constexpr (KN) {
        const float *b = W + (size_t)kc * N + n;
        f.v[2 * i + j] = make_float4(b[0], b[N], b[2 * (size_t)N], b[3 * (size_t)N]);
      } else {
        f.v[2 * i + j] = *reinterpret_cast<const float4 *>(W + (size_t)n * K + kc);
      }
    }
  }
}

template <int NI, int MT>
static __device__ __forceinline__ void mma_chunk(f32x4 (*acc)[MAX_NI], const BFrag<MAX_NI> &f, const float *__restrict__ A,
                                                 int K, int c, int c16, int q) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (32 * c + 16 * j < K) {          // scalar: K is a multiple of 16, a 16-deep half is all in or out
      float af[MT][4];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float4 av = *reinterpret_cast<const float4 *>(A + (16 * mt + c16) * ACT_LD + 32 * c + 16 * j + 4 * q);
        af[mt][0] = av.x; af[mt][1] = av.y; af[mt][2] = av.z; af[mt][3] = av.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const float4 bv = f.v[2 * i + j];
          const float bf[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {    // every weight fragment feeds all row tiles
            acc[mt][i] = mfma16(af[mt][e], bf[e], acc[mt][i]);
          }
        }
      }
    }
  }
}

// Every workgroup of a launch multiplies by the SAME weights; walking the K chunks in the same order, the 32 CUs of an
// XCD ask its L2 for the same 128-byte lines at the same moment.  With STDADK_KROT=1 a workgroup starts its walk at
// chunk `rot mod nchunk` and wraps around, so that at any instant the CUs of an XCD are spread over the chunks, i.e.
// over different lines / channels; only the order of a tile's fp32 partial sums changes.  Measured (round 3, VERDICT
// r2 item 3): no gain -- off by default (rot = 0 is the natural order), kept as a switch for that measurement.
static __device__ __forceinline__ int chunk_start(int rot, int nchunk) { return rot % nchunk; }

// mma_chunk for a chunk that lies entirely below K (no scalar test per 16-deep half: straight-line code)
template <int NI, int MT>
static __device__ __forceinline__ void mma_chunk_full(f32x4 (*acc)[MAX_NI], const BFrag<MAX_NI> &f, const float *__restrict__ A,
                                                      int c, int c16, int q) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float af[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float4 av = *reinterpret_cast<const float4 *>(A + (16 * mt + c16) * ACT_LD + 32 * c + 16 * j + 4 * q);
      af[mt][0] = av.x; af[mt][1] = av.y; af[mt][2] = av.z; af[mt][3] = av.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const float4 bv = f.v[2 * i + j];
        const float bf[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][i] = mfma16(af[mt][e], bf[e], acc[mt][i]);
      }
    }
  }
}

// The first 32-deep weight chunk of a GEMM phase, requested one phase EARLY (before the row-local
// LayerNorm / input phase that precedes the GEMM) so that its L2 round trip is hidden behind that phase.
// `rot`: this workgroup's rotation of the K-chunk order (chunk_start above), 0 = natural order.
template <bool KN = false>
static __device__ __forceinline__ void preload_w(BFrag<MAX_NI> &f, const float *__restrict__ W, int N, int K,
                                                 int wave, int c16, int q, int rot = 0) {
  // always all MAX_NI tiles (a tile the layer does not have is a clamped, unused load): one write pattern for the
  // fragment whatever the layer's width keeps it in registers
  if (wave < (N >> 4)) load_bfrag<MAX_NI, KN>(f, W, N, K, chunk_start(rot, (K + 31) >> 5), wave, c16, q);
}

// GEMM of a phase: acc[mt][i] (rows 16 mt.., N tile wave + NW*i) += A[R x K] (LDS, row stride ACT_LD) * W.
// M = 16..64 rows is the GEMV-like regime: every wave streams ITS OWN slice of W straight into VGPRs (no
// LDS staging, no workgroup barrier in the K loop), two 32-deep chunks in flight; chunk 0 is already in
// registers (preload_w).
template <int MT, int NI, bool KN>
static __device__ __forceinline__ void gemm16_loop(f32x4 (*acc)[MAX_NI], const float *__restrict__ A,
                                                   const float *__restrict__ W, int N, int K, int wave, int c16,
                                                   int q, BFrag<MAX_NI> &f0, int rot) {
  const int nchunk = (K + 31) >> 5;
  BFrag<MAX_NI> f1;
  int c = chunk_start(rot, nchunk);             // f0 holds this chunk (preload_w)
  for (int i = 0; i < nchunk; i += 2) {
    const int c1 = c + 1 == nchunk ? 0 : c + 1;
    if (i + 1 < nchunk) load_bfrag<NI, KN>(f1, W, N, K, c1, wave, c16, q);
    mma_chunk<NI, MT>(acc, f0, A, K, c, c16, q);
    if (i + 1 < nchunk) {
      const int c2 = c1 + 1 == nchunk ? 0 : c1 + 1;
      if (i + 2 < nchunk) load_bfrag<NI, KN>(f0, W, N, K, c2, wave, c16, q);
      mma_chunk<NI, MT>(acc, f1, A, K, c1, c16, q);
      c = c2;
    }
  }
}
// K = 32 NCH exactly (the widths 256 and 128 of every shipped configuration), natural chunk order: the chunk loop
// fully unrolled over a ring of three fragment buffers, chunks c + 1 and c + 2 in flight while chunk c is multiplied.
// Straight-line code matters here: in the rolled loop above the loads of the next chunk sit behind a scalar branch
// (the last chunk has no successor), and where the two paths meet the compiler can only wait for the larger of their
// outstanding-load counts -- s_waitcnt vmcnt(1) / vmcnt(0) right behind the loads it has just issued (ROCm 7.2), i.e.
// every chunk's MFMAs waited for the NEXT chunk's fragments: no weight chunk was ever in flight under the MFMAs, and
// the phase ran at ~60-70 % of the matrix pipe with four waves per SIMD covering for each other (round-3 per-wave
// stamps, tools/diag/wave_stamps.py).  Unrolled, every wait is an exact count.
template <int MT, int NI, bool KN, int NCH>
static __device__ __forceinline__ void gemm16_unrolled(f32x4 (*acc)[MAX_NI], const float *__restrict__ A,
                                                       const float *__restrict__ W, int N, int K, int wave, int c16,
                                                       int q, const BFrag<MAX_NI> &f0) {
  // (also measured: the A fragments of the next 16-deep step read from LDS ahead of the current step's MFMAs, two
  //  register sets -- 56.8 vs 57.3 us for the 16-row fused kernel, 403 vs 388 us at 64 rows; and that order pinned
  //  with sched_group_barrier -- 59.8 / 407 us: the compiler's own placement of the LDS reads stays)
  BFrag<MAX_NI> f[3];
  f[0] = f0;
  if (NCH > 1) load_bfrag<NI, KN>(f[1], W, N, K, 1, wave, c16, q);
  if (NCH > 2) load_bfrag<NI, KN>(f[2], W, N, K, 2, wave, c16, q);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    mma_chunk_full<NI, MT>(acc, f[c % 3], A, c, c16, q);
    if (c + 3 < NCH) load_bfrag<NI, KN>(f[c % 3], W, N, K, c + 3, wave, c16, q);
  }
}

template <int MT, bool KN = false>
static __device__ __forceinline__ void gemm16_pre(f32x4 (*acc)[MAX_NI], const float *__restrict__ A,
                                                  const float *__restrict__ W, int N, int K, int wave, int c16,
                                                  int q, BFrag<MAX_NI> &f0, int rot = 0) {
  if (wave >= (N >> 4)) return;                 // scalar: this wave has no N tile in a narrow layer

  if (MAX_NI > 1 && tiles_of(N) > 1) gemm16_loop<MT, MAX_NI, KN>(acc, A, W, N, K, wave, c16, q, f0, rot);
  else if (MAX_NI == 1 && rot == 0 && K == 256) gemm16_unrolled<MT, 1, KN, 8>(acc, A, W, N, K, wave, c16, q, f0);
  else if (MAX_NI == 1 && rot == 0 && K == 128) gemm16_unrolled<MT, 1, KN, 4>(acc, A, W, N, K, wave, c16, q, f0);
  else gemm16_loop<MT, 1, KN>(acc, A, W, N, K, wave, c16, q, f0, rot);
}

// ---- bf16 operands
// unconditional loads from clamped addresses like load_bfrag; a piece beyond K meets zeros in the A image
template <int NI>
static __device__ __forceinline__ void load_bfrag_h(BFragH<MAX_NI> &f, const u16 *__restrict__ Wn, int N, int K, int c,
                                                    int wave, int c16, int q) {
  const int k = 64 * c + 16 * q;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int n = min(16 * (wave + NW * i), N - 16) + c16;
    const uint4 *src = reinterpret_cast<const uint4 *>(Wn + (size_t)n * K + (k < K ? k : 0));
    f.v[2 * i] = src[0]; f.v[2 * i + 1] = src[1];
  }
}

template <int NI, int MT>
static __device__ __forceinline__ void mma_chunk_h(f32x4 (*acc)[MAX_NI], const BFragH<MAX_NI> &f, const u16 *__restrict__ A,
                                                   int c, int c16, int q) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const uint4 *ap = reinterpret_cast<const uint4 *>(A + (16 * mt + c16) * ABF_LD + 64 * c + 16 * q);
    const uint4 a0 = ap[0], a1 = ap[1];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      acc[mt][i] = mfma16h(a0, f.v[2 * i], acc[mt][i]);
      acc[mt][i] = mfma16h(a1, f.v[2 * i + 1], acc[mt][i]);
    }
  }
}

static __device__ __forceinline__ void preload_wh(BFragH<MAX_NI> &f, const u16 *__restrict__ Wn, int N, int K, int wave,
                                                  int c16, int q, int rot = 0) {
  if (wave < (N >> 4)) load_bfrag_h<MAX_NI>(f, Wn, N, K, chunk_start(rot, (K + 63) >> 6), wave, c16, q);
}

// acc[mt][i] += A[R x K] (bf16 image in LDS, columns K .. 64 ceil(K/64) zero) * W^T, W = [N][K] bf16 in global
// memory, chunk 0 already in registers (preload_wh), two chunks in flight
template <int MT, int NI>
static __device__ __forceinline__ void gemm16_loop_h(f32x4 (*acc)[MAX_NI], const u16 *__restrict__ A,
                                                     const u16 *__restrict__ Wn, int N, int K, int wave, int c16, int q,
                                                     BFragH<MAX_NI> &f0, int rot) {
  const int nchunk = (K + 63) >> 6;
  BFragH<MAX_NI> f1;
  int c = chunk_start(rot, nchunk);
  for (int i = 0; i < nchunk; i += 2) {
    const int c1 = c + 1 == nchunk ? 0 : c + 1;
    if (i + 1 < nchunk) load_bfrag_h<NI>(f1, Wn, N, K, c1, wave, c16, q);
    mma_chunk_h<NI, MT>(acc, f0, A, c, c16, q);
    if (i + 1 < nchunk) {
      const int c2 = c1 + 1 == nchunk ? 0 : c1 + 1;
      if (i + 2 < nchunk) load_bfrag_h<NI>(f0, Wn, N, K, c2, wave, c16, q);
      mma_chunk_h<NI, MT>(acc, f1, A, c1, c16, q);
      c = c2;
    }
  }
}
// the unrolled ring of gemm16_unrolled for the bf16 operands: K = 64 NCH exactly (256 -> 4 chunks, 128 -> 2)
template <int MT, int NI, int NCH>
static __device__ __forceinline__ void gemm16_unrolled_h(f32x4 (*acc)[MAX_NI], const u16 *__restrict__ A,
                                                         const u16 *__restrict__ Wn, int N, int K, int wave, int c16,
                                                         int q, const BFragH<MAX_NI> &f0) {
  BFragH<MAX_NI> f[3];
  f[0] = f0;
  if (NCH > 1) load_bfrag_h<NI>(f[1], Wn, N, K, 1, wave, c16, q);
  if (NCH > 2) load_bfrag_h<NI>(f[2], Wn, N, K, 2, wave, c16, q);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    mma_chunk_h<NI, MT>(acc, f[c % 3], A, c, c16, q);
    if (c + 3 < NCH) load_bfrag_h<NI>(f[c % 3], Wn, N, K, c + 3, wave, c16, q);
  }
}

template <int MT>
static __device__ __forceinline__ void gemm16_pre_h(f32x4 (*acc)[MAX_NI], const u16 *__restrict__ A,
                                                    const u16 *__restrict__ Wn, int N, int K, int wave, int c16, int q,
                                                    BFragH<MAX_NI> &f0, int rot = 0) {
  if (wave >= (N >> 4)) return;
  if (MAX_NI > 1 && tiles_of(N) > 1) gemm16_loop_h<MT, MAX_NI>(acc, A, Wn, N, K, wave, c16, q, f0, rot);
  else if (MAX_NI == 1 && rot == 0 && K == 256) gemm16_unrolled_h<MT, 1, 4>(acc, A, Wn, N, K, wave, c16, q, f0);
  else if (MAX_NI == 1 && rot == 0 && K == 128) gemm16_unrolled_h<MT, 1, 2>(acc, A, Wn, N, K, wave, c16, q, f0);
  else gemm16_loop_h<MT, 1>(acc, A, Wn, N, K, wave, c16, q, f0, rot);
}

// ---------------------------------------------------------------------------------------------
// layer 0 from the raw observations (TailDense0)
// ---------------------------------------------------------------------------------------------
// [X | phi | psi] of rows row0.. of the batch into LDS: columns [0, 256) in act0, [256, 512) in act1, zero beyond
// D; and into the feature buffer (row stride ldf = D rounded up to 32, padding zero) for the backward.
// Same arithmetic as rbf_build_kernel (phi_eval / psi_eval of basis.h).
template <int MT>
static __device__ __forceinline__ void d0_fill_features(const TailDense0 &z, float *act0, float *act1, int row0, int B) {
  constexpr int R = 16 * MT;
  const int D = z.p + z.Ks + z.Kt;
  const int Dp = (D + 31) & ~31;
  for (int idx = threadIdx.x; idx < R * Dp; idx += TT) {
    const int row = idx / Dp, col = idx - row * Dp;
    const int grow = min(row0 + row, B - 1);
    float v = 0.f;
    if (col < z.p) {
      v = z.X[(size_t)grow * z.p + col];
    } else if (col < z.p + z.Ks) {
      const int k = col - z.p;
      const float x = z.coords[2 * grow], y = z.coords[2 * grow + 1];
      const float cx = z.s_centers[2 * k], cy = z.s_centers[2 * k + 1];
      const float sc = knot_scale(z.s_bw[k], z.cal);
      v = z.basis == STDADK_BASIS_WENDLAND ? phi_eval<STDADK_BASIS_WENDLAND>(x, y, cx, cy, sc)
          : (z.basis == STDADK_BASIS_GAUSSIAN ? phi_eval<STDADK_BASIS_GAUSSIAN>(x, y, cx, cy, sc)
                                              : phi_eval<STDADK_BASIS_TRIANGULAR>(x, y, cx, cy, sc));
    } else if (col < D) {
      const int j = col - z.p - z.Ks;
      v = psi_eval(z.t[grow], z.t_centers[j], z.t_bw[j]);
    }
    if (col < TAIL_MAX_W) act0[row * ACT_LD + col] = v;
    else act1[row * ACT_LD + col - TAIL_MAX_W] = v;
    if (z.feats && row0 + row < B) z.feats[(size_t)(row0 + row) * z.ldf + col] = v;
  }
  // LDS columns [Dp, 256) / [Dp - 256, 256) are never read: the GEMM below stops at Dp
}

// B fragment of chunk c from W0^T [D][N] with every row index clamped below D (the A columns there are zero)
static __device__ __forceinline__ void d0_load_bfrag(BFrag<MAX_NI> &f, const float *__restrict__ W0T, int N, int D, int c, int wave,
                                              int c16, int q) {
  const int n = 16 * wave + c16;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = 32 * c + 16 * j + 4 * q;
    const float *b = W0T + n;
    f.v[j] = make_float4(b[(size_t)min(k, D - 1) * N], b[(size_t)min(k + 1, D - 1) * N],
                         b[(size_t)min(k + 2, D - 1) * N], b[(size_t)min(k + 3, D - 1) * N]);
  }
}

// acc += features (LDS, two 256-column halves) . W0^T, 32-deep chunks, two in flight like gemm16_pre
template <int MT>
static __device__ __forceinline__ void d0_gemm(f32x4 (*acc)[MAX_NI], const float *act0, const float *act1,
                                               const float *__restrict__ W0T, int N, int D, int wave, int c16, int q,
                                               BFrag<MAX_NI> &f0) {
  static_assert(MAX_NI == 1, "the dense layer 0 inside the tail launch is built for 16 waves");
  if (wave >= (N >> 4)) return;
  const int Kp = (D + 15) & ~15;                 // mma_chunk works in 16-deep halves
  const int nchunk = (D + 31) >> 5;
  constexpr int HC = TAIL_MAX_W / 32;            // chunks per LDS half
  BFrag<MAX_NI> f1;
  for (int c = 0; c < nchunk; c += 2) {
    if (c + 1 < nchunk) d0_load_bfrag(f1, W0T, N, D, c + 1, wave, c16, q);
    mma_chunk<1, MT>(acc, f0, c < HC ? act0 : act1 - TAIL_MAX_W, Kp, c, c16, q);
    if (c + 1 < nchunk) {
      if (c + 2 < nchunk) d0_load_bfrag(f0, W0T, N, D, c + 2, wave, c16, q);
      mma_chunk<1, MT>(acc, f1, c + 1 < HC ? act0 : act1 - TAIL_MAX_W, Kp, c + 1, c16, q);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// row-local phases of a layer, one wave per RPW rows of the tile, lane l owning the columns l + 64 cc
// ---------------------------------------------------------------------------------------------
// Specialised on the layer's width: CC = 64-column groups a lane walks, FULL = the width is exactly 64 CC.  With FULL
// nothing is masked: no select per element, and -- what cost more -- no exec-masked branch around every store (the
// compiler turns `if (column < h) store` into a save-exec / branch / restore block per store: twelve per row in the
// forward phase).  A 128-wide layer walks two groups instead of four masked ones.  The generic instantiation
// (CC = 4, masks) keeps every other width working.  Same arithmetic, in the same order, in all of them.
struct LnFwdCtx {
  float *xhat, *act, *rstd;      // NULL in eval mode
  int layer_id, h, B;
  float eps;
  uint64_t seed;
  float keep_scale;
  uint32_t drop_thr;
  bool drop_on, ln_on;
};

template <int RPW, int CC, bool FULL, bool BF>
static __device__ __forceinline__ void ln_fwd_rows(const LnFwdCtx &c, float *nxt, u16 *abf, int wave, int lane, int row0,
                                                   const float (&gv)[4], const float (&bev)[4]) {
  const int h = c.h;
  bool okc[CC];
#pragma unroll
  for (int cc = 0; cc < CC; ++cc) okc[cc] = FULL || lane + 64 * cc < h;
  const float inv_h = 1.0f / (float)h;
  float z[RPW][CC], mean[RPW], rs[RPW];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const float *zp = nxt + (RPW * wave + rr) * ACT_LD + lane;      // lane + 64 cc < 256 <= ACT_LD: in the row
    float sm = 0.f;
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
      const float t = zp[64 * cc];
      z[rr][cc] = okc[cc] ? t : 0.f;
      sm += z[rr][cc];
    }
    mean[rr] = sm;
  }
  if (c.ln_on) {
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) mean[rr] = wave_sum(mean[rr]) * inv_h;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      float sq = 0.f;
#pragma unroll
      for (int cc = 0; cc < CC; ++cc) {
        const float d = okc[cc] ? z[rr][cc] - mean[rr] : 0.f;
        sq += d * d;
      }
      rs[rr] = sq;
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) rs[rr] = ln_rstd(wave_sum(rs[rr]) * inv_h, c.eps);
  } else {
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) { mean[rr] = 0.f; rs[rr] = 1.f; }
  }
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = RPW * wave + rr;
    const int grow = row0 + row;
    float xh[CC], v[CC];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
      xh[cc] = c.ln_on ? (z[rr][cc] - mean[rr]) * rs[rr] : z[rr][cc];
      v[cc] = fmaxf(fmaf(xh[cc], gv[cc], bev[cc]), 0.f);
    }
    if (c.drop_on) {
      const uint32_t rowkey = drop_rowkey(c.seed, c.layer_id, grow);
#pragma unroll
      for (int pp = 0; pp < (CC + 1) / 2; ++pp) {                 // columns lane + 128 pp and + 64: one hash
        const uint32_t hh = drop_hash(rowkey, lane + 64 * pp);
        v[2 * pp] = (hh & 0xffffu) >= c.drop_thr ? v[2 * pp] * c.keep_scale : 0.f;
        if (2 * pp + 1 < CC) v[2 * pp + 1] = (hh >> 16) >= c.drop_thr ? v[2 * pp + 1] * c.keep_scale : 0.f;
      }
    }
    float *np = nxt + row * ACT_LD + lane;
#pragma unroll
    for (int cc = 0; cc < CC; ++cc)
      if (FULL || okc[cc]) np[64 * cc] = v[cc];
    if constexpr (BF) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)      // every column of the image up to 255: zero beyond the layer's width
        abf[row * ABF_LD + lane + 64 * cc] = (cc < CC && okc[cc < CC ? cc : 0]) ? to_bf16(v[cc < CC ? cc : 0]) : (u16)0;
    }
    if (c.xhat != nullptr && grow < c.B) {        // both wave-uniform
      float *xp = c.xhat + (size_t)grow * h + lane, *ap = c.act + (size_t)grow * h + lane;
#pragma unroll
      for (int cc = 0; cc < CC; ++cc)
        if (FULL || okc[cc]) { xp[64 * cc] = xh[cc]; ap[64 * cc] = v[cc]; }
      if (c.ln_on && c.rstd != nullptr && lane == 0) c.rstd[grow] = rs[rr];
    }
  }
}

struct LnBwdCtx {
  float *dz;                     // [B][h] out
  int layer_id, h, B;
  uint64_t seed;
  float keep_scale;
  uint32_t drop_thr;
  bool drop_on, ln_on;
};

// Dropout -> ReLU -> LayerNorm backward of the wave's rows: dZ into `cur` (in place over dA), into the bf16 image and
// into global memory; the wave's column partials of dgamma / dbeta / db are ADDED into pg / pb / pz (entries >= CC
// untouched).  Staged over the rows like the forward phase: the masked gradient and the two row sums of every row,
// then the rows' reductions (independent chains), then dZ and its stores.
template <int RPW, int CC, bool FULL, bool BF>
static __device__ __forceinline__ void ln_bwd_rows(const LnBwdCtx &c, float *cur, u16 *abf, int wave, int lane, int row0,
                                                   const float (&gv)[4], const float (&bev)[4], const float (&xv)[RPW][4],
                                                   const float (&rsv)[RPW], float (&pg)[4], float (&pb)[4], float (&pz)[4]) {
  const int h = c.h;
  bool okc[CC];
#pragma unroll
  for (int cc = 0; cc < CC; ++cc) okc[cc] = FULL || lane + 64 * cc < h;
  const float inv_h = 1.0f / (float)h;
  float dxh[RPW][CC], m1[RPW], m2[RPW];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = RPW * wave + rr;
    const int grow = row0 + row;
    const bool valid = grow < c.B;                       // scalar
    const float *cp = cur + row * ACT_LD + lane;         // lane + 64 cc < 256 <= ACT_LD: in the row
    float d[CC];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) d[cc] = cp[64 * cc];
    if (c.drop_on) {
      const uint32_t rowkey = drop_rowkey(c.seed, c.layer_id, grow);
#pragma unroll
      for (int pp = 0; pp < (CC + 1) / 2; ++pp) {
        const uint32_t hh = drop_hash(rowkey, lane + 64 * pp);
        d[2 * pp] = (hh & 0xffffu) >= c.drop_thr ? d[2 * pp] * c.keep_scale : 0.f;
        if (2 * pp + 1 < CC) d[2 * pp + 1] = (hh >> 16) >= c.drop_thr ? d[2 * pp + 1] * c.keep_scale : 0.f;
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
      const float x = xv[rr][cc];
      const float u = fmaf(x, gv[cc], bev[cc]);
      float dd = (valid && okc[cc] && u > 0.f) ? d[cc] : 0.f;
      if (c.ln_on) {
        pg[cc] += dd * x;
        pb[cc] += dd;
        dd *= gv[cc];
        s1 += dd;
        s2 += dd * x;
      }
      dxh[rr][cc] = dd;
    }
    m1[rr] = s1; m2[rr] = s2;
  }
  if (c.ln_on) {
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) m1[rr] = wave_sum(m1[rr]) * inv_h;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) m2[rr] = wave_sum(m2[rr]) * inv_h;
  }
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = RPW * wave + rr;
    const int grow = row0 + row;
    const bool valid = grow < c.B;
    float dz[CC];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
      dz[cc] = (c.ln_on && valid) ? rsv[rr] * (dxh[rr][cc] - m1[rr] - xv[rr][cc] * m2[rr]) : dxh[rr][cc];
      if (FULL || okc[cc]) pz[cc] += dz[cc];
    }
    float *cw = cur + row * ACT_LD + lane;
#pragma unroll
    for (int cc = 0; cc < CC; ++cc)
      if (FULL || okc[cc]) cw[64 * cc] = dz[cc];
    if constexpr (BF) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)      // columns h .. 255 of the image: zero (64-deep chunks)
        abf[row * ABF_LD + lane + 64 * cc] = (cc < CC && okc[cc < CC ? cc : 0]) ? to_bf16(dz[cc < CC ? cc : 0]) : (u16)0;
    }
    if (valid) {
      float *gp = c.dz + (size_t)grow * h + lane;
#pragma unroll
      for (int cc = 0; cc < CC; ++cc)
        if (FULL || okc[cc]) gp[64 * cc] = dz[cc];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// BF: the layers' GEMMs take bf16 operands (fp32 accumulate; LayerNorm, loss and everything kept for the backward
// stay fp32).  LDS then holds ONE fp32 tile (z / LayerNorm in place; the head reads it) and the bf16 image of the
// current activations in the place of the second fp32 tile -- with D0 the fp32 pair stays (the features of layer 0
// use both halves) and the image follows them.
template <int MT, bool D0 = false, bool BF = false>
static __device__ __forceinline__ void tail_fwd_body(const TailFwdArgs &a, float *smem, float *red, const int tile) {
  constexpr int R = 16 * MT, RPW = (R + NW - 1) / NW;
  float *act0 = smem, *act1 = (BF && !D0) ? smem : smem + R * ACT_LD;
  u16 *abf = reinterpret_cast<u16 *>(smem + (D0 ? 2 : 1) * R * ACT_LD);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: wave-uniform branches
  const int q = lane >> 4, c16 = lane & 15;
  const int row0 = tile * R;
  // workgroups b, b + 8, ... share an XCD: consecutive ones of them start their K walks one chunk apart
  const int rot = a.krot ? (int)(blockIdx.x >> 3) : 0;
  STAMP(0);
  static_assert(!D0 || MAX_NI == 1, "the dense layer 0 inside the tail launch is built for 16 waves");
  BFrag<MAX_NI> wpre;
  BFragH<MAX_NI> wpre_h;
  if constexpr (D0) {
    if (a.d0.on == 1) {
      if (wave < (a.d0.L0.h >> 4)) d0_load_bfrag(wpre, a.d0.W0T, a.d0.L0.h, a.d0.L0.hp, 0, wave, c16, q);
      d0_fill_features<MT>(a.d0, act0, act1, row0, a.B);
    }
  } else {
    if (a.n_layers > 0) {
      if constexpr (BF) preload_wh(wpre_h, a.L[0].Wbf, a.L[0].h, a.L[0].hp, wave, c16, q, rot);
      else preload_w(wpre, a.L[0].W, a.L[0].h, a.L[0].hp, wave, c16, q, rot);
    }
  }
  if constexpr (!D0) {
    // input tile: unconditional loads from clamped rows (rows >= B duplicate the last row; nothing
    // computed for them is ever stored), 4 per thread in flight
    const int v4 = a.h_in >> 2;            // <= 64 float4 per row => R*v4 <= 64*R
    constexpr int NLD = 64 * R / TT;
    float4 tv[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = min(tid + TT * i, R * v4 - 1);
      const int row = idx / v4, c4 = idx - row * v4;
      tv[i] = *reinterpret_cast<const float4 *>(a.a_in + (size_t)min(row0 + row, a.B - 1) * a.h_in + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + TT * i;
      if (idx < R * v4) {
        const int row = idx / v4, c4 = idx - row * v4;
        *reinterpret_cast<float4 *>(act0 + row * ACT_LD + 4 * c4) = tv[i];
        if constexpr (BF)
          *reinterpret_cast<uint2 *>(abf + row * ABF_LD + 4 * c4) = make_uint2(pack_bf16(tv[i].x, tv[i].y), pack_bf16(tv[i].z, tv[i].w));
      }
    }
    if constexpr (BF) {
      // the image's columns h_in .. 255 are zero: a 64-deep chunk may reach past K
      const int z4 = (TAIL_MAX_W - a.h_in) >> 2;
      for (int idx = tid; idx < R * z4; idx += TT) {
        const int row = idx / z4, c4 = idx - row * z4;
        *reinterpret_cast<uint2 *>(abf + row * ABF_LD + a.h_in + 4 * c4) = make_uint2(0u, 0u);
      }
    }
  }
  lds_barrier();
  const uint64_t seed = a.seed + (a.step_dev ? (uint64_t)a.step_dev[0] * 0x9E3779B97F4A7C15ULL : 0ULL);
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const uint32_t drop_thr = drop_threshold(a.drop_p);
  const bool drop_on = a.drop_p > 0.f, ln_on = a.layernorm != 0;      // scalar
  float *cur = act0, *nxt = act1;
  STAMP(1);
  // output-layer weights of q = 0 (the MSE head has Q = 1): requested now, used at the very end
  const int hl_head = a.n_layers ? a.L[a.n_layers - 1].h : (D0 ? a.d0.L0.h : a.h_in);
  float wo0[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) wo0[cc] = a.Wo[min(lane + 64 * cc, hl_head - 1)];
  const float bo0 = a.bo[0];

  for (int li = D0 ? -1 : 0; li < a.n_layers; ++li) {
    const TailLayer &L = li < 0 ? a.d0.L0 : a.L[li];     // li = -1: layer 0 from the features in LDS
    const int h = L.h, hp = L.hp;
    const int NT = h >> 4;
    // LayerNorm parameters of this layer: requested before the GEMM, consumed after it (one L2 round
    // trip hidden); clamped columns, no conditional loads
    float gv[4], bev[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int colc = min(lane + 64 * cc, h - 1);
      gv[cc] = a.layernorm ? L.g[colc] : 1.f;
      bev[cc] = a.layernorm ? L.be[colc] : 0.f;
    }
    // this wave's bias columns: requested before the GEMM like the LayerNorm parameters (it used to be a dependent
    // L2 round trip between the last MFMA and the stores of z: ~1 us per layer at the tail of every wave)
    float bias_v[MAX_NI];
#pragma unroll
    for (int i = 0; i < MAX_NI; ++i) bias_v[i] = L.b[min(16 * (wave + NW * i), h - 16) + c16];
    f32x4 acc[MT][MAX_NI];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < MAX_NI; ++i) acc[mt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool parts = D0 && li < 0 && a.d0.on == 2;    // scalar
    if (parts) {
      // site x time grid: the pre-activation of layer 0 is the sum of a per-site and a per-time row
      const int h4 = h >> 2;
      for (int idx = tid; idx < R * h4; idx += TT) {
        const int row = idx / h4, c4 = idx - row * h4;
        const int grow = min(row0 + row, a.B - 1);
        const int ti = grow / a.d0.S, si = grow - ti * a.d0.S;
        const float4 u = *reinterpret_cast<const float4 *>(a.d0.sp + (size_t)si * h + 4 * c4);
        const float4 w = *reinterpret_cast<const float4 *>(a.d0.tp + (size_t)ti * h + 4 * c4);
        const float4 bb = *reinterpret_cast<const float4 *>(L.b + 4 * c4);
        *reinterpret_cast<float4 *>(nxt + row * ACT_LD + 4 * c4) =
            make_float4((u.x + w.x) + bb.x, (u.y + w.y) + bb.y, (u.z + w.z) + bb.z, (u.w + w.w) + bb.w);
      }
    } else if (D0 && li < 0) {
      if constexpr (D0) {
        d0_gemm<MT>(acc, act0, act1, a.d0.W0T, h, hp, wave, c16, q, wpre);
        if (hp > TAIL_MAX_W) lds_barrier();      // z goes into act1, which held the second half of the features
      }
    } else {
      if (li == 0) WSTAMP(0);
      if constexpr (BF) gemm16_pre_h<MT>(acc, abf, L.Wbf, h, hp, wave, c16, q, wpre_h, rot);
      else gemm16_pre<MT>(acc, cur, L.W, h, hp, wave, c16, q, wpre, rot);
      if (li == 0) WSTAMP(1);
    }
    if (li + 1 < a.n_layers) {
      if constexpr (BF) preload_wh(wpre_h, a.L[li + 1].Wbf, a.L[li + 1].h, a.L[li + 1].hp, wave, c16, q, rot);
      else preload_w(wpre, a.L[li + 1].W, a.L[li + 1].h, a.L[li + 1].hp, wave, c16, q, rot);
    }
    STAMP(2 + 4 * (li < 0 ? 0 : li));
    // z = acc + bias into the other activation buffer
#pragma unroll
    for (int i = 0; i < MAX_NI; ++i) {
      const int t = wave + NW * i;
      if (t < NT && !parts) {
        const int col = 16 * t + c16;
        const float bv = bias_v[i];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) nxt[(16 * mt + 4 * q + r) * ACT_LD + col] = acc[mt][i][r] + bv;
      }
    }
    if (li == 0) WSTAMP(2);
    lds_barrier();
    STAMP(3 + 4 * (li < 0 ? 0 : li));
    // LayerNorm -> ReLU -> Dropout of the wave's rows, specialised on the layer's width (ln_fwd_rows below)
    {
      const LnFwdCtx c{L.xhat, L.act, L.rstd, L.layer_id, h, a.B, a.eps, seed, keep_scale, drop_thr, drop_on, ln_on};
      if (h == 256) ln_fwd_rows<RPW, 4, true, BF>(c, nxt, abf, wave, lane, row0, gv, bev);
      else if (h == 128) ln_fwd_rows<RPW, 2, true, BF>(c, nxt, abf, wave, lane, row0, gv, bev);
      else ln_fwd_rows<RPW, 4, false, BF>(c, nxt, abf, wave, lane, row0, gv, bev);
    }
    STAMP(4 + 4 * (li < 0 ? 0 : li));
    lds_barrier();
    STAMP(5 + 4 * (li < 0 ? 0 : li));
    float *tmp = cur; cur = nxt; nxt = tmp;
  }

  // output layer (+ loss): y[row][qq] = a_last[row,:] . Wo[qq,:] + bo[qq]
  const int hl = hl_head;
  float lsum = 0.f;
  const bool plain_mse = a.loss.kind == STDADK_LOSS_MSE && a.loss.y_cols == a.Q;
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = RPW * wave + rr;
    const int grow = row0 + row;
    float mine = 0.f;              // lane q keeps prediction q of this row (general objectives)
    for (int qq = 0; qq < a.Q; ++qq) {
      float s = 0.f;
      if (qq == 0) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          if (lane + 64 * cc < hl) s = fmaf(cur[row * ACT_LD + lane + 64 * cc], wo0[cc], s);
      } else {
        for (int col = lane; col < hl; col += 64) s = fmaf(cur[row * ACT_LD + col], a.Wo[qq * hl + col], s);
      }
      s = wave_sum(s);
      const float yv = s + (qq == 0 ? bo0 : a.bo[qq]);
      mine = lane == qq ? yv : mine;
      if (lane == 0 && grow < a.B) {
        a.y_pred[(size_t)grow * a.Q + qq] = yv;
        if (a.y && plain_mse) {
          const float d = yv - a.y[(size_t)grow * a.Q + qq];
          lsum = fmaf(d, d, lsum);
          if (a.dY) a.dY[(size_t)grow * a.Q + qq] = 2.0f * d * a.grad_scale;
        }
      }
    }
    if (a.y && !plain_mse) {
      // check loss / non-crossing / broadcast targets: lane q owns element (row, q)
      const float yup = __shfl(mine, lane + 1, 64);
      const float ydn = __shfl(mine, lane - 1, 64);
      if (lane < a.Q && grow < a.B) {
        const float yt = a.y[(size_t)grow * a.loss.y_cols + (a.loss.y_cols == 1 ? 0 : lane)];
        float dy;
        lsum += loss_elem(a.loss, a.Q, lane, loss_tau(a.loss, lane), mine, yup, ydn, yt, a.grad_scale, dy);
        if (a.dY) a.dY[(size_t)grow * a.Q + lane] = dy;
      }
    }
  }
  STAMP(14);
  if (a.y && a.loss_sum) {
    lsum = wave_sum(lsum);
    if (lane == 0) red[wave] = lsum;
    lds_barrier();
    if (tid == 0) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w];
      atomicAdd(a.loss_sum, t);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward (data path): dA / dZ of every hidden layer + column partials for dgamma, dbeta, db
// ---------------------------------------------------------------------------------------------
// LDS floats of the backward body (the launch code sizes the dynamic segment with it)
template <int MT, bool BF>
static constexpr size_t tail_bwd_lds_floats() {
  constexpr size_t R = 16 * MT;
  if (BF) return R * ACT_LD + (R * ABF_LD + 1) / 2 + 3 * NW * 256 + R * TAIL_MAXQ;
  return 2 * R * ACT_LD + (R * ACT_LD >= (size_t)3 * NW * 256 ? 0 : 3 * NW * 256) + R * TAIL_MAXQ;
}

// Head of the backward: dY of the tile's rows into LDS (sdy [R][Q]), dA_last = dY Wo into d0, and the partials of
// dWo / dbo over the tile's rows into part_head[tile][Q*hl (dWo) | Q (dbo)].  `scr`: LDS scratch [TT/256][Q][256]
// (32 / 64 rows), free until the first LayerNorm phase.  Q1: one output column.
template <int MT, bool BF, bool Q1>
static __device__ __forceinline__ void bwd_head(const TailBwdArgs &a, float *d0, float *scr, float *sdy, int row0, int tile,
                                                int hl) {
  constexpr int R = 16 * MT;
  constexpr int QM = Q1 ? 1 : TAIL_MAXQ;
  const int tid = threadIdx.x;
  const int Q = Q1 ? 1 : a.Q;
  if (Q1) {
    if (tid < R) sdy[tid] = row0 + tid < a.B ? a.dY[row0 + tid] : 0.f;
  } else if (tid < R * TAIL_MAXQ) {
    const int row = tid / Q, qq = tid - row * Q;
    sdy[tid] = (tid < R * Q && row0 + row < a.B) ? a.dY[(size_t)(row0 + row) * Q + qq] : 0.f;
  }
  // this thread's column of Wo (requested before the barrier), then its column of dA for every NG-th row
  constexpr int NG = TT / 256;
  const int g = tid >> 8, col = tid & 255;
  const int colc = min(col, hl - 1);
  float wo[QM];
#pragma unroll
  for (int qq = 0; qq < QM; ++qq) wo[qq] = a.Wo[min(qq, Q - 1) * hl + colc];
  lds_barrier();
  if (col < hl) {
#pragma unroll
    for (int i = 0; i < R / NG; ++i) {
      const int r = g + NG * i;
      float d = 0.f;
#pragma unroll
      for (int qq = 0; qq < QM; ++qq)
        if (Q1 || qq < Q) d = fmaf(sdy[r * Q + qq], wo[qq], d);
      d0[r * ACT_LD + col] = d;
    }
  }
  float *ph = a.part_head + (size_t)tile * Q * (hl + 1);
  const int nrow = min(R, a.B - row0);
  if constexpr (MT == 1) {
    if (tid < hl) {
      float pw[QM];
#pragma unroll
      for (int qq = 0; qq < QM; ++qq) pw[qq] = 0.f;
      // 16 rows of the last activations in flight at a time (clamped rows: sdy is 0 beyond the batch)
      float av[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) av[r] = a.act_last[(size_t)min(row0 + r, a.B - 1) * hl + tid];
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int qq = 0; qq < QM; ++qq)
          if (Q1 || qq < Q) pw[qq] = fmaf(sdy[r * Q + qq], av[r], pw[qq]);
#pragma unroll
      for (int qq = 0; qq < QM; ++qq)
        if (Q1 || qq < Q) ph[qq * hl + tid] = pw[qq];
    }
  } else {
    // 32 / 64 rows: the column sums over the tile's rows by ALL waves -- groups of 256 threads take an equal share
    // of the rows each (one batch of loads in flight per thread), their partials meet in LDS (the second gradient
    // tile, or the column-partial scratch under bf16 operands: both unused until the first LayerNorm phase).  With
    // the columns alone (128 threads of 1024 walking 64 rows in four dependent batches) this phase was 14 of the
    // backward's 75 us per 64-row tile (round-2 stamps).
    constexpr int RG = R / NG;
    if (col < hl) {
      float pw[QM];
#pragma unroll
      for (int qq = 0; qq < QM; ++qq) pw[qq] = 0.f;
      float av[RG];
#pragma unroll
      for (int r = 0; r < RG; ++r) av[r] = a.act_last[(size_t)min(row0 + g * RG + r, a.B - 1) * hl + col];
#pragma unroll
      for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int qq = 0; qq < QM; ++qq)
          if (Q1 || qq < Q) pw[qq] = fmaf(sdy[(g * RG + r) * Q + qq], av[r], pw[qq]);
#pragma unroll
      for (int qq = 0; qq < QM; ++qq)
        if (Q1 || qq < Q) scr[(g * TAIL_MAXQ + qq) * 256 + col] = pw[qq];
    }
    lds_barrier();
    if (Q1) {
      if (tid < hl) {
        float t = 0.f;
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) t += scr[gg * TAIL_MAXQ * 256 + tid];     // fixed order
        ph[tid] = t;
      }
    } else {
      for (int i = tid; i < Q * hl; i += TT) {
        const int qq = i / hl, c2 = i - qq * hl;
        float t = 0.f;
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) t += scr[(gg * TAIL_MAXQ + qq) * 256 + c2];     // fixed order
        ph[qq * hl + c2] = t;
      }
    }
  }
  if (tid < Q) {
    float sb = 0.f;
    for (int row = 0; row < nrow; ++row) sb += sdy[row * Q + tid];
    ph[Q * hl + tid] = sb;
  }
}


// BF: dA = dZ W with bf16 operands (the transposed weight copies WTbf as the K-contiguous operand): ONE fp32
// gradient tile (LayerNorm backward in place, the GEMM's output lands in it again) + the bf16 image of dZ.
template <int MT, bool BF = false>
static __device__ __forceinline__ void tail_bwd_body(const TailBwdArgs &a, float *smem, const int tile) {
  constexpr int R = 16 * MT, RPW = (R + NW - 1) / NW;
  float *d0 = smem, *d1 = BF ? smem : smem + R * ACT_LD;
  u16 *abf = reinterpret_cast<u16 *>(smem + R * ACT_LD);
  // column-partial scratch [3][NW][256]: with 64 rows the spare activation buffer is large enough and
  // free during the LayerNorm phase, so it is aliased there instead of taking another 48 KiB
  constexpr bool RED_ALIAS = !BF && (size_t)R * ACT_LD >= (size_t)3 * NW * 256;
  float *red_own = BF ? smem + R * ACT_LD + (R * ABF_LD + 1) / 2 : smem + 2 * R * ACT_LD;
  float *sdy = red_own + (RED_ALIAS ? 0 : 3 * NW * 256);   // [R][TAIL_MAXQ]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, c16 = lane & 15;
  const int row0 = tile * R;
  const int rot = a.krot ? (int)(blockIdx.x >> 3) : 0;          // as in the forward body
  STAMP(0);
  if (tile == 0 && a.zero_ints)
    for (int i = tid; i < a.n_zero; i += (int)blockDim.x) a.zero_ints[i] = 0;
  // Global inputs of a layer's LayerNorm-backward phase: all loads issued together from clamped
  // addresses, unconditionally (one L2 round trip), and one phase EARLY — for the last layer right here
  // (hidden behind the head phase),
  // for layer li-1 just before the dA GEMM of pass li, so the round trip hides behind the MFMA work.
  float gv[4], bev[4], xv[RPW][4], rsv[RPW];
  auto ln_inputs = [&](const TailLayer &Ln) {
    const int hn = Ln.h;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int colc = min(lane + 64 * cc, hn - 1);
      gv[cc] = a.layernorm ? Ln.g[colc] : 1.f;
      bev[cc] = a.layernorm ? Ln.be[colc] : 0.f;
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int growc = min(row0 + RPW * wave + rr, a.B - 1);
        xv[rr][cc] = Ln.xhat[(size_t)growc * hn + colc];
      }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) rsv[rr] = a.layernorm ? Ln.rstd[min(row0 + RPW * wave + rr, a.B - 1)] : 1.f;
  };
  ln_inputs(a.L[a.n_layers - 1]);

  const int hl = a.L[a.n_layers - 1].h;
  // head phase: dA of the last hidden layer and the output layer's weight-gradient partials of this tile; one
  // output column (the MSE head) is its own instantiation -- with a run-time Q every fused multiply-add of the
  // column sums sat behind a scalar branch
  if (a.Q == 1) bwd_head<MT, BF, true>(a, d0, BF ? red_own : d1, sdy, row0, tile, hl);
  else bwd_head<MT, BF, false>(a, d0, BF ? red_own : d1, sdy, row0, tile, hl);
  lds_barrier();
  const uint64_t seed = a.seed + (a.step_dev ? (uint64_t)a.step_dev[0] * 0x9E3779B97F4A7C15ULL : 0ULL);
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const uint32_t drop_thr = drop_threshold(a.drop_p);
  const bool drop_on = a.drop_p > 0.f, ln_on = a.layernorm != 0;      // scalar
  float *cur = d0, *nxt = d1;
  STAMP(1);                     // head phase done (dA of the last layer, dWo / dbo partials)

  for (int li = a.n_layers - 1; li >= 0; --li) {
    const TailLayer &L = a.L[li];
    const int h = L.h;
    BFrag<MAX_NI> wpre;
    BFragH<MAX_NI> wpre_h;
    if (li > 0) {                                                       // for the dA GEMM at the end of this pass
      if constexpr (BF) preload_wh(wpre_h, L.WTbf, L.hp, h, wave, c16, q, rot);
      else preload_w<true>(wpre, L.W, L.hp, h, wave, c16, q, rot);
    }
    // ---- (a) Dropout -> ReLU -> LayerNorm backward of the wave's rows, specialised on the layer's width
    // (ln_bwd_rows below).  Its global inputs (xhat rows, gamma, beta, rstd) were requested one phase early (ln_inputs).
    float pg[4], pb[4], pz[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) pg[cc] = pb[cc] = pz[cc] = 0.f;
    {
      const LnBwdCtx c{a.dZ[li], L.layer_id, h, a.B, seed, keep_scale, drop_thr, drop_on, ln_on};
      if (h == 256) ln_bwd_rows<RPW, 4, true, BF>(c, cur, abf, wave, lane, row0, gv, bev, xv, rsv, pg, pb, pz);
      else if (h == 128) ln_bwd_rows<RPW, 2, true, BF>(c, cur, abf, wave, lane, row0, gv, bev, xv, rsv, pg, pb, pz);
      else ln_bwd_rows<RPW, 4, false, BF>(c, cur, abf, wave, lane, row0, gv, bev, xv, rsv, pg, pb, pz);
    }
    // column partials of this workgroup's rows (`nxt` is not read again before the GEMM below refills it)
    float *red = RED_ALIAS ? nxt : red_own;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      red[(0 * NW + wave) * 256 + lane + 64 * cc] = pg[cc];
      red[(1 * NW + wave) * 256 + lane + 64 * cc] = pb[cc];
      red[(2 * NW + wave) * 256 + lane + 64 * cc] = pz[cc];
    }
    lds_barrier();
    if (tid < h) {
      float *pbase = a.part[li] + (size_t)tile * 3 * h;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[(k * NW + w) * 256 + tid];
        pbase[k * h + tid] = t;
      }
    }
    STAMP(2 + 3 * (a.n_layers - 1 - li));      // LayerNorm-backward phase + column partials of this pass
    if (li == 0) break;   // (the partials above are published by the barrier inside part (b) / kernel end)
    // ---- (b) dA_prev[16 x hp] = dZ[16 x h] W[h x hp], with W^T ([hp][h], K contiguous) as the B operand
    const int hp = L.hp;
    const int NT = hp >> 4;
    f32x4 acc[MT][MAX_NI];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < MAX_NI; ++i) acc[mt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ln_inputs(a.L[li - 1]);        // consumed after the GEMM, in the next pass
    lds_barrier();                 // every wave's dZ rows are in `cur`
    if constexpr (BF) gemm16_pre_h<MT>(acc, abf, L.WTbf, hp, h, wave, c16, q, wpre_h, rot);
    else gemm16_pre<MT, true>(acc, cur, L.W, hp, h, wave, c16, q, wpre, rot);
    STAMP(3 + 3 * (a.n_layers - 1 - li));      // dA GEMM of this pass
#pragma unroll
    for (int i = 0; i < MAX_NI; ++i) {
      const int t = wave + NW * i;
      if (t < NT) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) nxt[(16 * mt + 4 * q + r) * ACT_LD + 16 * t + c16] = acc[mt][i][r];
      }
    }
    lds_barrier();
    STAMP(4 + 3 * (a.n_layers - 1 - li));      // dA tile stored + barrier
    float *tmp = cur; cur = nxt; nxt = tmp;
  }
}

};  // struct Tail<NW>

}  // namespace stdadk
