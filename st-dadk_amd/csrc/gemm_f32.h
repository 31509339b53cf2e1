// fp32 MFMA GEMM used by the MLP layers (exact fp32: v_mfma_f32_32x32x2_f32 is a k-ordered fma chain).
#pragma once
#include "common.h"

namespace stdadk {

// C[M,N] (+bias[N]) = sum_k Aop[m][k] * Bop[k][n]
//   a_km == false : Aop[m][k] = A[m*lda + k]   (row-major M x K, K contiguous)
//   a_km == true  : Aop[m][k] = A[k*lda + m]   (stored K x M, reduction index is the row)
//   b_km == false : Bop[k][n] = B[n*ldb + k]   (stored N x K — nn.Linear weight layout)
//   b_km == true  : Bop[k][n] = B[k*ldb + n]   (stored K x N)
// splits > 1: split s covers k in [s*kps, (s+1)*kps) and writes its partial tile to
//   slab + s*slab_stride (row stride N, no bias); the caller sums the slabs.
struct GemmArgs {
  const float *A;
  int64_t lda;
  const float *B;
  int64_t ldb;
  float *C;
  int64_t ldc;
  const float *bias;  // may be NULL; only applied when splits == 1
  int M, N, K;
  int splits;         // >= 1
  int kps;            // K per split, multiple of 32 (ignored when splits == 1)
  float *slab;        // splits * slab_stride floats when splits > 1
  int64_t slab_stride;
  int always_slab = 0;  // write the (single) partial to the slab even when splits == 1
  int bf16 = 0;         // grouped TN jobs only (STDADK_FLAG_BF16): operands rounded to bf16 as they enter LDS,
                        // v_mfma_f32_32x32x16_bf16, fp32 accumulate
  int xcd_split = 0;    // grouped TN jobs only, set by gemm_tn_grouped_prepare: XCD-aware block order (splits % 8 == 0)
  int coherent_slab = 0;  // the slab tiles are written with device-scope (write-through) stores: a workgroup on another
                          // XCD sums them later in the SAME launch (FinArgs), set by launch_dw_all
};

// a table of TN products for one launch (gemm_tn_grouped_kernel) and of partial-sum reductions
constexpr int GEMM_GROUP_MAX = 8;
struct GemmGroup {
  int n = 0;
  GemmArgs job[GEMM_GROUP_MAX];
  int first_block[GEMM_GROUP_MAX + 1];
};
struct ReduceJob {
  const float *src;   // src[s*stride + i]
  float *dst;         // dst[i] = sum_s
  int n, splits;
  int64_t stride;
  int wide = 0;       // set by launch_reduce_jobs
};
constexpr int REDUCE_GROUP_MAX = 40;
constexpr int REDUCE_SQ_PARTS = 512;   // = STDADK_GRADSQ_PARTS
struct ReduceGroup {
  int n = 0;
  ReduceJob job[REDUCE_GROUP_MAX];
  int first_block[REDUCE_GROUP_MAX + 1];
  // optional: partial sums of squares of what this launch writes -- sq_parts[b] for reduce workgroup b (as many
  // entries as reduce_jobs_block_count() returns) -- and of an already final region sq_src[0..sq_n) --
  // sq_region_parts[0..256), by 256 extra workgroups; together the squared norm of a whole gradient
  float *sq_parts = nullptr;
  float *sq_region_parts = nullptr;
  const float *sq_src = nullptr;
  int64_t sq_n = 0;
  int *step_inc = nullptr;       // advanced by one (device step counter), or NULL
  int n_reduce_blocks = 0;       // set by launch_reduce_jobs
};
// Finishing work folded into the merged weight-gradient launch (dw_all.hip) instead of a reductions launch behind
// it: the workgroup that is LAST to deliver its K slice of an output tile (arrival counter per tile) sums the tile's
// slabs in slice order -- the sum the wide reduce job would have taken -- writes C and leaves the squares of what
// it wrote in the tile's slot; the column-partial ("tall") reduce jobs ride along as extra workgroups; every knot
// workgroup leaves the squares of its rows of dW0^T.  The slots replace the squared-norm partials of the
// reductions launch (the optimiser sums them in slot order: fixed order, no atomics on floats).
constexpr int FIN_TILES_MAX = 1024;
struct FinArgs {
  int *cnt = nullptr;         // [FIN_TILES_MAX] arrival counters, zero when the launch starts (the step's tail kernel
                              // clears them, TailBwdArgs::zero_ints); NULL = the products' slabs and the tall jobs are
                              // left to a reductions launch (n_tiles = n_tall = 0), only the knot slots are written
  float *slots = nullptr;     // squared-norm partials or NULL: [0, n_tiles) output tiles, [n_tiles, n_tiles + n_tall)
                              // tall reduce workgroups, then one per knot workgroup
  int tile0[GEMM_GROUP_MAX];  // first counter / slot of every grouped job
  int n_tiles = 0, n_tall = 0;
  int *step_inc = nullptr;    // advanced by one (device step counter), or NULL
};
int reduce_jobs_block_count(ReduceGroup &grp);    // fills first_block / wide; the launch's reduce workgroups
int launch_gemm_tn_grouped(GemmGroup &grp, hipStream_t st);     // every job: a_km = b_km = true, aligned
int gemm_tn_grouped_prepare(GemmGroup &grp, int *n_blocks);     // fills first_block; for launches that embed the group
int launch_reduce_jobs(ReduceGroup &grp, hipStream_t st);
bool gemm_tn_groupable(const float *A, int64_t lda, const float *B, int64_t ldb);

// Picks a split count so the launch has >= ~256 workgroups; returns splits and sets kps.
int gemm_pick_splits(int M, int N, int K, int *kps, bool big_tile);
bool gemm_use_big_tile(int M, int N);
int launch_gemm_f32(const GemmArgs &g, bool a_km, bool b_km, hipStream_t st);

// Floats of split-K slab workspace gemm_run() needs for this shape (0 when it does not split).
size_t gemm_slab_floats(int M, int N, int K);

// C = Aop * Bop (+bias): picks tile and split, launches, and sums the partial slabs into C unless
// `keep_slabs` (then the caller's next kernel consumes `*splits_out` slabs of M*N floats itself;
// with one split the un-biased product is in C).
int gemm_run(const float *A, int64_t lda, bool a_km, const float *Bm, int64_t ldb, bool b_km, int M, int N,
             int K, const float *bias, float *C, int64_t ldc, float *slab, bool keep_slabs,
             int *splits_out, hipStream_t st);

}  // namespace stdadk
