// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32; exact fp32, 64 FLOP/clk/SIMD).
//
// One workgroup = 4 waves (2x2) computes a BM x BN tile of C, each wave (BM/2)x(BN/2) as
// (BM/64)x(BN/64) MFMA tiles of 32x32.  K is walked in steps of BK = 32 through one LDS stage with
// register prefetch of the next step (global -> VGPR during the MFMAs of the current step).
//
// LDS images and fragment reads (bank rules: MI355X guide, "LDS"):
//   K-contiguous operand  (rows x BK, row stride BK+4 floats): one ds_read_b128 per lane gives 4
//     consecutive k of its row; the 144-byte row stride spreads the 16 lanes of a b128 group over
//     all 64 banks (conflict-free).
//   K-major operand (BK x rows, row stride = rows): ds_read_b32, 32 lanes read 32 consecutive
//     floats of one k-row (conflict-free).
// The MFMA sums k over the two lane halves (k = l>>5); a lane half h uses k = 8*s + 4*h + e for
// MFMA e of sub-step s, for BOTH operands, so any k order is fine as long as A and B agree.
#include "gemm_f32.h"
#include "gemm_body.h"

namespace stdadk {

template <int BM, int BN, bool A_KM, bool B_KM, int VA, int VB>
__global__ __launch_bounds__(GT) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[TileGeom<BM, A_KM>::SIZE + TileGeom<BN, B_KM>::SIZE];
  gemm_tile_body<BM, BN, A_KM, B_KM, VA, VB>(g, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

// Several independent C_j = A_j^T B_j products (reduction over the batch) in ONE launch: block ->
// (job, tile, split) through a prefix table; every job writes split-K slabs.
__global__ __launch_bounds__(GT) void gemm_tn_grouped_kernel(GemmGroup grp) {
  __shared__ __attribute__((aligned(16))) float lds[GROUP_LDS_FLOATS];
  gemm_tn_grouped_block(grp, (int)blockIdx.x, lds);
}

// dst_j[i] = sum_s src_j[s*stride_j + i], i < n_j, for a table of jobs, fixed summation order.
//   tall jobs (many partials, few columns: per-workgroup column partials): 16 columns x 16 s-groups
//   wide jobs (few partials, many columns: split-K slabs): one float4 of columns per thread, the
//              (<= 64) partials summed serially with independent, coalesced loads
__global__ __launch_bounds__(256) void reduce_jobs_kernel(ReduceGroup grp) {
  __shared__ float red[4][64];
  __shared__ float sqred[4];
  const int nrb = grp.n_reduce_blocks;
  if (blockIdx.x == 0 && threadIdx.x == 0 && grp.step_inc) grp.step_inc[0] += 1;
  if ((int)blockIdx.x >= nrb) {
    // squared-norm partials of the already final region (sq_region_parts != NULL only): block b of 256
    const int b = (int)blockIdx.x - nrb;
    float acc = 0.f;
    const int64_t stride = 256ll * 256;
    const int64_t i = (int64_t)b * 256 + threadIdx.x;
    const int64_t n4 = grp.sq_n / 4;
    const float4 *g4 = reinterpret_cast<const float4 *>(grp.sq_src);      // 16-byte aligned (checked by the host)
    for (int64_t j = i; j < n4; j += stride) {
      const float4 v = g4[j];
      acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc);
      acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
    }
    for (int64_t j = n4 * 4 + i; j < grp.sq_n; j += stride) acc = fmaf(grp.sq_src[j], grp.sq_src[j], acc);
    const float s = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sqred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) grp.sq_region_parts[b] = (sqred[0] + sqred[1]) + (sqred[2] + sqred[3]);
    return;
  }
  const float sq = reduce_job_block(grp, (int)blockIdx.x, &red[0][0]);   // squares of what this thread wrote
  if (grp.sq_parts) {              // workgroup-uniform
    const float s = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) sqred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) grp.sq_parts[blockIdx.x] = (sqred[0] + sqred[1]) + (sqred[2] + sqred[3]);
  }
}

int gemm_tn_grouped_prepare(GemmGroup &grp, int *n_blocks) {
  int nb = 0;
  for (int j = 0; j < grp.n; ++j) {
    GemmArgs &g = grp.job[j];
    STDADK_REQUIRE(g.slab && g.kps % BK == 0 && g.splits >= 1, STDADK_E_ARG, "grouped gemm: bad job %d", j);
    g.always_slab = 1;
    // XCD-aware order of a job's blocks (gemm_tn_grouped_block) when its K slices deal evenly to the 8 XCDs; its
    // first block is then a multiple of 8 (the blocks before it are padding that returns at once).
    // Environment STDADK_GEMM_XCD=0: plain order (diagnostic).
    static const bool xcd_off = [] { const char *e = getenv("STDADK_GEMM_XCD"); return e && e[0] == '0'; }();
    g.xcd_split = (!xcd_off && g.splits >= 8 && g.splits % 8 == 0) ? 1 : 0;
    if (g.xcd_split) nb = (nb + 7) & ~7;
    grp.first_block[j] = nb;
    nb += (int)(ceil_div(g.M, 64) * ceil_div(g.N, 64)) * g.splits;
  }
  *n_blocks = nb;
  return 0;
}

int launch_gemm_tn_grouped(GemmGroup &grp, hipStream_t st) {
  int nb = 0;
  int rc = gemm_tn_grouped_prepare(grp, &nb);
  if (rc) return rc;
  if (nb == 0) return 0;
  STDADK_LAUNCH(gemm_tn_grouped_kernel, dim3((unsigned)nb), dim3(GT), 0, st, grp);
  STDADK_CHECK_LAUNCH("gemm_tn_grouped");
  return 0;
}

int reduce_jobs_block_count(ReduceGroup &grp) {
  int nb = 0;
  for (int j = 0; j < grp.n; ++j) {
    ReduceJob &jb = grp.job[j];
    jb.wide = jb.splits <= 64 && jb.n % 4 == 0 && jb.stride % 4 == 0 &&
              ((reinterpret_cast<uintptr_t>(jb.src) | reinterpret_cast<uintptr_t>(jb.dst)) & 15) == 0;
    grp.first_block[j] = nb;
    nb += (int)ceil_div(jb.n, jb.wide ? 1024 : REDUCE_TALL_COLS);
  }
  grp.n_reduce_blocks = nb;
  return nb;
}

int launch_reduce_jobs(ReduceGroup &grp, hipStream_t st) {
  int nb = reduce_jobs_block_count(grp);
  STDADK_REQUIRE(!grp.step_inc || nb > 0, STDADK_E_ARG, "reduce_jobs: the step counter rides on the first reduce workgroup");
  if (grp.sq_region_parts) {
    STDADK_REQUIRE(grp.sq_src && (reinterpret_cast<uintptr_t>(grp.sq_src) & 15) == 0, STDADK_E_ARG,
                   "reduce_jobs: squared-norm partials of a region need an aligned region");
    nb += 256;
  }
  if (nb == 0) return 0;
  STDADK_LAUNCH(reduce_jobs_kernel, dim3((unsigned)nb), dim3(256), 0, st, grp);
  STDADK_CHECK_LAUNCH("reduce_jobs");
  return 0;
}

bool gemm_tn_groupable(const float *A, int64_t lda, const float *B, int64_t ldb) {
  return ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 && lda % 4 == 0 && ldb % 4 == 0;
}

static int vec_of(const float *p, int64_t ld) {
  uintptr_t a = reinterpret_cast<uintptr_t>(p);
  if ((a & 15) == 0 && ld % 4 == 0) return 4;
  if ((a & 7) == 0 && ld % 2 == 0) return 2;
  return 1;
}

bool gemm_use_big_tile(int M, int N) {
  // 128x128 tiles only when they alone give >= 128 workgroups; else 64x64 for parallelism
  return (int64_t)ceil_div(M, 128) * ceil_div(N, 128) >= 128;
}

int gemm_pick_splits(int M, int N, int K, int *kps, bool big) {
  const int t = big ? 128 : 64;
  int64_t tiles = ceil_div(M, t) * ceil_div(N, t);
  if (tiles < 1) tiles = 1;                 // an empty product still gets a (trivial) plan
  int s = 1;
  if (tiles < 256) s = (int)ceil_div(256, tiles);
  int max_s = (int)ceil_div(K, 4 * BK);   // at least 4 k-steps of work per split
  if (max_s < 1) max_s = 1;
  if (s > max_s) s = max_s;
  if (s > 64) s = 64;
  int per = (int)ceil_div(ceil_div(K, s), BK) * BK;
  if (per < BK) per = BK;
  s = (int)ceil_div(K, per);
  if (s < 1) s = 1;
  *kps = per;
  return s;
}

template <int BM, int BN, bool A_KM, bool B_KM>
static void launch_v(const GemmArgs &g, int va, int vb, hipStream_t st) {
  dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)g.splits);
  char name[96] = "gemm_f32_kernel";
  if (g_prof_on)
    snprintf(name, sizeof(name), "gemm_f32_kernel<%d,%d,%s%s> M=%d N=%d K=%d splits=%d", BM, BN,
             A_KM ? "T" : "N", B_KM ? "N" : "T", g.M, g.N, g.K, g.splits);
#define L(VA_, VB_) STDADK_LAUNCH_NAMED(name, (gemm_f32_kernel<BM, BN, A_KM, B_KM, VA_, VB_>), grid, dim3(GT), 0, st, g)
  if (va == 4 && vb == 4) L(4, 4);
  else if (va == 4 && vb == 2) L(4, 2);
  else if (va == 4 && vb == 1) L(4, 1);
  else if (va == 2 && vb == 4) L(2, 4);
  else if (va == 2 && vb == 2) L(2, 2);
  else if (va == 2 && vb == 1) L(2, 1);
  else if (va == 1 && vb == 4) L(1, 4);
  else if (va == 1 && vb == 2) L(1, 2);
  else L(1, 1);
#undef L
}

int launch_gemm_f32(const GemmArgs &g, bool a_km, bool b_km, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0) return 0;
  STDADK_REQUIRE(g.splits >= 1 && (g.splits == 1 || (g.slab && g.kps % BK == 0)), STDADK_E_ARG,
                 "gemm: bad split configuration");
  const int va = vec_of(g.A, g.lda), vb = vec_of(g.B, g.ldb);
  const bool big = gemm_use_big_tile(g.M, g.N);
  if (big) {
    if (!a_km && !b_km) launch_v<128, 128, false, false>(g, va, vb, st);
    else if (!a_km && b_km) launch_v<128, 128, false, true>(g, va, vb, st);
    else if (a_km && b_km) launch_v<128, 128, true, true>(g, va, vb, st);
    else STDADK_REQUIRE(false, STDADK_E_ARG, "gemm: layout (a_km, !b_km) not built");
  } else {
    if (!a_km && !b_km) launch_v<64, 64, false, false>(g, va, vb, st);
    else if (!a_km && b_km) launch_v<64, 64, false, true>(g, va, vb, st);
    else if (a_km && b_km) launch_v<64, 64, true, true>(g, va, vb, st);
    else STDADK_REQUIRE(false, STDADK_E_ARG, "gemm: layout (a_km, !b_km) not built");
  }
  STDADK_CHECK_LAUNCH("gemm_f32");
  return 0;
}


// out[i] = sum_s slab[s*stride + i] (+ bias[i % ncol])
__global__ void slab_sum_kernel(const float *__restrict__ slab, int splits, int64_t stride, int64_t n,
                                const float *__restrict__ bias, int ncol, float *__restrict__ out,
                                int64_t ld_out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = 0.f;
  for (int s = 0; s < splits; ++s) v += slab[(int64_t)s * stride + i];
  int64_t r = i / ncol;
  int c = (int)(i - r * ncol);
  if (bias) v += bias[c];
  out[r * ld_out + c] = v;
}


// C = Aop * Bop (+bias) through the split-K GEMM; partial slabs are summed into C unless
// `keep_slabs` (the caller's next kernel consumes the slabs itself).
int gemm_run(const float *A, int64_t lda, bool a_km, const float *Bm, int64_t ldb, bool b_km,
                   int M, int N, int K, const float *bias, float *C, int64_t ldc, float *slab,
                   bool keep_slabs, int *splits_out, hipStream_t st) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.bias = bias;
  g.M = M; g.N = N; g.K = K;
  bool big = gemm_use_big_tile(M, N);
  g.splits = gemm_pick_splits(M, N, K, &g.kps, big);
  g.slab = slab;
  g.slab_stride = (int64_t)M * N;
  if (keep_slabs && g.splits == 1) {  // consumer reads "1 slab" = C itself, bias added by consumer
    g.bias = nullptr;
  }
  int rc = launch_gemm_f32(g, a_km, b_km, st);
  if (rc) return rc;
  if (g.splits > 1 && !keep_slabs) {
    int64_t n = (int64_t)M * N;
    STDADK_LAUNCH(slab_sum_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, slab,
                       g.splits, g.slab_stride, n, bias, N, C, ldc);
    STDADK_CHECK_LAUNCH("slab_sum");
  }
  if (splits_out) *splits_out = g.splits;
  return 0;
}


size_t gemm_slab_floats(int M, int N, int K) {
  int kps;
  bool big = gemm_use_big_tile(M, N);
  int s = gemm_pick_splits(M, N, K, &kps, big);
  return s > 1 ? (size_t)s * M * N : 0;
}

}  // namespace stdadk

using namespace stdadk;

extern "C" size_t stdadk_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return align_up(gemm_slab_floats(M, N, K) * sizeof(float), 256);
}

extern "C" int stdadk_gemm_f32(const float *A, int64_t lda, int32_t a_km, const float *B, int64_t ldb,
                               int32_t b_km, int32_t M, int32_t N, int32_t K, const float *bias, float *C,
                               int64_t ldc, void *workspace, size_t workspace_bytes,
                               stdadk_stream_t stream) {
  STDADK_REQUIRE(M >= 0 && N >= 0 && K >= 0, STDADK_E_ARG, "gemm: negative size");
  if (M == 0 || N == 0) return 0;
  STDADK_REQUIRE(A && B && C, STDADK_E_ARG, "gemm: NULL pointer");
  STDADK_REQUIRE(!(a_km && !b_km), STDADK_E_ARG, "gemm: layout (a_km, !b_km) not built");
  STDADK_REQUIRE(ldc >= N, STDADK_E_SHAPE, "gemm: ldc < N");
  size_t need = gemm_slab_floats(M, N, K) * sizeof(float);
  STDADK_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), STDADK_E_WORKSPACE,
                 "gemm: workspace %zu < %zu bytes", workspace_bytes, need);
  return gemm_run(A, lda, a_km != 0, B, ldb, b_km != 0, M, N, K, bias, C, ldc, (float *)workspace, false,
                  nullptr, (hipStream_t)stream);
}
