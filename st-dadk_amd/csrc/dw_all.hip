// Every weight gradient of a training step that is a reduction over the batch, in ONE launch: the grouped
// split-K products dW_l = dZ_l^T a_(l-1) of the layers after the first (+ the temporal / covariate rows of
// dW0^T) on the matrix cores, and the per-knot gather of the spatial rows of dW0^T.  The two kinds of
// workgroup are independent (both only need the dZ of the backward chain), have the same shape (256
// threads) and complementary bottlenecks (staged MFMA tiles vs L2 row gathers), so sharing the CUs beats
// running them one after the other, and one launch + drain is saved.
#include <stdlib.h>

#include "gemm_body.h"
#include "l1_bwd_body.h"

namespace stdadk {

static_assert(GT == BW_T, "the GEMM tiles and the knot groups must share the workgroup shape");

// NK (fixed knots only): 2 neighbouring knots per wave, see l1_window_bwd_multi_body; 1: one knot per wave
// fin.cnt != NULL (FinArgs, gemm_f32.h): the launch also does what the reductions launch behind it did -- block
// order [GEMM tiles | tall reduce jobs | padding | knot groups]
template <int CPL, int BASIS, bool KNOTS, int NK>
__global__ __launch_bounds__(GT) void dw_all_kernel(GemmGroup grp, int n_gemm_blocks, L1BwdArgs a, FinArgs fin,
                                                    ReduceGroup tall) {
  __shared__ __attribute__((aligned(16))) float lds[GROUP_LDS_FLOATS];
  // GEMM tiles take the low block ids (dispatched first): measured 33.5 us vs 42.5 us the other way round
  if ((int)blockIdx.x < n_gemm_blocks) { gemm_tn_grouped_block(grp, (int)blockIdx.x, lds, &fin); return; }
  const int n_front = n_gemm_blocks + fin.n_tall;
  if ((int)blockIdx.x < n_front) {
    const int tb = (int)blockIdx.x - n_gemm_blocks;
    if (tb == 0 && threadIdx.x == 0 && fin.step_inc) fin.step_inc[0] += 1;
    const float sq = reduce_job_block(tall, tb, lds);
    if (fin.slots) {
      const float t = block4_sum(sq, lds + 256);
      if (threadIdx.x == 0) fin.slots[fin.n_tiles + tb] = t;
    }
    return;
  }
  // XCD-striped knot groups start at a multiple of 8, so that (group block & 7) is the XCD of the workgroup
  const int first = a.xcd_slots > 0 ? (n_front + 7) & ~7 : n_front;
  if ((int)blockIdx.x < first) return;
  float sq;
  if constexpr (NK > 1) sq = l1_window_bwd_multi_body<CPL, BASIS, NK>(a, (int)blockIdx.x - first);
  else sq = l1_window_bwd_body<CPL, BASIS, KNOTS>(a, (int)blockIdx.x - first);
  if (fin.slots) {                 // workgroup-uniform; every wave of the workgroup arrives (no wave exits early)
    const float t = block4_sum(sq, lds);
    if (threadIdx.x == 0) fin.slots[fin.n_tiles + fin.n_tall + ((int)blockIdx.x - first)] = t;
  }
}

// kernel arguments travel in the 4 KiB kernarg segment
static_assert(sizeof(GemmGroup) + sizeof(L1BwdArgs) + sizeof(FinArgs) + sizeof(ReduceGroup) + 16 <= 4096,
              "dw_all: kernel arguments exceed the kernarg segment");

int dw_all_knot_blocks(const L1BwdArgs &a_in) {
  L1BwdArgs a = a_in;
  const int nk = knots_per_wave(a);
  const int slots = knot_xcd_slots(a.g, a.kpart ? 1 : nk);
  return slots > 0 ? 8 * slots : (int)ceil_div(knot_group_count(a.g, nk), BW_T / 64);
}

int launch_dw_all(GemmGroup &grp, const L1BwdArgs &a_in, int basis, hipStream_t st, const FinArgs *fin_in,
                  ReduceGroup *tall_in, int *n_slots) {
  L1BwdArgs a = a_in;
  STDADK_REQUIRE(a.G <= 256, STDADK_E_ARG, "dw_all: G too large");
  STDADK_REQUIRE((int64_t)a.B * a.H < (1ll << 32), STDADK_E_ARG, "dw_all: B*H exceeds 32-bit offsets");
  STDADK_REQUIRE(!a.kpart || a.W0T, STDADK_E_ARG, "dw_all: knot sums need W0^T");
  int ng = 0;
  int rc = gemm_tn_grouped_prepare(grp, &ng);
  if (rc) return rc;
  const int nk = knots_per_wave(a);
  a.xcd_slots = knot_xcd_slots(a.g, a.kpart ? 1 : nk);
  const unsigned n_knot = a.xcd_slots > 0 ? 8u * (unsigned)a.xcd_slots
                                          : (unsigned)ceil_div(knot_group_count(a.g, nk), BW_T / 64);
  FinArgs fin;
  ReduceGroup tall;
  if (fin_in && fin_in->cnt) {
    STDADK_REQUIRE(tall_in, STDADK_E_ARG, "dw_all: finishing work without its reduce table");
    fin = *fin_in;
    tall = *tall_in;
    for (int j = 0; j < grp.n; ++j) grp.job[j].coherent_slab = 1;
    fin.n_tall = reduce_jobs_block_count(tall);
    int nt = 0;
    for (int j = 0; j < grp.n; ++j) {
      fin.tile0[j] = nt;
      nt += (int)(ceil_div(grp.job[j].M, 64) * ceil_div(grp.job[j].N, 64));
    }
    STDADK_REQUIRE(nt <= FIN_TILES_MAX, STDADK_E_ARG, "dw_all: %d output tiles (at most %d)", nt, FIN_TILES_MAX);
    fin.n_tiles = nt;
    if (n_slots) *n_slots = nt + fin.n_tall + (int)n_knot;
  } else if (fin_in && fin_in->slots) {      // only the knot workgroups' squared-norm slots
    fin.slots = fin_in->slots;
    if (n_slots) *n_slots = (int)n_knot;
  }
  const unsigned front = (unsigned)ng + (unsigned)fin.n_tall;
  const unsigned grid = (a.xcd_slots > 0 ? ((front + 7u) & ~7u) : front) + n_knot;
#define GO(CPL_, BS_)                                                                                      \
  do {                                                                                                     \
    if (a.kpart) STDADK_LAUNCH_NAMED("dw_all_kernel<knots>", (dw_all_kernel<CPL_, BS_, true, 1>),          \
                                     dim3(grid), dim3(GT), 0, st, grp, ng, a, fin, tall);                  \
    else if (nk == 2) STDADK_LAUNCH_NAMED("dw_all_kernel", (dw_all_kernel<CPL_, BS_, false, 2>), dim3(grid), \
                                          dim3(GT), 0, st, grp, ng, a, fin, tall);                         \
    else STDADK_LAUNCH_NAMED("dw_all_kernel", (dw_all_kernel<CPL_, BS_, false, 1>), dim3(grid),            \
                             dim3(GT), 0, st, grp, ng, a, fin, tall);                                      \
  } while (0)
  if (a.H == 256) { if (basis == STDADK_BASIS_WENDLAND) GO(4, 0); else GO(4, 2); }
  else if (a.H == 128) { if (basis == STDADK_BASIS_WENDLAND) GO(2, 0); else GO(2, 2); }
  else { set_error("dw_all: H=%d unsupported", a.H); return STDADK_E_SHAPE; }
#undef GO
  STDADK_CHECK_LAUNCH("dw_all");
  return 0;
}

}  // namespace stdadk
