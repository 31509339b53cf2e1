// Every weight gradient of a training step that is a reduction over the batch, in ONE launch: the grouped
// split-K products dW_l = dZ_l^T a_(l-1) of the layers after the first (+ the temporal / covariate rows of
// dW0^T) on the matrix cores, and the per-knot gather of the spatial rows of dW0^T.  The two kinds of
// workgroup are independent (both only need the dZ of the backward chain), have the same shape (256
// threads) and complementary bottlenecks (staged MFMA tiles vs L2 row gathers), so sharing the CUs beats
// running them one after the other, and one launch + drain is saved.
#include "gemm_body.h"
#include "l1_bwd_body.h"

namespace stdadk {

static_assert(GT == BW_T, "the GEMM tiles and the knot groups must share the workgroup shape");

// NK (fixed knots only): 2 neighbouring knots per wave, see l1_window_bwd_multi_body; 1: one knot per wave
template <int CPL, int BASIS, bool KNOTS, int NK>
__global__ __launch_bounds__(GT) void dw_all_kernel(GemmGroup grp, int n_gemm_blocks, L1BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[GROUP_LDS_FLOATS];
  // GEMM tiles take the low block ids (dispatched first): measured 33.5 us vs 42.5 us the other way round
  if ((int)blockIdx.x < n_gemm_blocks) { gemm_tn_grouped_block(grp, (int)blockIdx.x, lds); return; }
  // XCD-striped knot groups start at a multiple of 8, so that (group block & 7) is the XCD of the workgroup
  const int first = a.xcd_slots > 0 ? (n_gemm_blocks + 7) & ~7 : n_gemm_blocks;
  if ((int)blockIdx.x < first) return;
  if constexpr (NK > 1) l1_window_bwd_multi_body<CPL, BASIS, NK>(a, (int)blockIdx.x - first);
  else l1_window_bwd_body<CPL, BASIS, KNOTS>(a, (int)blockIdx.x - first);
}

int launch_dw_all(GemmGroup &grp, const L1BwdArgs &a_in, int basis, hipStream_t st) {
  L1BwdArgs a = a_in;
  STDADK_REQUIRE(a.G <= 256, STDADK_E_ARG, "dw_all: G too large");
  STDADK_REQUIRE((int64_t)a.B * a.H < (1ll << 32), STDADK_E_ARG, "dw_all: B*H exceeds 32-bit offsets");
  STDADK_REQUIRE(!a.kpart || a.W0T, STDADK_E_ARG, "dw_all: knot sums need W0^T");
  int ng = 0;
  int rc = gemm_tn_grouped_prepare(grp, &ng);
  if (rc) return rc;
  const int nk = knots_per_wave(a);
  a.xcd_slots = knot_xcd_slots(a.g, a.kpart ? 1 : nk);
  const unsigned grid = a.xcd_slots > 0 ? (unsigned)((ng + 7) & ~7) + 8u * (unsigned)a.xcd_slots
                                        : (unsigned)ng + (unsigned)ceil_div(knot_group_count(a.g, nk), BW_T / 64);
#define GO(CPL_, BS_)                                                                                      \
  do {                                                                                                     \
    if (a.kpart) STDADK_LAUNCH_NAMED("dw_all_kernel<knots>", (dw_all_kernel<CPL_, BS_, true, 1>),          \
                                     dim3(grid), dim3(GT), 0, st, grp, ng, a);                             \
    else if (nk == 2) STDADK_LAUNCH_NAMED("dw_all_kernel", (dw_all_kernel<CPL_, BS_, false, 2>), dim3(grid), \
                                          dim3(GT), 0, st, grp, ng, a);                                    \
    else STDADK_LAUNCH_NAMED("dw_all_kernel", (dw_all_kernel<CPL_, BS_, false, 1>), dim3(grid),            \
                             dim3(GT), 0, st, grp, ng, a);                                                 \
  } while (0)
  if (a.H == 256) { if (basis == STDADK_BASIS_WENDLAND) GO(4, 0); else GO(4, 2); }
  else if (a.H == 128) { if (basis == STDADK_BASIS_WENDLAND) GO(2, 0); else GO(2, 2); }
  else { set_error("dw_all: H=%d unsupported", a.H); return STDADK_E_SHAPE; }
#undef GO
  STDADK_CHECK_LAUNCH("dw_all");
  return 0;
}

}  // namespace stdadk
