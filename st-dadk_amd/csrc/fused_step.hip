// One launch for everything of a training step that is row-local: layer 0 on the window path (feature
// evaluation, W0^T row gather, LayerNorm/ReLU/Dropout), the remaining layers forward, the loss, and the
// activation-gradient chain backward — one workgroup carries its 16 cell-sorted observations through all
// of it.  Used for batches of at most 16 rows per CU (B <= 4096), where each separate launch costs more
// in launch + drain (~4.5 us) than it gains in balance.  The phases hand data over through global memory
// exactly as the separate kernels do; a __syncthreads() (vmcnt(0) + barrier) between them makes a phase's
// stores visible to the whole workgroup, and none of the lines was read by this CU earlier in the launch.
#include "l1_body.h"
#include "tail_body.h"

namespace stdadk {

using T16 = Tail<16>;
constexpr int TT = T16::TT;
static_assert(FW_T == TT, "the layer-0 body and the tail bodies must share the workgroup shape");

template <int CPL, bool LN, int BASIS, bool FREE, bool BF>
__global__ __launch_bounds__(TT) void l1_tail_kernel(L1FwdArgs l, TailFwdArgs f, TailBwdArgs b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[TT / 64];
  const int tile = l1_chunk_of(blockIdx.x, l.n_wg);          // XCD-aware tile order (l.rows_per_wg == 16)
  const int r0 = 16 * tile;
  if (r0 >= l.B) return;                                      // padding workgroup (grid is a multiple of 8)
#ifdef STDADK_DIAG   // diagnostic build: phase boundaries into slots 10..13 of the forward stamp buffer
#define PSTAMP(i) do { if (f.stamps && threadIdx.x == 0) f.stamps[tile * 16 + (i)] = wall_clock64(); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
  PSTAMP(10);
  l1_window_fwd_body<CPL, LN, BASIS, FREE>(l, smem, r0, min(r0 + 16, l.B));
  PSTAMP(11);
  __syncthreads();
  T16::tail_fwd_body<1, false, BF>(f, smem, red, tile);
  PSTAMP(12);
  __syncthreads();
  T16::tail_bwd_body<1, BF>(b, smem, tile);
  PSTAMP(13);
}

template <int CPL, bool LN, int BASIS, bool FREE, bool BF>
static int launch(const L1FwdArgs &l, const TailFwdArgs &f, const TailBwdArgs &b, hipStream_t st) {
  const int Kt_pad = (l.g.Kt + 3) & ~3;
  const size_t lds_l1 = ((size_t)l.g.Kt * 64 * CPL + (FW_T / 64) * (LIST * 2 + Kt_pad)) * sizeof(float);
  const size_t lds_fwd = BF ? (size_t)(16 * ACT_LD) * sizeof(float) + (size_t)16 * ABF_LD * sizeof(u16)
                            : (size_t)(2 * 16 * ACT_LD) * sizeof(float);
  const size_t lds_bwd = T16::tail_bwd_lds_floats<1, BF>() * sizeof(float);
  size_t lds = lds_l1 > lds_bwd ? lds_l1 : lds_bwd;
  if (lds_fwd > lds) lds = lds_fwd;
  auto kern = l1_tail_kernel<CPL, LN, BASIS, FREE, BF>;
  static size_t attr_lds = 0;     // raised on the first (eager) call of a configuration, never under capture
  if (lds > attr_lds) {
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void *>(kern), (int)lds);
    if (e != hipSuccess) { set_error("l1_tail: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_lds = lds;
  }
  STDADK_LAUNCH_NAMED(BF ? "l1_tail_kernel<bf16>" : "l1_tail_kernel", kern, dim3((unsigned)l.n_wg), dim3(TT), lds, st, l, f, b);
  STDADK_CHECK_LAUNCH("l1_tail");
  return 0;
}

bool l1_tail_supported(int64_t B, int H) { return B <= 16 * 256 && (H == 256 || H == 128); }

int l1_tail_launch(const L1FwdArgs &l_in, int basis, bool ln, const TailFwdArgs &f_in, const TailBwdArgs &b_in,
                   hipStream_t st) {
  L1FwdArgs l = l_in;
  TailFwdArgs f = f_in;
  TailBwdArgs b = b_in;
  f.krot = b.krot = tail_krot();
  STDADK_REQUIRE(l1_tail_supported(l.B, l.H) && f.B == l.B && b.B == l.B, STDADK_E_ARG,
                 "l1_tail: needs B <= 4096 and H in {128, 256}");
  STDADK_REQUIRE((int64_t)(l.g.p + l.g.Ks + l.g.Kt) * l.H < (1ll << 32), STDADK_E_ARG,
                 "l1_tail: D*H exceeds 32-bit offsets");
  l.rows_per_wg = 16;
  l.n_wg = (int)((ceil_div(l.B, 16) + 7) / 8 * 8);            // whole groups of 8 for the XCD mapping
  STDADK_REQUIRE((f.bf16 != 0) == (b.bf16 != 0), STDADK_E_ARG, "l1_tail: forward and backward disagree on bf16 operands");
  for (int i = 0; i < f.n_layers; ++i)
    STDADK_REQUIRE(!f.bf16 || (f.L[i].Wbf && (reinterpret_cast<uintptr_t>(f.L[i].Wbf) & 15) == 0), STDADK_E_ARG,
                   "l1_tail: STDADK_FLAG_BF16 needs params->W_bf16 of every layer after the first");
  for (int i = 1; i < b.n_layers; ++i)
    STDADK_REQUIRE(!b.bf16 || (b.L[i].WTbf && (reinterpret_cast<uintptr_t>(b.L[i].WTbf) & 15) == 0), STDADK_E_ARG,
                   "l1_tail: STDADK_FLAG_BF16 needs params->WT_bf16 of every layer after the first");
#define GO2(CPL_, BF_)                                                                                        \
  (basis == STDADK_BASIS_WENDLAND                                                                             \
       ? (ln ? ((l.halo || l.kperm) ? launch<CPL_, true, 0, true, BF_>(l, f, b, st) : launch<CPL_, true, 0, false, BF_>(l, f, b, st)) \
             : ((l.halo || l.kperm) ? launch<CPL_, false, 0, true, BF_>(l, f, b, st) : launch<CPL_, false, 0, false, BF_>(l, f, b, st))) \
       : (ln ? ((l.halo || l.kperm) ? launch<CPL_, true, 2, true, BF_>(l, f, b, st) : launch<CPL_, true, 2, false, BF_>(l, f, b, st)) \
             : ((l.halo || l.kperm) ? launch<CPL_, false, 2, true, BF_>(l, f, b, st) : launch<CPL_, false, 2, false, BF_>(l, f, b, st))))
#define GO(CPL_) (f.bf16 ? GO2(CPL_, true) : GO2(CPL_, false))
  if (l.H == 256) return GO(4);
  return GO(2);
#undef GO
#undef GO2
}

}  // namespace stdadk
