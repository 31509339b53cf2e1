// Fused MLP tail (see tail.h).  fp32 on the matrix cores with v_mfma_f32_16x16x4_f32: lane l holds
// A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15]; D[row = 4*(l>>4) + reg][col = l&15].
// A lane group q = l>>4 reads 4 consecutive k with one ds_read_b128 and feeds them to 4 MFMAs
// (k = 16j + 4q + e for MFMA e), for both operands, so the k order inside a 16-deep group is free.
//
// Replaces, for the layers after the first: nn.Linear / nn.LayerNorm / nn.ReLU / nn.Dropout forward
// (stnf/models/st_interp.py:656-690,880), nn.MSELoss (scripts/train_st_interp.py:549,621) and the
// activation-gradient half of loss.backward() (:693).
#include "tail.h"
#include "tail_body.h"

namespace stdadk {

// NW: waves per workgroup (tail_body.h): 16 = one 1024-thread workgroup per CU, the only shape the library
// instantiates; 8 = 512 threads, at most 128 registers and (32-row tiles) 76 KiB of LDS, so that two workgroups share
// a CU (see TAIL_DISPATCH below).
// D0: the launch starts from the raw observations (TailDense0 in tail.h); BF: bf16 operands (STDADK_FLAG_BF16)
#define TAIL_BOUNDS(NW) __launch_bounds__(64 * NW, NW == 8 ? 4 : 1)

template <int NW, int MT, bool D0, bool BF>
__global__ TAIL_BOUNDS(NW) void tail_fwd_kernel(TailFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[NW];
  Tail<NW>::template tail_fwd_body<MT, D0, BF>(a, smem, red, blockIdx.x);
}

template <int NW, int MT, bool BF>
__global__ TAIL_BOUNDS(NW) void tail_bwd_kernel(TailBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Tail<NW>::template tail_bwd_body<MT, BF>(a, smem, blockIdx.x);
}

// Training step: the forward chain, the loss and the backward chain of a row tile are all row-local, so
// one workgroup runs them back to back in ONE launch (saves a launch + drain and lets the backward find
// what the forward just wrote in L2).  The forward's global stores (xhat, act, rstd, dY) are complete
// and visible to the whole workgroup after the __syncthreads() (vmcnt(0) + barrier); none of those lines
// was read by this CU earlier in the launch, so no stale copy can sit in its L1.
template <int NW, int MT, bool D0, bool BF>
__global__ TAIL_BOUNDS(NW) void tail_fwd_bwd_kernel(TailFwdArgs f, TailBwdArgs b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[NW];
  Tail<NW>::template tail_fwd_body<MT, D0, BF>(f, smem, red, blockIdx.x);
  __syncthreads();
  Tail<NW>::template tail_bwd_body<MT, BF>(b, smem, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
bool tail_supported(const stdadk_mlp_desc *d, int first_layer) {
  if (first_layer < 1 || d->n_hidden < 1) return false;
  if (d->out_dim > TAIL_MAXQ) return false;
  for (int l = first_layer - 1; l < d->n_hidden; ++l)
    if (d->hidden[l] > TAIL_MAX_W || (d->hidden[l] & 15)) return false;
  return true;
}

// bf16 operands: one fp32 tile + the bf16 image (with the dense layer 0: the fp32 pair + the image)
static size_t fwd_lds(int R, bool d0, bool bf) {
  if (!bf) return (size_t)(2 * R * ACT_LD) * sizeof(float);
  return (size_t)((d0 ? 2 : 1) * R * ACT_LD) * sizeof(float) + (size_t)R * ABF_LD * sizeof(u16);
}
template <int NW>
static size_t bwd_lds(int R, bool bf) {
  using T = Tail<NW>;
  if (bf) return (R == 64 ? T::template tail_bwd_lds_floats<4, true>() : (R == 32 ? T::template tail_bwd_lds_floats<2, true>() : T::template tail_bwd_lds_floats<1, true>())) * sizeof(float);
  return (R == 64 ? T::template tail_bwd_lds_floats<4, false>() : (R == 32 ? T::template tail_bwd_lds_floats<2, false>() : T::template tail_bwd_lds_floats<1, false>())) * sizeof(float);
}
static_assert(Tail<16>::tail_bwd_lds_floats<4, true>() * sizeof(float) <= 160 * 1024, "bf16 backward tile does not fit the LDS");
int tail_rows(int64_t B, bool cap32) {
  // two or four 16-row tiles per workgroup (shared weight fragments) once that still gives every CU
  // a workgroup; one tile per workgroup for small batches
  static const int forced = [] { const char *e = getenv("STDADK_TAIL_ROWS"); return e ? atoi(e) : 0; }();
  if (forced == 16 || forced == 32 || (forced == 64 && !cap32)) return forced;      // measurement aid
  // (MI355X, C2 widths: 64 rows +5 % step throughput at B = 16 384 and 65 536 over 32 rows)
  if (ceil_div(B, 64) >= 256 && !cap32) return 64;
  return ceil_div(B, 32) >= 256 ? 32 : 16;
}

static int check_bf(int bf16, int n, const TailLayer *L, bool fwd, int first) {
  if (!bf16) return 0;
  for (int i = first; i < n; ++i)
    STDADK_REQUIRE((fwd ? L[i].Wbf : L[i].WTbf) != nullptr && (reinterpret_cast<uintptr_t>(fwd ? L[i].Wbf : L[i].WTbf) & 15) == 0,
                   STDADK_E_ARG, "tail: STDADK_FLAG_BF16 needs the 16-byte aligned bf16 weight copies of every layer after "
                   "the first (params->W_bf16 / WT_bf16; stdadk_bf16_shadow_refresh)");
  return 0;
}

static int check_d0(const TailFwdArgs &a) {
  if (!a.d0.on) return 0;
  const TailDense0 &z = a.d0;
  if (z.on == 2) {
    STDADK_REQUIRE(z.sp && z.tp && z.S > 0 && z.L0.b && z.L0.h <= TAIL_MAX_W && (z.L0.h & 15) == 0 && aligned16(z.sp) &&
                       aligned16(z.tp) && aligned16(z.L0.b),
                   STDADK_E_ARG, "tail: site x time parts need sp, tp (16-byte aligned), S > 0 and h0 <= %d", TAIL_MAX_W);
    return 0;
  }
  const int D = z.p + z.Ks + z.Kt;
  STDADK_REQUIRE(D >= 1 && D <= TAIL_D0_MAX && z.L0.hp == D && z.ldf == ((D + 31) & ~31) && z.W0T && z.coords && z.t &&
                     (z.p == 0 || z.X) && z.L0.h <= TAIL_MAX_W && (z.L0.h & 15) == 0,
                 STDADK_E_ARG, "tail: dense layer 0 needs D <= %d, ldf = D rounded up to 32 and h0 <= %d", TAIL_D0_MAX,
                 TAIL_MAX_W);
  return 0;
}

template <int NW, int MT, bool D0, bool BF>
static int launch_fwd(const TailFwdArgs &a, hipStream_t st) {
  constexpr int R = 16 * MT, TT = 64 * NW;
  static bool attr_done = false;
  if (!attr_done) {   // once per process, never inside a stream capture (the first step runs eagerly)
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void *>(tail_fwd_kernel<NW, MT, D0, BF>), (int)fwd_lds(R, D0, BF));
    if (e != hipSuccess) { set_error("tail_forward: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  STDADK_LAUNCH_NAMED(D0 ? (BF ? "tail_fwd_kernel<dense0,bf16>" : "tail_fwd_kernel<dense0>") : (BF ? "tail_fwd_kernel<bf16>" : "tail_fwd_kernel"),
                      (tail_fwd_kernel<NW, MT, D0, BF>), dim3((unsigned)ceil_div(a.B, R)), dim3(TT), fwd_lds(R, D0, BF), st, a);
  STDADK_CHECK_LAUNCH("tail_forward");
  return 0;
}

// dispatch on (rows, dense layer 0, bf16 operands); the bf16 + dense-0 combination is built for <= 32 rows.
// All launches use Tail<16> (1024 threads, one workgroup per CU): Tail<8> -- 512 threads, 32-row tiles, TWO
// workgroups per CU (confirmed by tools/diag/occupancy_tail8.hip), meant to overlap one tile's row-local phases with
// the other's GEMM phases -- measured 0-4 % SLOWER at 16 384 and 65 536 rows, fp32 and bf16 (round 2), and again in
// round 3 with the second workgroup of a CU started 3 .. 40 us late so that the two cannot run in lockstep: 451 us
// (no delay) / 452 / 457 / 457 / 461 / 470 us against 448 us for one 1024-thread workgroup at 65 536 rows
// (profiles/r03_tail_stagger_negative.txt).  The fp32 matrix instructions run at the fp32 VECTOR rate: a GEMM phase
// and a row-local phase compete for the same issue, so there is nothing to overlap -- what pays is fewer vector
// instructions in the row-local phases.  Tail<8> is not instantiated in the library.
#define TAIL_DISPATCH(FN, r, d0, bf, ...)                                                                    \
  ((bf) ? ((d0) ? ((r) == 32 ? FN<16, 2, true, true>(__VA_ARGS__) : FN<16, 1, true, true>(__VA_ARGS__))      \
                : ((r) == 64 ? FN<16, 4, false, true>(__VA_ARGS__)                                           \
                             : ((r) == 32 ? FN<16, 2, false, true>(__VA_ARGS__) : FN<16, 1, false, true>(__VA_ARGS__)))) \
        : ((d0) ? ((r) == 64 ? FN<16, 4, true, false>(__VA_ARGS__)                                           \
                             : ((r) == 32 ? FN<16, 2, true, false>(__VA_ARGS__) : FN<16, 1, true, false>(__VA_ARGS__))) \
                : ((r) == 64 ? FN<16, 4, false, false>(__VA_ARGS__)                                          \
                             : ((r) == 32 ? FN<16, 2, false, false>(__VA_ARGS__) : FN<16, 1, false, false>(__VA_ARGS__)))))

int tail_krot() {
  static const int on = [] { const char *e = getenv("STDADK_KROT"); return (e && e[0] == '1') ? 1 : 0; }();
  return on;
}

int tail_forward(const TailFwdArgs &a_in, hipStream_t st) {
  TailFwdArgs a = a_in;
  a.krot = tail_krot();
  const bool d0 = a.d0.on != 0, bf = a.bf16 != 0;
  const int r = tail_rows(a.B, bf && d0);
  if (int rc = check_d0(a)) return rc;
  if (int rc = check_bf(a.bf16, a.n_layers, a.L, true, 0)) return rc;
  return TAIL_DISPATCH(launch_fwd, r, d0, bf, a, st);
}

template <int NW, int MT, bool BF>
static int launch_bwd(const TailBwdArgs &a, hipStream_t st) {
  constexpr int R = 16 * MT, TT = 64 * NW;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void *>(tail_bwd_kernel<NW, MT, BF>), (int)bwd_lds<NW>(R, BF));
    if (e != hipSuccess) { set_error("tail_backward: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  STDADK_LAUNCH_NAMED(BF ? "tail_bwd_kernel<bf16>" : "tail_bwd_kernel", (tail_bwd_kernel<NW, MT, BF>),
                      dim3((unsigned)ceil_div(a.B, R)), dim3(TT), bwd_lds<NW>(R, BF), st, a);
  STDADK_CHECK_LAUNCH("tail_backward");
  return 0;
}

template <int NW, int MT, bool D0, bool BF>
static int launch_fwd_bwd(const TailFwdArgs &f, const TailBwdArgs &b, hipStream_t st) {
  constexpr int R = 16 * MT, TT = 64 * NW;
  const size_t lds = fwd_lds(R, D0, BF) > bwd_lds<NW>(R, BF) ? fwd_lds(R, D0, BF) : bwd_lds<NW>(R, BF);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void *>(tail_fwd_bwd_kernel<NW, MT, D0, BF>), (int)lds);
    if (e != hipSuccess) { set_error("tail_forward_backward: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  STDADK_LAUNCH_NAMED(D0 ? (BF ? "tail_fwd_bwd_kernel<dense0,bf16>" : "tail_fwd_bwd_kernel<dense0>")
                         : (BF ? "tail_fwd_bwd_kernel<bf16>" : "tail_fwd_bwd_kernel"),
                      (tail_fwd_bwd_kernel<NW, MT, D0, BF>), dim3((unsigned)ceil_div(f.B, R)), dim3(TT), lds, st, f, b);
  STDADK_CHECK_LAUNCH("tail_forward_backward");
  return 0;
}

int tail_forward_backward(const TailFwdArgs &f_in, const TailBwdArgs &b_in, hipStream_t st) {
  TailFwdArgs f = f_in;
  TailBwdArgs b = b_in;
  f.krot = b.krot = tail_krot();
  const bool d0 = f.d0.on != 0, bf = f.bf16 != 0;
  STDADK_REQUIRE((f.bf16 != 0) == (b.bf16 != 0), STDADK_E_ARG, "tail: forward and backward disagree on bf16 operands");
  const int r = tail_rows(f.B, bf && d0);
  if (int rc = check_d0(f)) return rc;
  if (int rc = check_bf(f.bf16, f.n_layers, f.L, true, 0)) return rc;
  if (int rc = check_bf(b.bf16, b.n_layers, b.L, false, 1)) return rc;
  return TAIL_DISPATCH(launch_fwd_bwd, r, d0, bf, f, b, st);
}

// `cap32`: the forward of this batch ran with the dense layer 0 and bf16 operands (32-row tiles at most); the
// partial buffers of both kernels are indexed by the same tile size
int tail_backward(const TailBwdArgs &a_in, hipStream_t st, bool cap32) {
  TailBwdArgs a = a_in;
  a.krot = tail_krot();
  const int r = tail_rows(a.B, cap32);
  if (int rc = check_bf(a.bf16, a.n_layers, a.L, false, 1)) return rc;
  if (a.bf16) return r == 64 ? launch_bwd<16, 4, true>(a, st) : (r == 32 ? launch_bwd<16, 2, true>(a, st) : launch_bwd<16, 1, true>(a, st));
  return r == 64 ? launch_bwd<16, 4, false>(a, st) : (r == 32 ? launch_bwd<16, 2, false>(a, st) : launch_bwd<16, 1, false>(a, st));
}

}  // namespace stdadk
