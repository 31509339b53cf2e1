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

// D0: the launch starts from the raw observations (TailDense0 in tail.h)
template <int MT, bool D0>
__global__ __launch_bounds__(TT) void tail_fwd_kernel(TailFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[TT / 64];
  tail_fwd_body<MT, D0>(a, smem, red, blockIdx.x);
}

template <int MT>
__global__ __launch_bounds__(TT) void tail_bwd_kernel(TailBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  tail_bwd_body<MT>(a, smem, blockIdx.x);
}

// Training step: the forward chain, the loss and the backward chain of a row tile are all row-local, so
// one workgroup runs them back to back in ONE launch (saves a launch + drain and lets the backward find
// what the forward just wrote in L2).  The forward's global stores (xhat, act, rstd, dY) are complete
// and visible to the whole workgroup after the __syncthreads() (vmcnt(0) + barrier); none of those lines
// was read by this CU earlier in the launch, so no stale copy can sit in its L1.
template <int MT, bool D0>
__global__ __launch_bounds__(TT) void tail_fwd_bwd_kernel(TailFwdArgs f, TailBwdArgs b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[TT / 64];
  tail_fwd_body<MT, D0>(f, smem, red, blockIdx.x);
  __syncthreads();
  tail_bwd_body<MT>(b, smem, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
bool tail_supported(const stdadk_mlp_desc *d, int first_layer) {
  if (first_layer < 1 || d->n_hidden < 1) return false;
  if (d->out_dim > TAIL_MAXQ) return false;
  for (int l = first_layer - 1; l < d->n_hidden; ++l)
    if (d->hidden[l] > TAIL_MAX_W || (d->hidden[l] & 15)) return false;
  return true;
}

static size_t fwd_lds(int R) { return (size_t)(2 * R * ACT_LD) * sizeof(float); }
static size_t bwd_lds(int R) {
  const bool alias = (size_t)R * ACT_LD >= (size_t)3 * NW * 256;
  return (size_t)(2 * R * ACT_LD + (alias ? 0 : 3 * NW * 256) + R * TAIL_MAXQ) * sizeof(float);
}

int tail_rows(int64_t B) {
  // two or four 16-row tiles per workgroup (shared weight fragments) once that still gives every CU
  // a workgroup; one tile per workgroup for small batches
  static const int forced = [] { const char *e = getenv("STDADK_TAIL_ROWS"); return e ? atoi(e) : 0; }();
  if (forced == 16 || forced == 32 || forced == 64) return forced;      // measurement aid
  // (MI355X, C2 widths: 64 rows +5 % step throughput at B = 16 384 and 65 536 over 32 rows)
  if (ceil_div(B, 64) >= 256) return 64;
  return ceil_div(B, 32) >= 256 ? 32 : 16;
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

template <int MT, bool D0>
static int launch_fwd(const TailFwdArgs &a, hipStream_t st) {
  constexpr int R = 16 * MT;
  static bool attr_done = false;
  if (!attr_done) {   // once per process, never inside a stream capture (the first step runs eagerly)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tail_fwd_kernel<MT, D0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds(R));
    if (e != hipSuccess) { set_error("tail_forward: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  STDADK_LAUNCH_NAMED(D0 ? "tail_fwd_kernel<dense0>" : "tail_fwd_kernel", (tail_fwd_kernel<MT, D0>),
                      dim3((unsigned)ceil_div(a.B, R)), dim3(TT), fwd_lds(R), st, a);
  STDADK_CHECK_LAUNCH("tail_forward");
  return 0;
}

int tail_forward(const TailFwdArgs &a, hipStream_t st) {
  const int r = tail_rows(a.B);
  if (int rc = check_d0(a)) return rc;
  if (a.d0.on) return r == 64 ? launch_fwd<4, true>(a, st) : (r == 32 ? launch_fwd<2, true>(a, st) : launch_fwd<1, true>(a, st));
  return r == 64 ? launch_fwd<4, false>(a, st) : (r == 32 ? launch_fwd<2, false>(a, st) : launch_fwd<1, false>(a, st));
}

template <int MT>
static int launch_bwd(const TailBwdArgs &a, hipStream_t st) {
  constexpr int R = 16 * MT;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tail_bwd_kernel<MT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds(R));
    if (e != hipSuccess) { set_error("tail_backward: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  STDADK_LAUNCH_NAMED("tail_bwd_kernel", (tail_bwd_kernel<MT>), dim3((unsigned)ceil_div(a.B, R)), dim3(TT), bwd_lds(R), st, a);
  STDADK_CHECK_LAUNCH("tail_backward");
  return 0;
}

template <int MT, bool D0>
static int launch_fwd_bwd(const TailFwdArgs &f, const TailBwdArgs &b, hipStream_t st) {
  constexpr int R = 16 * MT;
  const size_t lds = fwd_lds(R) > bwd_lds(R) ? fwd_lds(R) : bwd_lds(R);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tail_fwd_bwd_kernel<MT, D0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("tail_forward_backward: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  STDADK_LAUNCH_NAMED(D0 ? "tail_fwd_bwd_kernel<dense0>" : "tail_fwd_bwd_kernel", (tail_fwd_bwd_kernel<MT, D0>),
                      dim3((unsigned)ceil_div(f.B, R)), dim3(TT), lds, st, f, b);
  STDADK_CHECK_LAUNCH("tail_forward_backward");
  return 0;
}

int tail_forward_backward(const TailFwdArgs &f, const TailBwdArgs &b, hipStream_t st) {
  const int r = tail_rows(f.B);
  if (int rc = check_d0(f)) return rc;
  if (f.d0.on)
    return r == 64 ? launch_fwd_bwd<4, true>(f, b, st)
                   : (r == 32 ? launch_fwd_bwd<2, true>(f, b, st) : launch_fwd_bwd<1, true>(f, b, st));
  return r == 64 ? launch_fwd_bwd<4, false>(f, b, st)
                 : (r == 32 ? launch_fwd_bwd<2, false>(f, b, st) : launch_fwd_bwd<1, false>(f, b, st));
}

int tail_backward(const TailBwdArgs &a, hipStream_t st) {
  const int r = tail_rows(a.B);
  return r == 64 ? launch_bwd<4>(a, st) : (r == 32 ? launch_bwd<2>(a, st) : launch_bwd<1>(a, st));
}

}  // namespace stdadk
