// Fused "tail" of the MLP: every hidden layer after the first, the output layer and the MSE loss in
// ONE forward kernel, and the whole data path of backward (dA / dZ of every layer) in ONE kernel.
// All of it is row-local, so a workgroup carries a 16-row tile through the layers with the
// activations in LDS and the weights streamed from L2 in 32-deep K chunks onto the matrix cores.
#pragma once
#include "common.h"
#include "loss.h"

namespace stdadk {

constexpr int TAIL_MAX_LAYERS = STDADK_MAX_HIDDEN;   // hidden layers handled by one launch
constexpr int TAIL_ROWS = 64;                        // most rows a workgroup carries (tail_rows(B) picks 16, 32 or 64)
constexpr int TAIL_MIN_ROWS = 16;
constexpr int TAIL_MAX_W = 256;                      // widest layer the LDS plan holds
constexpr int TAIL_MAXQ = 8;

struct TailLayer {
  const uint16_t *Wbf;   // STDADK_FLAG_BF16: bf16 copy of W [h][hp], and of its transpose [hp][h] (the operand of
  const uint16_t *WTbf;  // dA = dZ W); NULL otherwise
  const float *W;      // [h][hp]  (nn.Linear layout)
  const float *b;      // [h]
  const float *g;      // LayerNorm gamma / beta or NULL
  const float *be;
  int h, hp;           // out / in width
  float *xhat;         // [B][h] saved normalised pre-activation (z when no LayerNorm)
  float *rstd;         // [B]
  float *act;          // [B][h] post ReLU/Dropout
  int layer_id;        // index in the model (dropout stream id)
};

// Materialising path with a small feature width (D <= TAIL_D0_MAX, first weight stored (in,out)): the launch
// starts from the raw observations -- it evaluates [X | phi | psi] of its rows into LDS (and into the
// feature buffer the backward reads), runs layer 0 as a GEMM over them and continues with the tail layers;
// replaces stdadk_rbf_build_f32 + the layer-0 GEMM + its LayerNorm/ReLU/Dropout kernel (3 launches).
constexpr int TAIL_D0_MAX = 2 * TAIL_MAX_W;
struct TailDense0 {
  int on;                           // 0: the launch starts from a_in (activations of the layer before)
                                    // 1: from the raw observations (fields below)
                                    // 2: from the two halves of layer 0's pre-activation on a site x time grid:
                                    //    z0[row] = sp[row % S] + tp[row / S] + b0 (rows time-major), L0 as for 1
  const float *sp, *tp;             // on = 2: [S][h0] spatial parts, [T][h0] temporal parts
  int S;
  const float *coords, *t, *X;      // [B][2], [B], [B][p]
  int p, Ks, Kt, basis;
  float cal;                        // calibration factor of the basis (st_interp.py:56-60)
  const float *s_centers, *s_bw;    // [Ks][2], [Ks] (bandwidths, not logs)
  const float *t_centers, *t_bw;    // [Kt]
  float *feats;                     // [B][ldf] or NULL (eval mode)
  int64_t ldf;                      // = D rounded up to 32
  const float *W0T;                 // [D][h0]
  TailLayer L0;                     // b, g, be, h = h0, hp = D, xhat, rstd, act, layer_id = 0 (W unused)
};

struct TailFwdArgs {
  int n_layers;                 // tail hidden layers in this launch (may be 0: head only)
  TailLayer L[TAIL_MAX_LAYERS];
  TailDense0 d0;
  const float *a_in;            // [B][h_in] input activations of the first tail layer
  int h_in;
  int B;
  const float *Wo, *bo;         // output layer [Q][h_last], [Q]
  int Q;
  float *y_pred;                // [B][Q]
  const float *y;               // targets [B][loss.y_cols] in the same row order, or NULL (no loss)
  LossDev loss;                 // objective evaluated on (y_pred, y)
  float grad_scale;
  float *dY;                    // [B][Q] = grad_scale * d(loss sum)/dy_pred, or NULL
  float *loss_sum;              // += the loss sum of this launch's rows, or NULL
  int layernorm;
  float eps, drop_p;
  uint64_t seed;
  const int *step_dev;
  int bf16;                     // STDADK_FLAG_BF16: the layers' GEMMs take bf16 operands (L[i].Wbf)
  int krot;                     // per-workgroup rotation of the K-chunk order (set by the launch code, tail_krot())
  unsigned long long *stamps;   // -DSTDADK_DIAG builds only: [blocks][16] wall-clock stamps (100 MHz), else NULL
};

struct TailBwdArgs {
  int n_layers;                 // hidden layers in this launch: model layers first_layer .. first_layer+n-1
  TailLayer L[TAIL_MAX_LAYERS]; // L[0] is the FIRST of them (its W is not needed: no dA below it); W of the
                                // others is read in place as the [K][N] operand of dA = dZ . W
  int B;
  const float *Wo;              // [Q][h_last]
  int Q;
  const float *dY;              // [B][Q]
  const float *act_last;        // [B][h_last] activations feeding the output layer
  float *part_head;             // [nblk][Q*h_last | Q] partials of dWo then dbo
  float *dZ[TAIL_MAX_LAYERS];   // [B][h_l] out: gradient w.r.t. the pre-LayerNorm output of layer l
  float *part[TAIL_MAX_LAYERS]; // [nblk][3][h_l] column partials (dgamma, dbeta, db) per workgroup
  int layernorm;
  float drop_p;
  uint64_t seed;
  const int *step_dev;
  int bf16;                     // STDADK_FLAG_BF16: dA = dZ W with bf16 operands (L[i].WTbf)
  int krot;                     // as in TailFwdArgs
  int *zero_ints = nullptr;     // n_zero ints the workgroup of tile 0 clears (arrival counters of the weight-gradient
  int n_zero = 0;               // launch behind this one, FinArgs), or NULL
  unsigned long long *stamps;   // -DSTDADK_DIAG builds only (see TailFwdArgs), else NULL
};

bool tail_supported(const stdadk_mlp_desc *d, int first_layer);
// 1 with STDADK_KROT=1 (measurement aid, off by default): workgroups of a launch walk the K chunks of the shared
// weights from different starting chunks (tail_body.h: chunk_start).  Measured in round 3 at no gain (the GEMM phases
// of the 16-row tiles 6.92 vs 7.12 us, the fused kernel 57.56 vs 57.64 us in-kernel: DESIGN.md section 8) -- the CUs
// of an XCD asking for the same lines at the same time is not what paces those phases.
int tail_krot();
// rows per workgroup the launches will use for a batch of B rows (`cap32`: bf16 operands together with the dense
// layer 0 inside the launch keep three activation images in LDS, which fit for at most 32 rows)
int tail_rows(int64_t B, bool cap32 = false);
int tail_forward(const TailFwdArgs &a, hipStream_t st);
int tail_backward(const TailBwdArgs &a, hipStream_t st, bool cap32 = false);
// training: forward (with the loss) and backward of every row tile in one launch
int tail_forward_backward(const TailFwdArgs &f, const TailBwdArgs &b, hipStream_t st);

}  // namespace stdadk
#include "window.h"
namespace stdadk {
// training, B <= 4096 on the window path: layer 0 + the tail forward + loss + tail backward in one launch
// (fused_step.hip); rows_per_wg / n_wg of `l` are set by the callee
bool l1_tail_supported(int64_t B, int H);
int l1_tail_launch(const L1FwdArgs &l, int basis, bool layernorm, const TailFwdArgs &f, const TailBwdArgs &b,
                   hipStream_t st);

}  // namespace stdadk
