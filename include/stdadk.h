/*
 * stdadk.h — C ABI of libstdadk.so, the MI355X (gfx950) implementation of ST-DADK's
 * spatio-temporal interpolation hot path.
 *
 * The reference (STLABTW/ST-DADK) is pure Python on PyTorch and has no FFI layer; the boundary its
 * driver sees is the nn.Module surface of stnf/models/st_interp.py.  Each entry point below names
 * the reference function (file:line, relative to the reference root) whose arithmetic it replaces.
 * The Python face (st-dadk_amd/stnf) binds these symbols with ctypes; INTEGRATION.md shows the
 * stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer (hipMalloc / torch
 *     allocator) unless the parameter says "host".  Inputs are borrowed, outputs are written into
 *     caller-provided buffers; the library allocates nothing and keeps no global mutable state.
 *   - `stream` is a hipStream_t (0 = default stream).  Every call only enqueues work on it:
 *     no host<->device sync, no host read of device scalars => hipGraph-capturable.
 *   - all matrices are row-major fp32; `ld*` is the row stride in elements.
 *   - return value: 0 on success; <0 argument error (STDADK_E_*); >0 a hipError_t.
 *     stdadk_last_error() returns a thread-local message for the last non-zero return.
 */
#ifndef STDADK_H
#define STDADK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STDADK_ABI_VERSION 8
#define STDADK_MAX_HIDDEN 8
#define STDADK_MAX_LEVELS 8
#define STDADK_SUMSQ_PARTS 256 /* partial sums written by stdadk_sumsq_f32 */

#define STDADK_E_ARG (-1)     /* null pointer / negative size / bad enum       */
#define STDADK_E_SHAPE (-2)   /* shapes inconsistent with the descriptor      */
#define STDADK_E_ALIGN (-3)   /* buffer not aligned as documented             */
#define STDADK_E_WORKSPACE (-4) /* workspace smaller than *_workspace_bytes() */

typedef void *stdadk_stream_t; /* hipStream_t */

/* SpatialBasisEmbedding.basis_function, stnf/models/st_interp.py:56-60,451-458 */
enum { STDADK_BASIS_WENDLAND = 0, STDADK_BASIS_GAUSSIAN = 1, STDADK_BASIS_TRIANGULAR = 2 };

int stdadk_abi_version(void);
const char *stdadk_last_error(void);

/* ------------------------------------------------------------------------------------------
 * A2-A5  feature builder (materialising):  out[b, :] = [ X[b, 0:p] | phi(coords[b]) | psi(t[b]) ]
 *
 * Replaces SpatialBasisEmbedding.forward + _wendland/_gaussian/_triangular
 * (stnf/models/st_interp.py:433-491), TemporalBasisEmbedding.forward (:583-596) and the
 * torch.cat of STInterpMLP.forward (:843-846).
 *   phi[b,k] = f( sqrt((x-cx_k)^2+(y-cy_k)^2) / (bw_k * cal) ),  cal = 1 / 0.223477 / 0.654714
 *   psi[b,j] = exp(-0.5 * ((t-c_j)/bw_j)^2)
 * coords [B,2], t [B] (the reference's (B,1) column), X [B,p] or NULL when p == 0,
 * s_centers [Ks,2], s_bw [Ks], t_centers [Kt], t_bw [Kt]; Ks or Kt may be 0.
 * out [B, ld_out], ld_out >= p+Ks+Kt; columns beyond p+Ks+Kt (row padding) are zero-filled.
 * ------------------------------------------------------------------------------------------ */
int stdadk_rbf_build_f32(const float *coords, const float *t, const float *X, int64_t B, int32_t p,
                         const float *s_centers, const float *s_bw, int64_t Ks, int32_t basis,
                         const float *t_centers, const float *t_bw, int64_t Kt, float *out,
                         int64_t ld_out, stdadk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A6  MLP  (Linear -> LayerNorm -> ReLU -> Dropout) x L -> Linear      st_interp.py:656-690,880
 * ------------------------------------------------------------------------------------------ */
typedef struct stdadk_mlp_desc {
  int32_t n_hidden;                  /* L, 0..STDADK_MAX_HIDDEN                               */
  int32_t in_dim;                    /* D = p + Ks + Kt                                       */
  int32_t hidden[STDADK_MAX_HIDDEN]; /* hidden_dims                                           */
  int32_t out_dim;                   /* Q (1 for MSE regression)                              */
  int32_t layernorm;                 /* 0/1; eps below (nn.LayerNorm default 1e-5)            */
  float ln_eps;
  float dropout_p;                   /* 0 => no dropout layers                                */
} stdadk_mlp_desc;

/* Device pointers of the parameters / their gradients; layer l = 0..L-1 are the hidden Linear
 * (+LayerNorm) layers, layer L is the output Linear.  W[l] is (out,in) row-major contiguous as
 * nn.Linear stores it; ln_g/ln_b are NULL when layernorm == 0.
 * W_bf16 / WT_bf16 (parameters only, read with STDADK_FLAG_BF16; NULL otherwise): bfloat16 copies of the
 * hidden layers l >= 1 -- W_bf16[l] is W[l] rounded to nearest-even, (out,in) row-major; WT_bf16[l] its
 * transpose (in,out) row-major (the K-contiguous operand of dA = dZ W).  stdadk_bf16_shadow_refresh makes
 * them, the optimiser entry points keep them current (stdadk_bf16_shadow). */
typedef struct stdadk_mlp_tensors {
  float *W[STDADK_MAX_HIDDEN + 1];
  float *b[STDADK_MAX_HIDDEN + 1];
  float *ln_g[STDADK_MAX_HIDDEN];
  float *ln_b[STDADK_MAX_HIDDEN];
  uint16_t *W_bf16[STDADK_MAX_HIDDEN + 1];
  uint16_t *WT_bf16[STDADK_MAX_HIDDEN + 1];
} stdadk_mlp_tensors;

/* Bytes of workspace stdadk_mlp_forward_f32/backward_f32 need for B rows (saved activations +
 * split-K slabs).  The same workspace must be passed to backward untouched. */
size_t stdadk_mlp_workspace_bytes(const stdadk_mlp_desc *desc, int64_t B);

/* y_pred[B,Q] = mlp(features[B, ldf]).  `training` != 0 applies dropout (keep-mask from the
 * counter-based generator keyed by drop_seed, or from drop_mask[l] (uint8 [B,h_l], 1 = keep) when
 * that host array of device pointers is non-NULL) and saves what backward needs in `workspace`. */
int stdadk_mlp_forward_f32(const stdadk_mlp_desc *desc, const stdadk_mlp_tensors *params,
                           const float *features, int64_t ldf, int64_t B, float *y_pred,
                           void *workspace, size_t workspace_bytes, int32_t training,
                           uint64_t drop_seed, const uint8_t *const *drop_mask,
                           stdadk_stream_t stream);

/* A8  loss.backward() through A6 (scripts/train_st_interp.py:693): given dY[B,Q] = dLoss/dy_pred,
 * writes every parameter gradient into `grads` (overwrite, not accumulate).  Knots are buffers in
 * the fixed-basis model, so no gradient flows into the features and dX of layer 0 is skipped. */
int stdadk_mlp_backward_f32(const stdadk_mlp_desc *desc, const stdadk_mlp_tensors *params,
                            const stdadk_mlp_tensors *grads, const float *features, int64_t ldf,
                            int64_t B, const float *dY, void *workspace, size_t workspace_bytes,
                            uint64_t drop_seed, const uint8_t *const *drop_mask,
                            stdadk_stream_t stream);

/* A7  nn.MSELoss() (scripts/train_st_interp.py:549,621) fused with its gradient:
 *   loss_sum[0] += sum((y_pred-y)^2)   (caller divides by the global element count)
 *   dY[i] = 2*(y_pred[i]-y[i])*grad_scale          (grad_scale = 1/(B*Q) for the plain mean)
 * n = B*Q elements.  dY may be NULL (evaluation); loss_sum may be NULL. */
int stdadk_mse_f32(const float *y_pred, const float *y, int64_t n, float grad_scale, float *dY,
                   float *loss_sum, stdadk_stream_t stream);

/* N3  quantile / multi-quantile objectives (scripts/train_st_interp.py:37-88,622-658) fused with
 * their gradient, as a generalisation of stdadk_mse_f32.  For y_pred [B,Q] and y [B,y_cols]:
 *   STDADK_LOSS_MSE     : sum (y_pred - y)^2
 *   STDADK_LOSS_PINBALL : sum_q sum_b max((tau_q-1) e, tau_q e), e = y - y_pred   (quantile_loss, :37-50;
 *                         the multi-quantile mean over q of per-quantile means, :625-632, is this sum
 *                         divided by B*Q), plus the prediction-level non-crossing penalty
 *                         nc_weight * mean_b sum_k relu(y_pred[b,k] - y_pred[b,k+1])^nc_power
 *                         (non_crossing_penalty, :53-88, reduction "mean"; added as nc_weight*Q*sum so
 *                         that loss_sum / (B*Q) is the reference's batch loss).
 *   loss_sum[0] += that sum;   dY = grad_scale * d(sum)/dy_pred     (grad_scale = 1/(B*Q) for the mean)
 * Sub-gradients follow torch: max() splits ties 1/2:1/2, relu'(0) = 0.  y_cols == 1 broadcasts one
 * target over the Q outputs (the (B,1) targets of multi-quantile training). */
#define STDADK_LOSS_MSE 0
#define STDADK_LOSS_PINBALL 1
#define STDADK_MAX_Q 8
typedef struct stdadk_loss_desc {
  int32_t kind;             /* STDADK_LOSS_*                                                    */
  int32_t y_cols;           /* columns of y: Q, or 1 (broadcast)                                */
  float tau[STDADK_MAX_Q];  /* quantile levels of the Q outputs (PINBALL)                       */
  float nc_weight;          /* prediction-level non-crossing weight, 0 = off (PINBALL, Q > 1)   */
  int32_t nc_power;         /* 1 (hinge) or 2 (squared hinge)                                   */
} stdadk_loss_desc;
int stdadk_loss_f32(const stdadk_loss_desc *loss, const float *y_pred, const float *y, int64_t B,
                    int32_t Q, float grad_scale, float *dY, float *loss_sum, stdadk_stream_t stream);

/* N3  delta-reparameterised output head (stnf/models/st_interp.py:671-686,849-877): the Q output
 * rows are cumulative sums of per-quantile vectors delta_k = (delta_k0 | delta_k1..d),
 *   beta_k = sum_{l<=k} delta_l,   Wo[k,:] = beta_k[1:],  bo[k] = beta_k[0],
 * so the step-level entry points run unchanged on (Wo, bo).  delta is [Q, ldd] (ldd >= d+1).
 * stdadk_delta_head_f32 builds Wo [Q,d] and bo [Q]; stdadk_delta_head_backward_f32 maps (dWo, dbo)
 * back, d_delta_l = sum_{k>=l} d beta_k (overwrite), and adds the parameter-level penalty of
 * compute_p_nc_delta_penalty (scripts/train_st_interp.py:91-160, used at :640-649):
 *   P = sum_{k>=2} (delta_k0 - max(delta_k0, S_k)),  S_k = sum_j max(0, -delta_kj)
 *   d_delta += lambda_grad * dP/d delta;   loss_sum[0] += lambda_loss * P   (loss_sum may be NULL)
 * with torch's sub-gradients (max ties 1/2:1/2, clamp(min=0) passes the gradient at 0). */
int stdadk_delta_head_f32(const float *delta, int64_t ldd, int32_t Q, int32_t d, float *Wo,
                          float *bo, stdadk_stream_t stream);
int stdadk_delta_head_backward_f32(const float *delta, const float *dWo, const float *dbo,
                                   int64_t ldd, int32_t Q, int32_t d, float lambda_grad,
                                   float lambda_loss, float *d_delta, float *loss_sum,
                                   stdadk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A9  clip_grad_norm_ + AdamW + EMA on flat fp32 buffers of n elements
 * (scripts/train_st_interp.py:696-712, stnf/utils/ema.py:52-66, torch.optim.AdamW).
 *   stdadk_sumsq_f32 : parts[0..STDADK_SUMSQ_PARTS) = partial sums of g^2 (all entries written, fixed
 *                      order => reproducible; several tensors may each fill their own block of
 *                      partials).  After an all-reduce of the gradients this gives the post-reduce
 *                      global norm the clip must use.
 *   stdadk_adamw_ema_f32: ss = sum(sumsq_parts[0..n_parts));
 *       coef = min(1, max_norm/(sqrt(ss)+1e-6)) if max_norm > 0 else 1
 *       g' = g*coef*grad_mul; p *= 1-lr*wd; m = b1*m+(1-b1)*g'; v = b2*v+(1-b2)*g'^2;
 *       p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps);
 *       ema = decay*ema + (1-decay)*p   (skipped when ema == NULL)
 *   lr is read from the device scalar lr_dev[0] when non-NULL (graph-replayable schedules),
 *   else from `lr`.  The 1-based number of the step being applied is `step` (host value) or, when
 *   step_dev is non-NULL, the device scalar step_dev[0].  That counter also feeds the dropout
 *   generator (as "steps completed") in the forward/backward of a step, so it must be advanced
 *   AFTER the backward and BEFORE stdadk_adamw_ema_f32: pass it as `step_inc` to
 *   stdadk_sumsq_f32 (block 0 adds 1) or call stdadk_step_advance when no clipping is used.
 * ------------------------------------------------------------------------------------------ */
int stdadk_sumsq_f32(const float *g, int64_t n, float *parts, int32_t *step_inc,
                     stdadk_stream_t stream);

/* BASELINE config C3 ("bf16 MLP with MFMA"): fp32 MASTER weights, bf16 OPERAND copies.  A shadow table names
 * up to STDADK_MAX_HIDDEN regions of a flat fp32 parameter buffer, each a (rows, cols) row-major matrix at
 * element offset `off` (off and cols multiples of 4), with the bf16 copy `dst` [rows][cols] and its transpose
 * `dst_t` [cols][rows] (either may be NULL).  stdadk_bf16_shadow_refresh rewrites every region from p (after
 * load_state_dict, an EMA swap, ...); the optimiser entry points below take the table as an optional last
 * argument and rewrite the copies from the values they have just stepped, in the same pass (offsets relative
 * to the `p` of that call).  Rounding: round-to-nearest-even (v_cvt_pk_bf16_f32). */
typedef struct stdadk_bf16_region {
  int64_t off;          /* first element of the matrix in the flat buffer                     */
  int32_t rows, cols;
  uint16_t *dst;        /* [rows][cols] bf16 or NULL                                          */
  uint16_t *dst_t;      /* [cols][rows] bf16 or NULL                                          */
} stdadk_bf16_region;
typedef struct stdadk_bf16_shadow {
  int32_t n;                                     /* regions in use                            */
  stdadk_bf16_region r[STDADK_MAX_HIDDEN];
} stdadk_bf16_shadow;
int stdadk_bf16_shadow_refresh(const float *p, const stdadk_bf16_shadow *shadow, stdadk_stream_t stream);

int stdadk_step_advance(int32_t *step_dev, stdadk_stream_t stream);

/* Non-finite guard (ABI 7; scripts/train_st_interp.py:724-733 leaves the epoch at the first batch whose loss is NaN).
 * `loss_watch` is the running objective accumulator the step's kernels add into (the loss_sum of the step-level entry
 * points), `nonfinite_step` a device int32 that starts at 0: the optimiser launch of a step -- which runs after the
 * step's last addition to the accumulator -- stores the 1-based number of the step it applies there when the
 * accumulator is not finite and the word is still 0.  A running sum stays non-finite once it is, so the word ends up
 * holding the FIRST step whose objective was not finite; the host reads it when it reads the loss (no extra sync per
 * step).  Both NULL = off. */
int stdadk_adamw_ema_f32(float *p, const float *g, float *m, float *v, float *ema, int64_t n,
                         float lr, const float *lr_dev, float beta1, float beta2, float eps,
                         float weight_decay, int32_t step, const int32_t *step_dev, float max_norm,
                         const float *sumsq_parts, int32_t n_parts, float grad_mul, float ema_decay,
                         const stdadk_bf16_shadow *shadow, const float *loss_watch, int32_t *nonfinite_step,
                         stdadk_stream_t stream);

/* A9 with two parameter groups in one launch each (learnable knots: the MLP parameters and the knot
 * tensors are clipped on their own norms and stepped with their own learning rates,
 * scripts/train_st_interp.py:470-483,698-705): stdadk_sumsq2_f32 = stdadk_sumsq_f32 on two buffers
 * (step_inc advanced once), stdadk_adamw_ema2_f32 = stdadk_adamw_ema_f32 on two groups. */
typedef struct stdadk_adam_group {
  float *p;                   /* parameters, gradients, Adam moments, EMA shadow (NULL = no EMA) */
  const float *g;
  float *m, *v, *ema;
  int64_t n;
  float lr;                   /* used when lr_dev is NULL                                        */
  const float *lr_dev;
  float max_norm;             /* <= 0: no clipping                                               */
  const float *sumsq_parts;   /* partial sums of squares of g (stdadk_sumsq*_f32)                */
  int32_t n_parts;
  const stdadk_bf16_shadow *shadow;   /* bf16 operand copies to keep current (NULL = none)       */
} stdadk_adam_group;
int stdadk_sumsq2_f32(const float *g0, int64_t n0, float *parts0, const float *g1, int64_t n1,
                      float *parts1, int32_t *step_inc, stdadk_stream_t stream);
int stdadk_adamw_ema2_f32(const stdadk_adam_group *g0, const stdadk_adam_group *g1, float beta1,
                          float beta2, float eps, float weight_decay, int32_t step,
                          const int32_t *step_dev, float grad_mul, float ema_decay,
                          const float *loss_watch, int32_t *nonfinite_step, stdadk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Step-level entry points: observations in, predictions / gradients out.
 *
 * One call = the whole of STInterpMLP.forward (st_interp.py:827-882) [+ nn.MSELoss and
 * loss.backward(), scripts/train_st_interp.py:617-621,693].  Two interchangeable paths:
 *   dense  : stdadk_rbf_build_f32 materialises [X|phi|psi] and every layer is a dense MFMA GEMM;
 *   window : for compact-support bases (Wendland, triangular) on the uniform multi-resolution
 *            grid of _init_uniform (st_interp.py:152-185) only the <= 6x6 knots per level that can
 *            be non-zero for an observation are touched: observations are binned into a G x G
 *            cell grid, layer 0 gathers W0^T rows per observation (fused with LayerNorm/ReLU/
 *            Dropout) and every row of dW0^T is accumulated by the one wave that owns it.  phi is
 *            evaluated with the same arithmetic as the dense path; exact zeros are skipped.
 * The window path needs the first layer's weight (and gradient) stored TRANSPOSED, (in,out)
 * row-major (flag STDADK_FLAG_W0_T), hidden[0] in {128, 256} and p <= 16; otherwise, or with
 * STDADK_FLAG_DENSE, the dense path runs (it accepts either W0 layout).
 * ------------------------------------------------------------------------------------------ */
typedef struct stdadk_basis_desc {
  int32_t p;                        /* covariate columns in front of the basis block            */
  int32_t basis;                    /* STDADK_BASIS_*                                            */
  int32_t n_levels;                 /* uniform grid levels; 0 = knots are not a uniform grid     */
  int32_t side[STDADK_MAX_LEVELS];  /* level l is side x side, knot index = ix*side + iy         */
  int64_t Ks, Kt;                   /* spatial / temporal knot counts                            */
  const float *s_centers;           /* [Ks,2]  device                                            */
  const float *s_bw;                /* [Ks]    device                                            */
  const float *t_centers;           /* [Kt]    device                                            */
  const float *t_bw;                /* [Kt]    device                                            */
} stdadk_basis_desc;

#define STDADK_FLAG_DENSE 1 /* force the materialising path                                    */
#define STDADK_FLAG_W0_T 2  /* params->W[0] and grads->W[0] are (in,out) row-major              */
#define STDADK_FLAG_PREBINNED 16 /* window path: the batch was already binned into this workspace by
                              * stdadk_bin_batch_f32 (e.g. on another stream while the previous step
                              * ran); the step entry points skip the binning and ignore coords/t/X/y */
#define STDADK_FLAG_BF16 32 /* BASELINE config C3: the Linear layers AFTER the first run on the bf16 matrix
                              * cores (v_mfma_f32_16x16x32_bf16 in the fused tail kernels, v_mfma_f32_32x32x16_bf16
                              * in the grouped weight-gradient products): operands rounded to bf16 at the
                              * operand boundary (activations as they enter LDS, weights from
                              * params->W_bf16 / WT_bf16), fp32 accumulation, fp32 LayerNorm statistics,
                              * fp32 loss, fp32 master weights / gradients / optimiser state.  phi and psi are
                              * evaluated in fp32 and layer 0 stays fp32 on both paths.  Needs the fused tail
                              * kernels (hidden widths multiples of 16 up to 256); STDADK_E_ARG otherwise */
#define STDADK_FLAG_SCATTERED 64 /* the spatial knots are NOT a uniform grid (gmm / random_site initialisers,
                              * st_interp.py:187-343; learnable or fixed): basis->n_levels levels whose SIZES are
                              * basis->side[l] (knot COUNT of level l, any positions; sum = Ks).  The window path
                              * then bins the knots of every level into a 32 x 32 cell grid each step and an
                              * observation gathers the knots of the cells within the level's largest support
                              * radius of its own (same sums as the materialising path, zeros skipped).  Worth it
                              * when the supports are small against the domain; the caller decides (the Python
                              * face: expected candidates <= 35 % of the knots)                                  */
#define STDADK_FLAG_WINDOW 8 /* take the window path whenever it is supported, even for small knot
                              * tables (default: tables under 1024 knots run the materialising
                              * path, which is faster there: a coarse level's few knots each own a
                              * large share of the batch in the per-knot gather of dW0^T)          */
#define STDADK_FLAG_LOG_BW 4 /* learnable knots: basis->s_bw holds LOG-bandwidths (the parameter is
                              * log(bw), used as exp() of it: st_interp.py:101-102,143-148) and the
                              * centres may have moved off the grid.  With n_levels > 0 (the knots
                              * STARTED as the uniform grid, knot k still indexed ix*side+iy) the
                              * window path still applies: every step it widens the candidate
                              * window per level to ceil(max_k (s_k + |c_k - grid_k|_inf)(side-1))
                              * cells, which is guaranteed to contain every knot whose support
                              * reaches an observation; n_levels == 0 runs the materialising path */

/* 1 when the (basis, mlp, flags) combination runs the window path, else 0. */
int32_t stdadk_step_uses_window(const stdadk_basis_desc *basis, const stdadk_mlp_desc *mlp,
                                int32_t flags);
size_t stdadk_step_workspace_bytes(const stdadk_basis_desc *basis, const stdadk_mlp_desc *mlp,
                                   int64_t B, int32_t flags);

/* A2-A6 / A10  forward of one batch: y_pred[B,Q] in the caller's row order.
 *   training == 0: eval mode (evaluate_model / dense-grid prediction, scripts/train_st_interp.py:
 *                  884-961,1091-1107,1232-1248,1378-1409): no dropout.
 *   training != 0: train mode; dropout keep-masks come from drop_seed + step_dev[0] (device int32
 *                  step counter, may be NULL; a captured graph draws fresh masks per replay), and
 *                  the workspace keeps what stdadk_backward_f32 needs. */
int stdadk_forward_f32(const stdadk_basis_desc *basis, const stdadk_mlp_desc *mlp,
                       const stdadk_mlp_tensors *params, const float *coords, const float *t,
                       const float *X, int64_t B, float *y_pred, void *workspace,
                       size_t workspace_bytes, int32_t training, uint64_t drop_seed,
                       const int32_t *step_dev, int32_t flags, stdadk_stream_t stream);

/* A8  loss.backward() (scripts/train_st_interp.py:693) for the batch whose TRAINING forward left
 * its state in `workspace` (same basis/mlp/B/flags/drop_seed/step_dev): dY[B,Q] = dLoss/dy_pred in
 * the caller's row order; every parameter gradient is overwritten in `grads`. */
int stdadk_backward_f32(const stdadk_basis_desc *basis, const stdadk_mlp_desc *mlp,
                        const stdadk_mlp_tensors *params, const stdadk_mlp_tensors *grads, int64_t B,
                        const float *dY, void *workspace, size_t workspace_bytes, uint64_t drop_seed,
                        const int32_t *step_dev, int32_t flags, stdadk_stream_t stream);

/* N2  learnable knots (DA-STDK; st_interp.py:94-150, scripts/train_st_interp.py:660-672): gradients
 * of the batch loss w.r.t. the spatial centres [Ks,2] and LOG-bandwidths [Ks], for the batch whose
 * backward (stdadk_backward_f32 / stdadk_train_fwd_bwd_f32 with STDADK_FLAG_LOG_BW) has just left dZ
 * of the first layer (materialising path) or the raw per-knot sums (window path: the wave that owns a
 * knot's row of dW0^T accumulates them in the same pass) in `workspace`:
 *   G = dZ0 . W0[:, p:p+Ks]                                  (what autograd sends into phi)
 *   r = |x - c_k| / (exp(log_bw_k) cal);   d_c_k = sum_b G phi'(r) (-(x-c_k)/(|x-c_k| s_k)) (0 at zero
 *   distance, as cdist's backward has it);   d_log_bw_k = sum_b G phi'(r) (-r)
 * then, when `kt` is given, what the training driver adds on top (all per knot, fixed order):
 *   + penalty_grad_scale * d/dc [ domain_weight  * sum (max(0,-c) + max(0,c-1))^2      (:493-526)
 *                               + movement_weight * sum |c - centers_init|^2 ]         (:528-546)
 *   x exp(-damping_strength * max(|c - centers_init| - damping_threshold, 0))  on d_c  (:111-141, the
 *     gradient hook; applied after the penalties, as the hook sees the accumulated gradient)
 *   loss_sum[0] += penalty_loss_scale * (weighted penalties)            (loss_sum may be NULL)
 * kt == NULL gives the plain data gradient (the module-level autograd path: the hook and the
 * penalties then stay with the caller).  coords are the batch's [B,2] again (unused, may be NULL, on
 * the window path); d_centers / d_log_bw are overwritten. */
typedef struct stdadk_knot_train {
  const float *centers_init;   /* [Ks,2] device; required for damping / movement               */
  int32_t gradient_damping;    /* 0 / 1                                                         */
  float damping_threshold, damping_strength;
  float domain_weight, movement_weight;
  float penalty_grad_scale;    /* 1, or 1/world_size when the ranks' gradients are summed       */
  float penalty_loss_scale;    /* rows*Q of this rank's batch: loss_sum is in those units        */
} stdadk_knot_train;
int stdadk_knot_backward_f32(const stdadk_basis_desc *basis, const stdadk_mlp_desc *mlp,
                             const stdadk_mlp_tensors *params, const float *coords, int64_t B,
                             void *workspace, size_t workspace_bytes, int32_t flags,
                             const stdadk_knot_train *kt, float *d_centers, float *d_log_bw,
                             float *loss_sum, stdadk_stream_t stream);

/* N2 at module level: the backward of SpatialBasisEmbedding.forward (st_interp.py:433-460) on its own, for callers
 * that differentiate through `model.spatial_basis(coords)` directly rather than through STInterpMLP.forward:
 * d_phi [B, ld] = dLoss/dphi (column k <-> knot k), centres [Ks,2], LOG-bandwidths [Ks] ->
 *   d_centers[k] = sum_b d_phi[b,k] phi'(r) (-(x-c_k)/(|x-c_k| s_k)),  d_log_bw[k] = sum_b d_phi[b,k] phi'(r) (-r)
 * (both overwritten; no damping hook, no penalties: those stay with the caller's autograd). */
size_t stdadk_knot_grad_workspace_bytes(int64_t B, int64_t Ks);
int stdadk_knot_grad_f32(const float *coords, int64_t B, const float *d_phi, int64_t ld,
                         const float *centers, const float *log_bw, int64_t Ks, int32_t basis,
                         float *d_centers, float *d_log_bw, void *workspace, size_t workspace_bytes,
                         stdadk_stream_t stream);

/* A2-A8 in one call: training forward, the batch objective and its gradient, backward:
 *   loss == NULL (nn.MSELoss): loss_sum[0] += sum((y_pred-y)^2), grads = d/dparams of
 *   grad_scale * sum((y_pred-y)^2), y [B,Q];
 *   loss != NULL: the objective of stdadk_loss_f32 (N3), y [B, loss->y_cols].
 * (grad_scale = 1/(rows*Q) of the GLOBAL batch, so summing the ranks' gradients gives the global
 * mean.)  y_pred [B,Q] (caller's row order) is optional: NULL skips writing it.
 * aux_stream (optional, NULL = none): a second stream onto which independent kernels of the step
 * are forked (event fork/join, capturable); on return every kernel has been joined into `stream`. */
int stdadk_train_fwd_bwd_f32(const stdadk_basis_desc *basis, const stdadk_mlp_desc *mlp,
                             const stdadk_mlp_tensors *params, const stdadk_mlp_tensors *grads,
                             const float *coords, const float *t, const float *X, const float *y,
                             int64_t B, float grad_scale, const stdadk_loss_desc *loss,
                             float *loss_sum, float *y_pred, void *workspace, size_t workspace_bytes,
                             uint64_t drop_seed, const int32_t *step_dev, int32_t flags,
                             stdadk_stream_t stream, stdadk_stream_t aux_stream);

/* A0 + A2-A8: the same step on rows idx[b] (int64, device) of device-RESIDENT observation arrays
 * coords_all [N,2], t_all [N], X_all [N,p], y_all [N, y cols]: the window path's binning reads the
 * rows in place, so the batch producer (scripts/train_st_interp.py:413-460,609-612) costs no launch
 * of its own.  Window path only (STDADK_E_ARG otherwise: gather with stdadk_gather_batch_f32 and call
 * stdadk_train_fwd_bwd_f32).  y_pred, when given, is in batch order (row b <-> idx[b]). */
int stdadk_train_fwd_bwd_indexed_f32(const stdadk_basis_desc *basis, const stdadk_mlp_desc *mlp,
                                     const stdadk_mlp_tensors *params, const stdadk_mlp_tensors *grads,
                                     const float *coords_all, const float *t_all, const float *X_all,
                                     const float *y_all, const int64_t *idx, int64_t B, float grad_scale,
                                     const stdadk_loss_desc *loss, float *loss_sum, float *y_pred,
                                     void *workspace, size_t workspace_bytes, uint64_t drop_seed,
                                     const int32_t *step_dev, int32_t flags, stdadk_stream_t stream,
                                     stdadk_stream_t aux_stream);

/* Sparsity penalties on the first Linear's weights: STInterpMLP.compute_sparsity_penalty /
 * _compute_penalty_for_block (stnf/models/st_interp.py:724-825) as the batch body adds them to the loss
 * before loss.backward() (scripts/train_st_interp.py:674-691).  A basis function's group is its row of
 * W0^T (w0_t = 1: W0 stored (D, H0), the engine's layout) or its column of W0 (w0_t = 0: (H0, D), the
 * nn.Linear layout); the p covariate rows carry no penalty (:763-765).
 *   ELEMENT       lambda_l1 * sum |w|
 *   GROUP         lambda_group * sum over basis functions of ||w_j||_2
 *   SPARSE_GROUP  both
 * dW0 += grad_scale * d(applied penalties)/dW0 with torch's sub-gradients (sign(0) = 0; an all-zero group has
 * gradient 0); loss_sum += loss_scale * (applied penalties); penalties[0] += spatial, penalties[1] += temporal
 * value (reported whether applied or not, like the reference's dict).  dW0, loss_sum, penalties may be NULL. */
#define STDADK_SPARSITY_NONE 0
#define STDADK_SPARSITY_ELEMENT 1
#define STDADK_SPARSITY_GROUP 2
#define STDADK_SPARSITY_SPARSE_GROUP 3
typedef struct {
  int32_t kind;            /* STDADK_SPARSITY_*                                             */
  float lambda_l1;         /* sparsity_lambda_l1                                            */
  float lambda_group;      /* sparsity_lambda_group                                         */
  int32_t apply_spatial;   /* sparsity_apply_to_spatial  (train_st_interp.py:679,688)       */
  int32_t apply_temporal;  /* sparsity_apply_to_temporal (train_st_interp.py:680,690)       */
} stdadk_sparsity_desc;
int stdadk_sparsity_f32(const stdadk_sparsity_desc *s, const float *W0, float *dW0, int64_t ld,
                        int32_t w0_t, int32_t H0, int32_t p, int32_t Ks, int32_t Kt, float grad_scale,
                        float loss_scale, float *loss_sum, float *penalties, stdadk_stream_t stream);

/* A0 + A2-A9 in ONE call (single GPU): stdadk_train_fwd_bwd(_indexed)_f32, then clip_grad_norm_ +
 * AdamW + EMA (stdadk_sumsq_f32 + stdadk_adamw_ema_f32) on the flat buffers of `opt`, whose gradient
 * buffer [g, g+n) must contain every tensor of `grads` (gaps zero).  Knowing the whole step lets the
 * library take the gradient's squared norm out of the launches that produce it (window path: no
 * separate pass over the gradient); otherwise it runs the separate kernels.  idx may be NULL (rows
 * 0..B-1 of the given arrays).  opt->step_dev (device int32, required) is read by dropout and AdamW and
 * advanced once.  `sparsity` (may be NULL): the first-layer penalties of stdadk_sparsity_f32, their
 * gradient added before clipping and their value x B x Q into loss_sum, like the batch body does.
 * Data-parallel training keeps the split calls: the all-reduce sits between them. */
#define STDADK_GRADSQ_PARTS 512
typedef struct stdadk_optim_desc {
  float *p, *g, *m, *v, *ema;   /* flat fp32 buffers of n elements; ema may be NULL                 */
  int64_t n;
  float lr;                     /* used when lr_dev is NULL                                        */
  const float *lr_dev;
  float beta1, beta2, eps, weight_decay;
  int32_t *step_dev;            /* device step counter (starts at 0)                               */
  float max_norm;               /* clip_grad_norm_ threshold; <= 0: no clipping                    */
  float *sumsq_parts;           /* [STDADK_GRADSQ_PARTS] device scratch                            */
  float ema_decay;
  const stdadk_bf16_shadow *shadow;   /* bf16 operand copies to keep current (NULL = none)           */
  int32_t *nonfinite_step;      /* non-finite guard on the call's loss_sum (see stdadk_adamw_ema_f32), or NULL */
} stdadk_optim_desc;
int stdadk_train_step_f32(const stdadk_basis_desc *basis, const stdadk_mlp_desc *mlp,
                          const stdadk_mlp_tensors *params, const stdadk_mlp_tensors *grads,
                          const float *coords, const float *t, const float *X, const float *y,
                          const int64_t *idx, int64_t B, float grad_scale, const stdadk_loss_desc *loss,
                          const stdadk_sparsity_desc *sparsity, float *loss_sum, void *workspace,
                          size_t workspace_bytes, uint64_t drop_seed, int32_t flags,
                          const stdadk_optim_desc *opt, stdadk_stream_t stream);

/* stdadk_train_step_f32 on rows `idx` of the RESIDENT arrays, which also prepares the NEXT batch (ABI 8): when the
 * next batch takes the one-launch binning of small batches (window path, next_B <= 8192), rows next_idx of the
 * same arrays are binned into `next_workspace` by extra workgroups of this step's optimiser launch -- no launch of
 * its own, no second stream -- and *next_binned is set to 1: the caller then steps on next_workspace with
 * STDADK_FLAG_PREBINNED (as after stdadk_bin_batch_f32).  Otherwise *next_binned = 0 and nothing was prepared.
 * The resident arrays are always given (also when THIS step runs with STDADK_FLAG_PREBINNED: it reads its rows from
 * `workspace` then); next_y_cols as y_cols of stdadk_bin_batch_f32.  next_workspace must not be `workspace`. */
int stdadk_train_step_next_f32(const stdadk_basis_desc *basis, const stdadk_mlp_desc *mlp,
                               const stdadk_mlp_tensors *params, const stdadk_mlp_tensors *grads,
                               const float *coords_all, const float *t_all, const float *X_all, const float *y_all,
                               const int64_t *idx, int64_t B, float grad_scale, const stdadk_loss_desc *loss,
                               const stdadk_sparsity_desc *sparsity, float *loss_sum, void *workspace,
                               size_t workspace_bytes, uint64_t drop_seed, int32_t flags, const stdadk_optim_desc *opt,
                               const int64_t *next_idx, int64_t next_B, int32_t next_y_cols, void *next_workspace,
                               size_t next_workspace_bytes, int32_t *next_binned, stdadk_stream_t stream);

/* A10 on a site x time prediction grid (the dense inference callers loop over time slices with the SAME S
 * sites in each, scripts/train_st_interp.py:1091-1107,1232-1248,1378-1409).  Layer 0's pre-activation of row
 * (ti, s) is  sum_k phi_k(s) W0^T[p+k,:]  +  sum_j psi_j(t_ti) W0^T[p+Ks+j,:]  +  b0 : a per-site row plus a per-time
 * row, so the basis evaluation and the W0^T gather are done once per site instead of once per (site, time):
 *   stdadk_spatial_partial_f32   sp[S][h0]  (window path, fixed grid knots, p = 0; workspace as for a forward of S rows)
 *   stdadk_temporal_partial_f32  tp[T][h0]
 *   stdadk_forward_parts_f32     y_pred[T*S][Q], rows time-major: z0 = sp[row % S] + tp[row / S] + b0, then
 *                                LayerNorm/ReLU of layer 0, the remaining layers and the output layer (eval mode)
 * Same sums as stdadk_forward_f32 in another order of addition (agreement to fp32 rounding, tested).
 * All three need the first weight stored (in,out) (STDADK_FLAG_W0_T). */
int stdadk_spatial_partial_f32(const stdadk_basis_desc *basis, const stdadk_mlp_desc *mlp,
                               const stdadk_mlp_tensors *params, const float *coords, int64_t S, float *out,
                               void *workspace, size_t workspace_bytes, int32_t flags, stdadk_stream_t stream);
int stdadk_temporal_partial_f32(const stdadk_basis_desc *basis, const stdadk_mlp_desc *mlp,
                                const stdadk_mlp_tensors *params, const float *t_values, int64_t T, float *out,
                                int32_t flags, stdadk_stream_t stream);
int stdadk_forward_parts_f32(const stdadk_mlp_desc *mlp, const stdadk_mlp_tensors *params, const float *sp,
                             int64_t S, const float *tp, int64_t T, float *y_pred, stdadk_stream_t stream);

/* A0, window path: the batch-preparation half of the step on its own — gathers rows idx[b] (NULL = rows
 * 0..B-1) of coords_all / t_all / X_all / y_all [N, y_cols] and bins them into `workspace`, exactly as
 * the step entry points would.  Software pipelining: call it for batch k+1 on a second stream (and a
 * second workspace) while step k runs, then run step k+1 with STDADK_FLAG_PREBINNED.  The caller
 * orders the streams (the workspace must not be in use by an unfinished step). */
int stdadk_bin_batch_f32(const stdadk_basis_desc *basis, const stdadk_mlp_desc *mlp,
                         const float *coords_all, const float *t_all, const float *X_all,
                         const float *y_all, const int64_t *idx, int64_t B, int32_t y_cols,
                         void *workspace, size_t workspace_bytes, int32_t flags,
                         stdadk_stream_t stream);

/* A0  batch producer (scripts/train_st_interp.py:413-460 dataset + collate, :609-612 H2D): rows
 * idx[b] (int64) of the device-resident observation arrays into contiguous batch buffers, one launch.
 * coords [N,2], t [N], y [N,Q], X [N,p] (NULL when p == 0). */
int stdadk_gather_batch_f32(const float *coords, const float *t, const float *y, const float *X,
                            const int64_t *idx, int64_t B, int32_t Q, int32_t p, float *coords_out,
                            float *t_out, float *y_out, float *X_out, stdadk_stream_t stream);

/* Integer bookkeeping of the window path, exposed for bit-exact tests:
 *   stdadk_bin_obs_f32: cell key of every observation (cx*G+cy, cx = clamp(floor(x*G),0,G-1)),
 *     cell_start[G*G+1] (first sorted position of every cell) and perm[B] (sorted position ->
 *     original index; row-major cells, ascending original index inside a cell).
 *   stdadk_knot_windows_i32: per observation and level the first knot (ix0, iy0) of the window
 *     [floor(x*(side-1))-2, +6) clamped into the grid, and its feature column
 *     col0 = p + level_offset + ix0*side + iy0.  Arrays are [B, n_levels] int32. */
size_t stdadk_bin_workspace_bytes(int64_t B, int32_t G);
int stdadk_bin_obs_f32(const float *coords, int64_t B, int32_t G, int32_t *keys,
                       int32_t *cell_start, int32_t *perm, void *workspace, size_t workspace_bytes,
                       stdadk_stream_t stream);
int stdadk_knot_windows_i32(const float *coords, int64_t B, const int32_t *sides_host,
                            int32_t n_levels, int32_t p, int32_t *ix0, int32_t *iy0, int32_t *col0,
                            stdadk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The fp32 MFMA GEMM the Linear layers are built from (nn.Linear forward / autograd backward:
 * st_interp.py:660,689), exposed so that tests and the roofline measurement can drive it directly.
 *   C[M,N] (+bias[N]) = sum_k Aop[m][k] * Bop[k][n]
 *   a_km == 0: Aop[m][k] = A[m*lda + k]      a_km != 0: Aop[m][k] = A[k*lda + m]
 *   b_km == 0: Bop[k][n] = B[n*ldb + k]      b_km != 0: Bop[k][n] = B[k*ldb + n]
 * (a_km != 0 with b_km == 0 is not built.)  workspace: stdadk_gemm_workspace_bytes(M,N,K) bytes
 * of split-K slab space (may be 0 / NULL when the shape does not split).
 * ------------------------------------------------------------------------------------------ */
size_t stdadk_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K);

int stdadk_gemm_f32(const float *A, int64_t lda, int32_t a_km, const float *B, int64_t ldb,
                    int32_t b_km, int32_t M, int32_t N, int32_t K, const float *bias, float *C,
                    int64_t ldc, void *workspace, size_t workspace_bytes, stdadk_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Measurement aid: when enabled, every kernel the library launches is bracketed by two HIP events
 * on its launch stream.  stdadk_profile_collect synchronises them and writes one "name\tms\n" line
 * per launch (launch order) into buf; it returns the bytes needed including the terminator (call
 * again with a bigger buffer if that exceeds cap).  Enabling clears earlier records.  Not for use
 * under stream capture; host-side, single-threaded.
 * ------------------------------------------------------------------------------------------ */
int stdadk_profile_enable(int32_t on);
int64_t stdadk_profile_collect(char *buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* STDADK_H */
