// Index-window (compact-support) path of the first layer: observation binning + fused
// basis -> Linear(+LayerNorm+ReLU+Dropout) forward + owner-computes dW backward.
#pragma once
#include "common.h"

namespace stdadk {

constexpr int WIN = 6;          // knots per axis that can lie within 2.5 spacings of a point
constexpr int WIN_MAX_P = 16;   // covariates handled by the window path

// Device view of the uniform multi-resolution knot grid (A1, stnf/models/st_interp.py:152-185):
// level l is a side[l] x side[l] grid, knot index off[l] + ix*side + iy.
struct GridView {
  int n_levels;
  int side[STDADK_MAX_LEVELS];
  int off[STDADK_MAX_LEVELS];
  int scattered;         // STDADK_FLAG_SCATTERED: the knots of a level sit anywhere (gmm / random_site initialisers,
  int cnt[STDADK_MAX_LEVELS];   // st_interp.py:187-343); level l = knots [off[l], off[l] + cnt[l]), side[] unused
  int p;                 // feature column of spatial knot 0
  int Ks, Kt;
  float cal;             // calibration factor of the basis
  const float *centers;  // [Ks][2]
  const float *bw;       // [Ks]
  const float *t_centers;
  const float *t_bw;
};

// Buffers of the observation binning (all device memory, caller-carved).
struct BinBuffers {
  int *keys;        // [B]     cell of every observation (original order)
  int *hist;        // [G*G]   zeroed by bin_obs
  int *cursor;      // [G*G]
  int *cell_start;  // [G*G+1] first sorted position of every cell
  int *perm_tmp;    // [B]
  int *perm;        // [B]     sorted position -> original index (ascending inside a cell)
  float *xs, *ys, *ts;  // [B] sorted coordinates / times
  float *y_s;       // [B*Q]   sorted targets (may be NULL)
  float *X_s;       // [B*p]   sorted covariates (may be NULL)
};

int pick_cell_grid(int64_t B);   // G: cells per axis, power of two in [8, 256]
struct BinSmallArgs;
// the one-launch binning of small batches (bin_body.h) as arguments another launch can carry
BinSmallArgs bin_small_args(const float *coords, const float *t, const float *y, int Q, const float *X, int p, int B,
                            int G, const BinBuffers &bb, const int64_t *idx);
bool bin_small_eligible(int B, int G);

// idx (optional): the batch is rows idx[b] of resident arrays coords/t/y/X; perm holds batch positions
int bin_obs(const float *coords, const float *t, const float *y, int Q, const float *X, int p,
            int B, int G, const BinBuffers &bb, hipStream_t st, const int64_t *idx = nullptr,
            bool many_small = false);   // many_small: never the one-workgroup kernel (see stdadk_bin_batch_f32)

struct L1FwdArgs {
  GridView g;
  const float *xs, *ys, *ts, *Xs;
  int B, H;
  const float *W0T;     // [D][H]
  const float *b0, *gamma, *beta;
  float eps;
  float *xhat, *rstd, *act, *psi;   // xhat / rstd / psi may be NULL (eval mode: kept only for a backward)
  int ld_psi;
  float drop_p;
  uint64_t seed;
  const int *step_dev;
  int rows_per_wg, n_wg;
  // free (learnable) knots: per-level half-width R of the candidate window around the observation's
  // grid cell, from the device floats [level][HALO_SPLIT] written by knot_halo() every step (see
  // halo_half_width); NULL = fixed grid knots (R = 3)
  const float *halo;
  // scattered knots (g.scattered): per level the knots binned into a Gk x Gk cell grid by knot_bins() --
  // kcs[l][Gk*Gk + 1] first position of every cell in kperm (positions are global: level l's run starts at
  // g.off[l]), kperm[Ks] the knot ids cell by cell (ascending inside a cell), reach[l] = max support radius of
  // the level.  An observation's candidates are the knots of the cells within ceil(reach Gk) + 1 of its own.
  const int *kcs = nullptr;
  const int *kperm = nullptr;
  const float *reach = nullptr;
  int Gk = 0;
  // raw != 0: only the spatial part of the pre-activation, sum_k phi_k(s) W0^T[p+k,:] (no bias, no temporal
  // rows, no LayerNorm/ReLU/Dropout), written to act -- the per-site half of a site x time prediction grid
  int raw = 0;
};

// z0 = [X|phi|psi] W0 + b0 -> LN -> ReLU -> Dropout for sorted observations; also writes psi.
int l1_window_forward(const L1FwdArgs &a, int basis, bool layernorm, hipStream_t st);
bool l1_window_supported(int n_levels, int basis, int H, int p, int Kt);
struct L1BwdArgs;
int knot_group_count(const GridView &g, int nk);  // groups of the per-knot gather of dW0^T, nk knots per wave
int knots_per_wave(const L1BwdArgs &a);
int knot_xcd_slots(const GridView &g, int nk);    // workgroups per XCD of the XCD-striped group order, 0 = off

struct L1BwdArgs {
  GridView g;
  const float *xs, *ys;
  const int *cell_start;
  int G, B, H;
  const float *dZ;      // [B][H] sorted order
  float *dW0T;          // [D][H]
  int xcd_slots = 0;    // > 0: two-knots-per-wave groups in XCD-striped order, this many workgroups per XCD
                        // (set by the launchers, see knot_xcd_slots)
  // learnable knots: W0^T (rows p + k) and the raw knot sums [3][Ks] (d cx, d cy, d log_bw) this kernel
  // also produces; NULL = fixed knots
  const float *W0T;
  float *kpart;
};

// Scattered knots: counting sort of every level's knots into a Gk x Gk cell grid (one workgroup per level;
// in-cell order by knot id, so the lists -- and with them every sum the forward takes in list order -- do not
// depend on the arrival order of the atomics), plus the level's largest support radius.  Runs every step: learnable
// knots move.  log_bw (optional): the bandwidths are given as logs; exp() of them is also written to bw_out.
constexpr int KNOT_CELLS = 32;     // Gk
int knot_bins(const GridView &g, int *kcs, int *kperm, int *kperm_tmp, float *reach, hipStream_t st,
              const float *log_bw = nullptr, float *bw_out = nullptr);

// Per level: R = ceil(max_k (s_k + |c_k - grid_k|_inf) * (side-1)), the half-width (in grid cells) of the
// candidate window that is guaranteed to contain every knot whose support reaches an observation,
// wherever the learnable knots have moved and however their bandwidths have changed.
// With log_bw the bandwidths are taken as exp(log_bw) and also written to bw_out (one launch for both).
constexpr int HALO_SPLIT = 16;     // workgroups (partial maxima) per level
int knot_halo(const GridView &g, float *halo, hipStream_t st, const float *log_bw = nullptr, float *bw_out = nullptr);

// dW0T[p + k, :] = sum_b phi[b,k] dZ[b,:] for every spatial knot k (each knot row owned by one
// wave: no atomics, summation in sorted-observation order => bitwise reproducible).
int l1_window_backward(L1BwdArgs a, int basis, hipStream_t st);

}  // namespace stdadk
#include "gemm_f32.h"
namespace stdadk {
// the grouped dW products of the other layers and this per-knot gather as ONE launch (dw_all.hip)
// fin (optional, fin->cnt != NULL): the launch also finishes the products (FinArgs: last-arriving K slice sums its
// tile), runs the tall reduce jobs of `tall` and leaves squared-norm slots; *n_slots = how many slots it writes
int launch_dw_all(GemmGroup &grp, const L1BwdArgs &a, int basis, hipStream_t st, const FinArgs *fin = nullptr,
                  ReduceGroup *tall = nullptr, int *n_slots = nullptr);
int dw_all_knot_blocks(const L1BwdArgs &a);      // knot workgroups of the launch (for sizing the slots)

// out[perm[i]*Q + q] = in[i*Q + q]
int unpermute_rows(const float *in, const int *perm, int B, int Q, float *out, hipStream_t st);

}  // namespace stdadk
