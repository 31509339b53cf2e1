// N2: learnable knots (DA-STDK) — gradients of the loss w.r.t. the spatial centres and log-bandwidths
// (st_interp.py:94-150,433-460), the gradient-damping hook (:111-141) and the domain / movement
// penalties (:493-546).
#pragma once
#include "common.h"

namespace stdadk {

constexpr int KNOT_TILE = 64;        // knots per workgroup (one per lane)
constexpr int KNOT_MAX_SLABS = 32;   // row slabs -> partial sums per knot

int knot_slabs(int64_t B);

// out[k] = exp(in[k])
int launch_exp(const float *in, int64_t n, float *out, hipStream_t st);

struct KnotGradArgs {
  const float *coords;     // [B][2]
  int B;
  const float *dFeat;      // [B][ld]: dL/d features; spatial column k at dFeat[b*ld + p + k]
  int64_t ld;
  int p;
  const float *centers;    // [Ks][2]
  const float *bw;         // [Ks] bandwidths (already exp'ed)
  int Ks;
  int basis;
  float *part;             // [slabs][3][Ks] partial sums (dcx, dcy, dlog_bw)
  int slabs;
};
int launch_knot_grad(const KnotGradArgs &a, hipStream_t st);

struct KnotFinishArgs {
  const float *part; int slabs; int Ks;
  const float *centers;        // [Ks][2]
  const float *centers_init;   // [Ks][2] or NULL
  int damping; float thr, strength;
  float dom_w, mov_w;          // penalty weights (0 = off)
  float pen_grad_scale, pen_loss_scale;
  float *d_centers;            // [Ks][2]
  float *d_log_bw;             // [Ks]
  float *loss_sum;             // or NULL
};
int launch_knot_finish(const KnotFinishArgs &a, hipStream_t st);

}  // namespace stdadk
