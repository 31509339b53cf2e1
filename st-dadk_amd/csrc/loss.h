// Batch objectives fused with their gradient (MSE, check loss + non-crossing penalty) and the
// delta-reparameterised head.  loss_elem() is the one arithmetic both the standalone kernel and
// the epilogue of the fused tail kernel use.
#pragma once
#include "common.h"

namespace stdadk {

struct LossDev {
  int kind;      // STDADK_LOSS_*
  int y_cols;    // 1 or Q
  float tau[STDADK_MAX_Q];
  float nc_w;    // 0 = off
  int nc_pow;    // 1 or 2
};

// NULL = MSE against y [B,Q].  Returns non-zero (error text set) on a bad descriptor.
int make_loss(const stdadk_loss_desc *l, int Q, LossDev *out);
inline bool loss_is_plain_mse(const LossDev &L, int Q) { return L.kind == STDADK_LOSS_MSE && L.y_cols == Q; }

int launch_loss(const LossDev &L, const float *yp, const float *y, int64_t B, int Q, float scale, float *dY,
                float *loss_sum, hipStream_t st);

// tau of output q without indexing the kernel-argument array by a vector register
__device__ __forceinline__ float loss_tau(const LossDev &L, int q) {
  float t = L.tau[0];
#pragma unroll
  for (int k = 1; k < STDADK_MAX_Q; ++k) t = q == k ? L.tau[k] : t;
  return t;
}

// One (row, q) element: prediction yq, its neighbours yup = y_pred[q+1], ydn = y_pred[q-1] (read only
// when they exist), target yt.  Returns the element's share of the loss sum, dy = its gradient.
__device__ __forceinline__ float loss_elem(const LossDev &L, int Q, int q, float tau, float yq, float yup,
                                           float ydn, float yt, float gs, float &dy) {
  if (L.kind == STDADK_LOSS_MSE) {
    const float d = yq - yt;
    dy = 2.0f * d * gs;
    return d * d;
  }
  const float e = yt - yq;
  float term = fmaxf((tau - 1.0f) * e, tau * e);
  float g = e > 0.f ? -tau : (e < 0.f ? 1.0f - tau : 0.5f - tau);
  if (L.nc_w > 0.f) {
    float f = 0.f, gn = 0.f;
    if (q + 1 < Q) {
      const float d = yq - yup;
      if (d > 0.f) { f = L.nc_pow == 1 ? d : d * d; gn = L.nc_pow == 1 ? 1.0f : 2.0f * d; }
    }
    if (q > 0) {
      const float d = ydn - yq;
      if (d > 0.f) gn -= L.nc_pow == 1 ? 1.0f : 2.0f * d;
    }
    const float w = L.nc_w * (float)Q;
    term = fmaf(w, f, term);
    g = fmaf(w, gn, g);
  }
  dy = g * gs;
  return term;
}

}  // namespace stdadk
