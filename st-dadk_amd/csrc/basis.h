// The radial basis arithmetic shared by the materialising builder and the window path, so that
// both produce bit-identical phi (st_interp.py:447-448 r = dist/(bw*cal); :470-471 Wendland C4;
// :481 Gaussian; :491 triangular).
#pragma once
#include "common.h"

namespace stdadk {

template <int BASIS>
__device__ __forceinline__ float basis_eval(float r) {
  if (BASIS == STDADK_BASIS_WENDLAND) {
    // (1-r)^6_+ (35 r^2 + 18 r + 3)/3 with r clamped to <= 1
    r = fminf(r, 1.0f);
    float om = 1.0f - r;
    float om2 = om * om;
    float om6 = om2 * om2 * om2;
    float poly = fmaf(fmaf(35.0f, r, 18.0f), r, 3.0f);
    return om6 * poly * (1.0f / 3.0f);
  } else if (BASIS == STDADK_BASIS_GAUSSIAN) {
    return expf(-0.5f * r * r);
  } else {
    return fmaxf(1.0f - r, 0.0f);
  }
}

// inverse scale of a knot: r = dist * knot_scale(bw, cal)
__device__ __forceinline__ float knot_scale(float bw, float cal) { return 1.0f / (bw * cal); }

template <int BASIS>
__device__ __forceinline__ float phi_eval(float x, float y, float cx, float cy, float sc) {
  float dx = x - cx, dy = y - cy;
  float d = __builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy));
  return basis_eval<BASIS>(d * sc);
}

// psi = exp(-0.5 ((t-c)/bw)^2)   (st_interp.py:590-594)
__device__ __forceinline__ float psi_eval(float t, float c, float bw) {
  float s = (t - c) / bw;
  return expf(-0.5f * s * s);
}

// floor(u) clamped to [0, n-1]; NaN and -inf -> 0, +inf -> n-1  (integer bookkeeping contract)
__device__ __forceinline__ int floor_clamp(float u, int n) {
  float f = floorf(u);
  return (f >= 0.f) ? ((f < (float)n) ? (int)f : n - 1) : 0;
}

// first knot of the WIN-wide window along one axis: clamp(floor(x*(side-1)) - 2, 0, max(side-WIN,0))
__device__ __forceinline__ int window_start(float x, int side, int win) {
  int c = floor_clamp(x * (float)(side - 1), side);
  int hi = side - win;
  hi = hi > 0 ? hi : 0;
  int s = c - 2;
  return s < 0 ? 0 : (s > hi ? hi : s);
}

}  // namespace stdadk
