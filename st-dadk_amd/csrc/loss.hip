// N3: batch objectives (stdadk_loss_f32) and the delta-reparameterised head
// (stdadk_delta_head_f32 / stdadk_delta_head_backward_f32).  See include/stdadk.h.
#include "loss.h"

namespace stdadk {

int make_loss(const stdadk_loss_desc *l, int Q, LossDev *out) {
  LossDev L;
  L.kind = STDADK_LOSS_MSE; L.y_cols = Q; L.nc_w = 0.f; L.nc_pow = 1;
  for (int k = 0; k < STDADK_MAX_Q; ++k) L.tau[k] = 0.5f;
  if (l) {
    STDADK_REQUIRE(l->kind == STDADK_LOSS_MSE || l->kind == STDADK_LOSS_PINBALL, STDADK_E_ARG,
                   "loss: unknown kind %d", l->kind);
    STDADK_REQUIRE(Q >= 1 && Q <= STDADK_MAX_Q, STDADK_E_ARG, "loss: Q=%d outside 1..%d", Q,
                   STDADK_MAX_Q);
    STDADK_REQUIRE(l->y_cols == Q || l->y_cols == 1, STDADK_E_ARG, "loss: y_cols=%d must be 1 or Q=%d",
                   l->y_cols, Q);
    L.kind = l->kind; L.y_cols = l->y_cols;
    if (l->kind == STDADK_LOSS_PINBALL) {
      for (int k = 0; k < Q; ++k) {
        STDADK_REQUIRE(l->tau[k] > 0.f && l->tau[k] < 1.f, STDADK_E_ARG, "loss: tau[%d]=%g outside (0,1)", k,
                       (double)l->tau[k]);
        L.tau[k] = l->tau[k];
      }
      STDADK_REQUIRE(l->nc_weight >= 0.f, STDADK_E_ARG, "loss: negative nc_weight");
      STDADK_REQUIRE(l->nc_weight == 0.f || l->nc_power == 1 || l->nc_power == 2, STDADK_E_ARG,
                     "Unsupported power=%d; use 1 or 2.", l->nc_power);
      L.nc_w = Q > 1 ? l->nc_weight : 0.f;
      L.nc_pow = l->nc_power == 2 ? 2 : 1;
    }
  }
  *out = L;
  return 0;
}

// one thread per (row, q) element
__global__ void loss_rows_kernel(LossDev L, const float *__restrict__ yp, const float *__restrict__ y,
                                 int64_t n, int Q, float scale, float *__restrict__ dY,
                                 float *__restrict__ loss_sum) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / Q;
    const int q = (int)(i - r * Q);
    const float yq = yp[i];
    const float yup = q + 1 < Q ? yp[i + 1] : 0.f;
    const float ydn = q > 0 ? yp[i - 1] : 0.f;
    const float yt = L.y_cols == 1 ? y[r] : y[i];
    float dy;
    acc += loss_elem(L, Q, q, loss_tau(L, q), yq, yup, ydn, yt, scale, dy);
    if (dY) dY[i] = dy;
  }
  if (loss_sum) {
    __shared__ float red[4];
    float s = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
  }
}

int launch_loss(const LossDev &L, const float *yp, const float *y, int64_t B, int Q, float scale, float *dY,
                float *loss_sum, hipStream_t st) {
  const int64_t n = B * Q;
  if (n == 0) return 0;
  int64_t blocks = ceil_div(n, 256);
  if (blocks > 1024) blocks = 1024;
  STDADK_LAUNCH(loss_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, L, yp, y, n, Q, scale, dY,
                loss_sum);
  STDADK_CHECK_LAUNCH("loss");
  return 0;
}

// beta_k = sum_{l<=k} delta_l, split into the output layer's rows and biases; one thread per column
__global__ void delta_head_kernel(const float *__restrict__ delta, int64_t ldd, int Q, int d,
                                  float *__restrict__ Wo, float *__restrict__ bo) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > d) return;
  float beta = 0.f;
  for (int k = 0; k < Q; ++k) {
    beta += delta[k * ldd + j];
    if (j == 0) bo[k] = beta; else Wo[(int64_t)k * d + j - 1] = beta;
  }
}

// one workgroup: S_k by block reduction, then the reverse cumulative sums per column
__global__ __launch_bounds__(256) void delta_head_bwd_kernel(const float *__restrict__ delta,
                                                            const float *__restrict__ dWo,
                                                            const float *__restrict__ dbo, int64_t ldd,
                                                            int Q, int d, float lam_g, float lam_l,
                                                            float *__restrict__ d_delta,
                                                            float *__restrict__ loss_sum) {
  __shared__ float red[4];
  __shared__ float wS[STDADK_MAX_Q];     // share of max(delta_k0, S_k) that flows into S_k
  __shared__ float Psum;
  const int tid = threadIdx.x;
  if (tid == 0) Psum = 0.f;
  if (tid < STDADK_MAX_Q) wS[tid] = 0.f;
  __syncthreads();
  for (int k = 1; k < Q; ++k) {
    float s = 0.f;
    for (int j = 1 + tid; j <= d; j += 256) s += fmaxf(-delta[k * ldd + j], 0.f);
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
      const float S = red[0] + red[1] + red[2] + red[3];
      const float d0 = delta[k * ldd];
      wS[k] = S > d0 ? 1.0f : (S == d0 ? 0.5f : 0.f);
      Psum += d0 - fmaxf(d0, S);
    }
    __syncthreads();
  }
  for (int j = tid; j <= d; j += 256) {
    float acc = 0.f;
    for (int k = Q - 1; k >= 0; --k) {
      acc += j == 0 ? dbo[k] : dWo[(int64_t)k * d + j - 1];
      float gp = 0.f;
      if (k >= 1) gp = j == 0 ? wS[k] : (delta[k * ldd + j] <= 0.f ? wS[k] : 0.f);
      d_delta[k * ldd + j] = fmaf(lam_g, gp, acc);
    }
  }
  if (tid == 0 && loss_sum && lam_l != 0.f) atomicAdd(loss_sum, lam_l * Psum);
}

}  // namespace stdadk

using namespace stdadk;

extern "C" int stdadk_loss_f32(const stdadk_loss_desc *loss, const float *y_pred, const float *y, int64_t B,
                               int32_t Q, float grad_scale, float *dY, float *loss_sum,
                               stdadk_stream_t stream) {
  STDADK_REQUIRE(B >= 0 && Q >= 1, STDADK_E_ARG, "loss: bad sizes B=%lld Q=%d", (long long)B, Q);
  if (B == 0) return 0;
  STDADK_REQUIRE(y_pred && y, STDADK_E_ARG, "loss: NULL pointer");
  LossDev L;
  int rc = make_loss(loss, Q, &L);
  if (rc) return rc;
  return launch_loss(L, y_pred, y, B, Q, grad_scale, dY, loss_sum, (hipStream_t)stream);
}

extern "C" int stdadk_delta_head_f32(const float *delta, int64_t ldd, int32_t Q, int32_t d, float *Wo,
                                     float *bo, stdadk_stream_t stream) {
  STDADK_REQUIRE(Q >= 1 && Q <= STDADK_MAX_Q && d >= 1 && ldd >= d + 1, STDADK_E_ARG,
                 "delta_head: bad sizes Q=%d d=%d ldd=%lld", Q, d, (long long)ldd);
  STDADK_REQUIRE(delta && Wo && bo, STDADK_E_ARG, "delta_head: NULL pointer");
  STDADK_LAUNCH(delta_head_kernel, dim3((unsigned)ceil_div(d + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                delta, ldd, Q, d, Wo, bo);
  STDADK_CHECK_LAUNCH("delta_head");
  return 0;
}

extern "C" int stdadk_delta_head_backward_f32(const float *delta, const float *dWo, const float *dbo,
                                              int64_t ldd, int32_t Q, int32_t d, float lambda_grad,
                                              float lambda_loss, float *d_delta, float *loss_sum,
                                              stdadk_stream_t stream) {
  STDADK_REQUIRE(Q >= 1 && Q <= STDADK_MAX_Q && d >= 1 && ldd >= d + 1, STDADK_E_ARG,
                 "delta_head_backward: bad sizes Q=%d d=%d ldd=%lld", Q, d, (long long)ldd);
  STDADK_REQUIRE(delta && dWo && dbo && d_delta, STDADK_E_ARG, "delta_head_backward: NULL pointer");
  STDADK_LAUNCH(delta_head_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, delta, dWo, dbo, ldd, Q, d,
                lambda_grad, lambda_loss, d_delta, loss_sum);
  STDADK_CHECK_LAUNCH("delta_head_backward");
  return 0;
}
