// internal: the optimiser launch that also bins the next batch (optim.hip: adamw_bin_kernel)
#pragma once
#include "common.h"

namespace stdadk {
struct BinSmallArgs;
int adamw_ema_with_binning(float *p, const float *g, float *m, float *v, float *ema, int64_t n, float lr,
                           const float *lr_dev, float beta1, float beta2, float eps, float weight_decay,
                           const int32_t *step_dev, float max_norm, const float *sumsq_parts, int32_t n_parts,
                           float ema_decay, const stdadk_bf16_shadow *shadow, const float *loss_watch,
                           int32_t *nonfinite_step, stdadk_stream_t stream, const BinSmallArgs &bin);
}  // namespace stdadk
