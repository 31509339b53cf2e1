"""Quantile objectives of the training driver on device tensors (scripts/train_st_interp.py:37-248).

The reference defines these in its driver script with torch / numpy ops; here the batch-sized ones run
through libstdadk (stdadk_loss_f32, stdadk_delta_head_backward_f32) and return device scalars without
a host sync.  They are evaluation helpers: training through `TrainStep` fuses the same arithmetic
into the step, and the reference's own torch versions keep working on this package's `forward()`."""
import numpy as np
import torch

from . import _native as N


def _sum(desc, y_pred, y):
    y_pred = y_pred.detach().contiguous().float()
    y = y.detach().contiguous().float().view(y_pred.shape[0], -1)
    acc = torch.zeros(1, device=y_pred.device)
    N.loss(desc, y_pred, y, 0.0, None, acc)
    return acc[0]


def quantile_loss(y_pred, y_true, quantile):
    """mean(max((q-1) e, q e)), e = y_true - y_pred, for (B,1) predictions (reference :37-50)."""
    y_pred = y_pred.view(y_pred.shape[0], -1)
    Q = y_pred.shape[1]
    desc = N.make_loss("pinball", Q, 1 if y_true.numel() == y_pred.shape[0] else Q, [float(quantile)] * Q)
    return _sum(desc, y_pred, y_true) / y_pred.numel()


def multi_quantile_loss(y_pred, y_true, quantile_levels, non_crossing_weight=0.0, non_crossing_power=1):
    """The multi-quantile batch objective (reference :625-658): mean over levels of the per-level
    check loss, plus non_crossing_weight * non_crossing_penalty(y_pred, "mean", power)."""
    Q = y_pred.shape[1]
    desc = N.make_loss("pinball", Q, 1 if y_true.numel() == y_pred.shape[0] else Q, quantile_levels,
                       non_crossing_weight, non_crossing_power)
    return _sum(desc, y_pred, y_true) / y_pred.numel()


def non_crossing_penalty(y_pred_multi_q, reduction="mean", power=1):
    """sum_k relu(q_k - q_{k+1})^power per row, mean / sum over rows (reference :53-88)."""
    if y_pred_multi_q.dim() != 2 or y_pred_multi_q.shape[1] < 2:
        return torch.tensor(0.0, device=y_pred_multi_q.device)
    if power not in (1, 2):
        raise ValueError(f"Unsupported power={power}; use 1 or 2.")
    if reduction not in ("mean", "sum"):
        raise ValueError(f"Unsupported reduction='{reduction}'; use 'mean' or 'sum'.")
    B, Q = y_pred_multi_q.shape
    # the penalty alone = objective(nc_weight = 1) - objective(nc_weight = 0) on any targets
    zero = torch.zeros(B, 1, device=y_pred_multi_q.device)
    taus = [0.5] * Q
    with_nc = _sum(N.make_loss("pinball", Q, 1, taus, 1.0, power), y_pred_multi_q, zero)
    without = _sum(N.make_loss("pinball", Q, 1, taus, 0.0, power), y_pred_multi_q, zero)
    per_batch = (with_nc - without) / Q
    return per_batch / B if reduction == "mean" else per_batch


def compute_p_nc_delta_penalty(delta_params):
    """P_nc(delta) = sum_{k>=2} (delta_k0 - max(delta_k0, sum_j max(0, -delta_kj))) (reference :91-160)."""
    if delta_params is None or len(delta_params) < 2:
        dev = delta_params[0].device if delta_params else torch.device("cpu")
        return torch.tensor(0.0, device=dev)
    delta = torch.stack([p.detach().float() for p in delta_params])
    Q, d1 = delta.shape
    z = torch.zeros(Q, d1 - 1, device=delta.device)
    acc = torch.zeros(1, device=delta.device)
    N.delta_head_backward(delta, z, torch.zeros(Q, device=delta.device), 0.0, 1.0, torch.empty_like(delta), acc)
    return acc[0]


def check_loss_numpy(y_pred, y_true, quantile):
    """Mean check loss on host arrays (reference :163-175)."""
    e = np.asarray(y_true) - np.asarray(y_pred)
    return np.mean(np.maximum((quantile - 1) * e, quantile * e))


def compute_crps(predictions_dict, y_true, weights=None):
    """Equation 4.6: CRPS = 2 * sum_k w_k * check loss at tau_k; uniform w_k = 1/K by default, given
    weights are normalised (reference :178-222)."""
    quantiles = sorted(predictions_dict.keys())
    K = len(quantiles)
    if K == 0:
        raise ValueError("predictions_dict cannot be empty")
    if K == 1:
        return 2.0 * check_loss_numpy(predictions_dict[quantiles[0]], y_true, quantiles[0])
    if weights is None:
        w = np.ones(K) / K
    else:
        w = np.asarray(weights)
        if len(w) != K:
            raise ValueError(f"weights length ({len(w)}) must match number of quantiles ({K})")
        w = w / w.sum()
    return 2.0 * sum(w[i] * check_loss_numpy(predictions_dict[q], y_true, q) for i, q in enumerate(quantiles))


def compute_crps_multi_quantile(preds, y_true, quantile_levels, weights=None):
    """CRPS of an (N,Q) prediction array (reference :225-248).  Device tensors with uniform weights
    are reduced on the device (CRPS = 2 x the mean check loss over all (row, level) pairs)."""
    if isinstance(preds, torch.Tensor) and preds.is_cuda and weights is None:
        y = y_true if isinstance(y_true, torch.Tensor) else torch.as_tensor(np.asarray(y_true), device=preds.device)
        return float(2.0 * multi_quantile_loss(preds, y.to(preds.device).float().view(-1, 1), quantile_levels))
    preds = preds.detach().cpu().numpy() if isinstance(preds, torch.Tensor) else np.asarray(preds)
    y_true = y_true.detach().cpu().numpy() if isinstance(y_true, torch.Tensor) else np.asarray(y_true)
    if y_true.ndim > 1:
        y_true = y_true.flatten()
    return compute_crps({q: preds[:, i] for i, q in enumerate(quantile_levels)}, y_true, weights=weights)
