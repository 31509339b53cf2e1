"""RMSE / MAE / R^2 on host arrays (reference stnf/utils/metrics.py:9-81)."""
from typing import Dict, Union

import numpy as np
import torch


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def _basic(yt, yp):
    ok = ~(np.isnan(yt) | np.isnan(yp))
    yt, yp = yt[ok], yp[ok]
    err = yt - yp
    mse = np.mean(err ** 2)
    return yt, mse, np.mean(np.abs(err)), np.sum(err ** 2)


def compute_metrics(y_true: Union[np.ndarray, torch.Tensor], y_pred: Union[np.ndarray, torch.Tensor],
                    per_horizon: bool = False) -> Dict[str, float]:
    y_true, y_pred = _np(y_true), _np(y_pred)
    yt, mse, mae, ss_res = _basic(y_true.flatten(), y_pred.flatten())
    ss_tot = np.sum((yt - np.mean(yt)) ** 2)
    out = {'rmse': float(np.sqrt(mse)), 'mae': float(mae), 'r2': float(1 - ss_res / (ss_tot + 1e-8)),
           'mse': float(mse)}
    if per_horizon and y_true.ndim == 4:
        rm, ma = [], []
        for h in range(y_true.shape[1]):
            _, mse_h, mae_h, _ = _basic(y_true[:, h].flatten(), y_pred[:, h].flatten())
            rm.append(float(np.sqrt(mse_h)))
            ma.append(float(mae_h))
        out['rmse_per_horizon'], out['mae_per_horizon'] = rm, ma
    return out


def compute_spatial_metrics(y_true, y_pred, coords, n_bins: int = 5) -> Dict[str, list]:
    """RMSE / MAE per ring of distance from the origin (reference stnf/utils/metrics.py:67-146): sites
    (axis 2 of the (B,H,S,1) arrays) are grouped into `n_bins` equal-width rings of |coords|, empty rings
    are skipped, NaNs ignored.  Returns bin_centers / rmse_by_distance / mae_by_distance."""
    yt, yp = _np(y_true), _np(y_pred)
    dist = np.hypot(np.asarray(coords)[:, 0], np.asarray(coords)[:, 1])
    edges = np.linspace(0, dist.max(), n_bins + 1)
    out = {'bin_centers': [], 'rmse_by_distance': [], 'mae_by_distance': []}
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = (dist >= lo) & (dist < hi)
        if not sel.any():
            continue
        a, b = yt[:, :, sel, :].ravel(), yp[:, :, sel, :].ravel()
        ok = ~(np.isnan(a) | np.isnan(b))
        err = a[ok] - b[ok]
        out['rmse_by_distance'].append(float(np.sqrt(np.mean(err ** 2))) if err.size else float('nan'))
        out['mae_by_distance'].append(float(np.mean(np.abs(err))) if err.size else float('nan'))
        out['bin_centers'].append(float((lo + hi) / 2))
    return out


def print_metrics(metrics: Dict[str, float], prefix: str = ""):
    """Human-readable dump of compute_metrics' output (reference :149-164)."""
    print(f"{prefix} Metrics:")
    for label, key in (("RMSE:", 'rmse'), ("MAE: ", 'mae'), ("R²:  ", 'r2')):
        print(f"  {label} {metrics[key]:.6f}")
    if 'rmse_per_horizon' in metrics:
        print(f"  RMSE per horizon: {metrics['rmse_per_horizon']}")
