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
