"""Dense-grid predictions of a trained model and their `predictions.npz` record.

The reference's driver evaluates the model at every site for every time index, one `model(...)` call per time
slice (scripts/train_st_interp.py:1228-1248: `t = t_idx / (T - 1)` or 0 when T == 1, no covariates, the MEDIAN
quantile column of a multi-quantile model), and stores the (T, S) array next to the observed field and the three
masks (scripts/train_st_interp.py:2551-2560).  Here the same grid is one `Predictor.predict_grid` call on the
device (layer 0 once per site and once per time instead of once per (site, time)); on host tensors the model's own
forward runs time slice by time slice like the reference.
"""
import os

import numpy as np
import torch

NPZ_KEYS = ("predictions", "true", "coords", "train_mask", "valid_mask", "test_mask")


@torch.no_grad()
def predict_all_times(model, coords, T, max_rows=None):
    """(T, S) float64 array of predictions at the S sites `coords` (ndarray or tensor, (S, 2)) for the time indices
    0 .. T-1 (normalised t = t_idx / (T - 1), 0.0 when T == 1); multi-output models contribute their median column
    output_dim // 2 (scripts/train_st_interp.py:1240-1245).  Runs under eval() and restores the mode."""
    if getattr(model, "p", 0) != 0:
        raise ValueError("predict_all_times: the dense grid has no covariates (p must be 0)")
    dev = next(model.parameters()).device
    c = torch.as_tensor(np.asarray(coords) if not torch.is_tensor(coords) else coords).float().to(dev)
    if c.dim() != 2 or c.shape[1] != 2:
        raise ValueError(f"predict_all_times: coords must be (S, 2), got {tuple(c.shape)}")
    T = int(T)
    S = c.shape[0]
    tv = torch.arange(T, dtype=torch.float32) / (T - 1) if T > 1 else torch.zeros(1, dtype=torch.float32)
    was_training = model.training
    model.eval()
    try:
        if dev.type == "cuda":
            from ..engine import Predictor
            out = Predictor(model).predict_grid(c, tv.to(dev), max_rows=max_rows)        # (T, S, Q)
        else:
            out = torch.stack([model(torch.zeros(S, 0), c, torch.full((S, 1), float(tv[i]))) for i in range(T)])
    finally:
        model.train(was_training)
    q = out.shape[2]
    col = out[:, :, q // 2] if q > 1 else out[:, :, 0]
    return col.double().cpu().numpy()


def save_predictions_npz(output_dir, predictions, true, coords, train_mask, valid_mask, test_mask):
    """Writes `<output_dir>/predictions.npz` with the reference's keys (scripts/train_st_interp.py:2551-2560) and
    returns its path.  Shapes are checked: predictions / true / masks (T, S), coords (S, 2)."""
    predictions = np.asarray(predictions)
    T, S = predictions.shape
    arrs = dict(predictions=predictions, true=np.asarray(true), coords=np.asarray(coords),
                train_mask=np.asarray(train_mask), valid_mask=np.asarray(valid_mask), test_mask=np.asarray(test_mask))
    for k in ("true", "train_mask", "valid_mask", "test_mask"):
        if arrs[k].shape != (T, S):
            raise ValueError(f"save_predictions_npz: {k} has shape {arrs[k].shape}, predictions {(T, S)}")
    if arrs["coords"].shape != (S, 2):
        raise ValueError(f"save_predictions_npz: coords has shape {arrs['coords'].shape}, expected {(S, 2)}")
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(str(output_dir), "predictions.npz")
    np.savez(path, **arrs)
    return path
