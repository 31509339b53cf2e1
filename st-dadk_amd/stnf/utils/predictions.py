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


@torch.no_grad()
def evaluate_model(model, dataset, config=None, predictor=None):
    """The driver's `evaluate_model` (scripts/train_st_interp.py:884-961) on a `DeviceDataset` (or any object with
    `.coords (N,2)`, `.t (N,1)`, `.y (N,1)`): predictions of the whole set under eval(), then on the host, in the
    reference's float32 numpy arithmetic: 'mse', 'mae', 'rmse' (of the median column for 'multi-quantile'), plus
    'check_loss' for `regression_type == 'quantile'` with `current_quantile`, and 'crps', 'mean_check_loss' and the
    'check_loss' alias for 'multi-quantile'.  The reference walks a DataLoader in batches of min(max(16 B, 32768), N)
    (:2292-2293); the predictions do not depend on the batching, so the set goes through `Predictor.predict` in
    its own chunks (device models) or one forward (host models).  `predictor`: reuse one across calls."""
    from ..losses import check_loss_numpy, compute_crps_multi_quantile      # host arithmetic (the predictions are small)
    dev = next(model.parameters()).device
    was_training = model.training
    model.eval()
    try:
        if dev.type == "cuda":
            if predictor is None:
                from ..engine import Predictor
                predictor = Predictor(model)
            preds_t = predictor.predict(dataset.coords, dataset.t)
        else:
            n = dataset.coords.shape[0]
            preds_t = model(torch.zeros(n, 0), dataset.coords, dataset.t.view(-1, 1))
    finally:
        model.train(was_training)
    pred = preds_t.float().cpu().numpy()                        # (N, Q) float32, like the reference's stacked batches
    obs = dataset.y.float().cpu().numpy().reshape(-1, 1)
    kind = (config or {}).get("regression_type", "mean")
    levels = list((config or {}).get("quantile_levels", [0.1, 0.5, 0.9])) if kind == "multi-quantile" else None
    # point metrics: the single output, or the median level's column of a multi-quantile head
    point = pred[:, len(levels) // 2:len(levels) // 2 + 1] if levels is not None else pred
    err = point - obs
    sq = float(np.mean(err * err))
    out = {"mse": sq, "mae": float(np.mean(np.abs(err))), "rmse": float(np.sqrt(np.mean(err * err)))}
    if kind == "quantile" and "current_quantile" in config:
        out["check_loss"] = float(check_loss_numpy(pred, obs, config["current_quantile"]))
    if levels is not None:
        per_level = [float(check_loss_numpy(pred[:, j:j + 1], obs, tau)) for j, tau in enumerate(levels)]
        out["crps"] = float(compute_crps_multi_quantile(pred, obs, levels))
        out["mean_check_loss"] = out["check_loss"] = float(np.mean(per_level))      # the reference keeps both names
    return out
