"""Device-resident observation set.

The reference turns an observation mask into a Python list of per-sample dicts
(`create_dataset_from_mask`, scripts/train_st_interp.py:413-450), stacks them per batch
(`collate_fn`, :453-460) and copies four small tensors to the device every step (:609-612).
Here the whole set is built once, vectorised, and lives in HBM; a training step takes an index
slice (`stnf.engine.TrainStep.step_indexed`)."""
import numpy as np
import torch


class DeviceDataset:
    """coords (N,2), t (N,1), y (N,1), X (N,p) fp32 tensors on one device, in the reference's
    sample order (np.argwhere(mask): time-major, then site), NaN targets dropped,
    t = t_idx/(T-1) (0 when T == 1) exactly as train_st_interp.py:439."""

    def __init__(self, coords, t, y, X=None):
        self.coords, self.t, self.y, self.X = coords, t, y, X
        self.n = coords.shape[0]

    @classmethod
    def from_mask(cls, z_data, coords, mask, p_covariates=0, device="cuda"):
        z = np.asarray(z_data)
        T, S = z.shape
        ti, si = np.nonzero(np.asarray(mask, dtype=bool))          # same order as np.argwhere
        yv = z[ti, si]
        keep = ~np.isnan(yv)
        ti, si, yv = ti[keep], si[keep], yv[keep]
        t_norm = (ti / (T - 1) if T > 1 else np.zeros_like(ti, dtype=np.float64)).astype(np.float32)
        c = np.asarray(coords, dtype=np.float32)[si]
        dev = torch.device(device)
        X = torch.zeros(len(ti), p_covariates, device=dev) if p_covariates > 0 else None
        return cls(torch.from_numpy(np.ascontiguousarray(c)).to(dev),
                   torch.from_numpy(t_norm.reshape(-1, 1)).to(dev),
                   torch.from_numpy(yv.astype(np.float32).reshape(-1, 1)).to(dev), X)

    def __len__(self):
        return self.n

    def epoch_batches(self, batch_size, generator=None, shuffle=True):
        """Index slices of one epoch (last one ragged, like DataLoader without drop_last).  `batch_size` may be
        a list of sizes summing to len(self) (the data-parallel schedule of stnf.distributed.epoch_schedule)."""
        dev = self.coords.device
        idx = torch.randperm(self.n, device=dev, generator=generator) if shuffle else torch.arange(self.n, device=dev)
        if not isinstance(batch_size, int):
            batch_size = [int(b) for b in batch_size]
            if sum(batch_size) != self.n:
                raise ValueError(f"batch sizes sum to {sum(batch_size)}, the set has {self.n} rows")
        return idx.split(batch_size)
