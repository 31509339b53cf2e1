"""Data I/O.  Only the loader the interpolation driver uses is provided
(reference scripts/train_st_interp.py:2187 -> stnf/dataio/kaust_loader.py:19-76); the reference's
sliding-window forecasting API is not on the hot path."""
from .kaust_loader import load_kaust_csv_single
from .device_dataset import DeviceDataset

__all__ = ['load_kaust_csv_single', 'DeviceDataset']
