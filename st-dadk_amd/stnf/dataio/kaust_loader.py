"""KAUST CSV (x, y, t, z) -> dense (T, S) field.  Host-side, runs once per experiment."""
from typing import Dict, Tuple

import numpy as np
import pandas as pd


def load_kaust_csv_single(data_path: str, normalize: bool = True) -> Tuple[np.ndarray, np.ndarray, Dict]:
    """Same contract as the reference's loader (stnf/dataio/kaust_loader.py:19-76):

    z_data (T,S) float32 with NaN where a (t, site) pair is absent, T = max(t) (t is 1-based);
    coords (S,2) float32 in first-appearance order of the unique (x,y) pairs;
    metadata {'z_mean','z_std'} when normalize (population std over the present values).
    Vectorised (factorize + scatter) instead of the reference's per-row Python loop; a duplicated
    (t, site) row keeps the LAST value, as the reference's sequential assignment does."""
    df = pd.read_csv(data_path)
    print(f"[INFO] Loaded data: {len(df)} rows")
    xy = df[['x', 'y']]
    site_idx, uniques = pd.factorize(pd.MultiIndex.from_frame(xy), sort=False)
    S = len(uniques)
    print(f"[INFO] Total sites: {S}")
    first = np.full(S, -1, dtype=np.int64)
    # first-appearance row of every site (factorize numbers sites in order of appearance)
    rows = np.arange(len(df))
    first[site_idx[::-1]] = rows[::-1]
    coords = xy.values[first].astype(np.float32)
    # purely spatial files (KAUST 1a: columns x, y, z) have no time column; the reference reads df['t']
    # (kaust_loader.py:54) and needs an adapter file with t = 1 for them -- here they load as one time slice
    t_vals = df['t'].values if 't' in df.columns else np.ones(len(df), dtype=np.int64)
    T = int(t_vals.max())
    print(f"[INFO] Time range: 1 ~ {T}")
    z_data = np.full((T, S), np.nan, dtype=np.float32)
    z_data[t_vals.astype(np.int64) - 1, site_idx] = df['z'].values.astype(np.float32)
    metadata = {}
    if normalize:
        z_flat = z_data[~np.isnan(z_data)]
        z_mean, z_std = z_flat.mean(), z_flat.std()
        z_data = (z_data - z_mean) / z_std
        metadata['z_mean'], metadata['z_std'] = z_mean, z_std
        print(f"[INFO] Normalized z: mean={z_mean:.4f}, std={z_std:.4f}")
    return z_data, coords, metadata
