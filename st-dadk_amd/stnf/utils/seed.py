"""One seed for every generator the driver uses: Python's `random`, numpy's global RandomState and
torch (host generator plus every HIP device).  Counterpart of the reference's stnf/utils/seed.py:9-27;
the cuDNN determinism switches it sets have no meaning on ROCm and are left out."""
import random

import numpy as np
import torch


def set_seed(seed: int):
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    print(f"[INFO] Seed set to {seed}")
