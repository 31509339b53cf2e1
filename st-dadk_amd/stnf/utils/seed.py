"""Seed fixing (reference stnf/utils/seed.py:9-27)."""
import random

import numpy as np
import torch


def set_seed(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)          # also seeds every HIP device generator
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    print(f"[INFO] Seed set to {seed}")
