"""Host utilities the driver imports (reference stnf/utils/__init__.py:4-6)."""
from .seed import set_seed
from .metrics import compute_metrics
from .ema import ModelEMA

__all__ = ['set_seed', 'compute_metrics', 'ModelEMA']
