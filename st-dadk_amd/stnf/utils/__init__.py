"""Host-side helpers of the training driver: seeding, error metrics, the EMA shadow.  Same public names
as the reference package (its stnf/utils/__init__.py:4-14); nothing here touches the device library."""
from . import ema as _ema
from . import metrics as _metrics
from . import seed as _seed

set_seed = _seed.set_seed
compute_metrics = _metrics.compute_metrics
compute_spatial_metrics = _metrics.compute_spatial_metrics
print_metrics = _metrics.print_metrics
ModelEMA = _ema.ModelEMA

__all__ = sorted(n for n in dir() if not n.startswith('_'))
