"""Model classes of the interpolation path: `STInterpMLP` and the config-dict factory `create_model`
(the two names the reference's stnf.models exports), implemented over libstdadk in st_interp.py."""
from . import st_interp as _impl

STInterpMLP = _impl.STInterpMLP
create_model = _impl.create_model
__all__ = ('STInterpMLP', 'create_model')
