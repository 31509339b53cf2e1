"""STNF models (same exports as the reference's stnf/models/__init__.py:4-6)."""
from .st_interp import STInterpMLP, create_model

__all__ = ['STInterpMLP', 'create_model']
