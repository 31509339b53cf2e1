"""stnf — MI355X (gfx950) build of ST-DADK's spatio-temporal interpolation hot path.

Same import surface as the reference package for the path that is built here
(`stnf.models`, `stnf.dataio.kaust_loader.load_kaust_csv_single`, `stnf.utils`); the arithmetic
runs in libstdadk.so (hand-written HIP).  Put `st-dadk_amd/` on PYTHONPATH and the reference's
`scripts/train_st_interp.py` imports this package instead of its own.
"""
__version__ = "0.1.0"
