"""rbf_build GB/s against rows per workgroup (STDADK_RBF_ROWS) at footprints inside and past the Infinity Cache.
usage (MI355X box): python tools/bench_rbf_rows.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "st-dadk_amd")); sys.path.insert(0, %r)
import bench
from stnf.models import STInterpMLP
from stnf import _native as N
dev = torch.device("cuda:0")
for wlname, B in (("c2", 4096), ("c2", 16384), ("c2", 65536), ("c4", 4096), ("c4", 16384)):
    wl = bench.WORKLOADS[wlname]
    m = STInterpMLP(p=0, k_spatial_centers=wl["k_spatial_centers"], k_temporal_centers=wl["k_temporal_centers"],
                    hidden_dims=wl["hidden_dims"]).to(dev)
    D = m.input_dim
    c, t, _ = bench.synth(B, 1, dev)
    feats = torch.empty(B, (D + 31) // 32 * 32, device=dev)
    dt = bench.time_events(lambda: N.rbf_build(c, t.view(-1), None, m.spatial_basis.centers, m.spatial_basis._bandwidths,
                                               "wendland", m.temporal_basis.centers, m.temporal_basis.bandwidths, feats), 20)
    print(f"rows={os.environ.get('STDADK_RBF_ROWS', 'auto'):>4s} {wlname} B={B:6d} {B * (12 + 4 * D) / 1e6:8.1f} MB  {dt * 1e6:8.1f} us  {B * (12 + 4 * D) / dt / 1e12:5.2f} TB/s", flush=True)
    del feats
''' % (ROOT, ROOT)
for rows in ("auto", "1", "2", "4", "8", "16", "32", "64", "128"):
    env = dict(os.environ)
    if rows != "auto":
        env["STDADK_RBF_ROWS"] = rows
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
