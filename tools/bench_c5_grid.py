"""BASELINE config C5: 10 M-point dense prediction grid (100 000 sites x 100 times), forward only, C2 model.
usage (MI355X box): python tools/bench_c5_grid.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-dadk_amd"))
from stnf.models import STInterpMLP
from stnf.engine import Predictor

d = torch.device("cuda:0")
torch.manual_seed(0)
S, T = 100000, 100
coords = torch.rand(S, 2, device=d)
tv = torch.arange(T, device=d, dtype=torch.float32) / (T - 1)
m = STInterpMLP(p=0, k_spatial_centers=[1024, 4096, 5184], k_temporal_centers=[10, 15, 45], hidden_dims=[256, 256, 128],
                dropout=0.1, layernorm=True).to(d)
m.eval()
pr = Predictor(m)
for name, fn in (("predict_grid (per-site half of layer 0 once per site)", lambda: pr.predict_grid(coords, tv)),
                 ("predict (row by row, 262 144-row chunks)",
                  lambda: pr.predict(coords.repeat(T, 1), tv.repeat_interleave(S)))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {dt * 1e3:.1f} ms per 10 M points = {S * T / dt / 1e6:.0f} M points/s", flush=True)
a = pr.predict_grid(coords[:5000], tv[:7])
b = pr.predict(coords[:5000].repeat(7, 1), tv[:7].repeat_interleave(5000)).view(7, 5000, 1)
print("max abs difference on a 5000 x 7 corner:", float((a - b).abs().max()))
