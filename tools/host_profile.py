import os, sys, time, cProfile, pstats

import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "st-dadk_amd")); sys.path.insert(0, ROOT)
from stnf.models import STInterpMLP
from stnf.engine import TrainStep
import bench
wl = bench.WORKLOADS["c2"]
B = 4096; n_obs = 100000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
coords = torch.rand(n_obs, 2, generator=g).to(dev); t = torch.rand(n_obs, generator=g).to(dev); y = torch.rand(n_obs, 1, generator=g).to(dev)
perm = torch.randperm(n_obs).to(dev)
which = sys.argv[1] if len(sys.argv) > 1 else "fixed"
mk = dict(p=0, k_spatial_centers=wl["k_spatial_centers"], k_temporal_centers=wl["k_temporal_centers"], hidden_dims=wl["hidden_dims"], dropout=0.1, layernorm=True)
ek = {}
if which == "learn":
    mk.update(spatial_learnable=True, gradient_damping=True, damping_threshold=0.0, damping_strength=5.0); ek = dict(domain_penalty_weight=0.01)
m = STInterpMLP(**mk).to(dev); m.train()
eng = TrainStep(m, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999, max_batch=B, **ek)
nb = n_obs // B
sl = lambda i: perm[(i % nb) * B:(i % nb) * B + B]
for i in range(20): eng.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1))
k = 2000
t0 = time.perf_counter()
for i in range(20, 20 + k): eng.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1))
torch.cuda.synchronize(); print(which, "wall us/step", (time.perf_counter() - t0) / k * 1e6)
pr = cProfile.Profile(); pr.enable()
for i in range(20, 20 + k): eng.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1))
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
