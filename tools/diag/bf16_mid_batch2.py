"""bench.py's sequence around its bf16 sweep, step by step, to find what makes the 16 384-row bf16 line slow there."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "st-dadk_amd")):
    sys.path.insert(0, p)
import torch
import bench
from stnf.models import STInterpMLP
from stnf.engine import TrainStep
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["c2"]
n_obs = 100_000
coords, t, y = bench.synth(n_obs, 2025, dev)
perm = torch.randperm(n_obs, device=dev)

def timed(b2, k2, eng_kw=None, graph=False, tag=""):
    torch.manual_seed(0)
    m2 = STInterpMLP(p=0, k_spatial_centers=wl["k_spatial_centers"], k_temporal_centers=wl["k_temporal_centers"],
                     hidden_dims=wl["hidden_dims"], dropout=0.1, layernorm=True).to(dev).train()
    e2 = TrainStep(m2, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999, max_batch=b2, use_graph=graph, **(eng_kw or {}))
    nb2 = max(n_obs // b2, 1)
    sl = lambda i: perm[(i % nb2) * b2:(i % nb2) * b2 + b2]
    for i in range(5):
        e2.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(5, 5 + k2):
        e2.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / k2
    print(f"{tag} B={b2} {eng_kw} graph={graph}: {dt * 1e3:.3f} ms/step = {b2 / dt / 1e6:.1f} M obs/s", flush=True)
    del e2, m2

order = sys.argv[1] if len(sys.argv) > 1 else "abcd"
for ch in order:
    if ch == "a":
        timed(4096, 200, tag="headline-like")
    if ch == "b":
        timed(16384, 40, tag="sweep"); timed(65536, 40, tag="sweep")
    if ch == "c":
        timed(4096, 100, graph=True, tag="graph")
    if ch == "d":
        for b2 in (4096, 16384, 65536):
            timed(b2, 40 if b2 > 4096 else 100, dict(dtype="bf16"), tag="bf16")
