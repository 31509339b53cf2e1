"""Per-kernel event times of the fused train step with learnable knots (DA-STDK), config C2, B = 4096."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-dadk_amd")); sys.path.insert(0, ROOT)
from stnf.models import STInterpMLP
from stnf.engine import TrainStep
from stnf import _native as N
import bench
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["c2"]
B = 4096
coords, t, y = bench.synth(B, 1, dev)
t = t.view(-1).contiguous()
torch.manual_seed(0)
m = STInterpMLP(p=0, k_spatial_centers=wl["k_spatial_centers"], k_temporal_centers=wl["k_temporal_centers"],
                hidden_dims=wl["hidden_dims"], dropout=0.1, layernorm=True, spatial_learnable=True,
                gradient_damping=True, damping_threshold=0.0, damping_strength=5.0).to(dev)
m.train()
eng = TrainStep(m, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999, max_batch=B, domain_penalty_weight=0.01)
for phase in range(3):
    for _ in range(30):
        eng.step(None, coords, t, y)
    torch.cuda.synchronize()
    n = 10
    N.profile_enable(True)
    for _ in range(n):
        eng.step(None, coords, t, y)
    torch.cuda.synchronize()
    recs = N.profile_collect()
    N.profile_enable(False)
    agg = {}
    for nm, ms in recs:
        a = agg.setdefault(nm.split("(")[0].replace("stdadk::", "")[:60], [0, 0.0]); a[0] += 1; a[1] += ms
    tot = sum(v[1] for v in agg.values()) / n * 1e3
    print(f"learnable c2 B={B} after {30 * (phase + 1) + 10 * phase} steps: kernel sum {tot:.1f} us/step")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:60s} x{v[0] / n:.1f}  {v[1] / n * 1e3:8.1f} us", flush=True)
