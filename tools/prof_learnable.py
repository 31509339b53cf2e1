import os, sys
sys.path.insert(0, "/root/repo/st-dadk_amd"); sys.path.insert(0, "/root/repo")
import torch
from stnf.models import STInterpMLP
from stnf.engine import TrainStep
from stnf import _native as N
B = 4096
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = STInterpMLP(k_spatial_centers=[1024, 4096, 5184], dropout=0.1, spatial_learnable=True, gradient_damping=True,
                damping_threshold=0.0, damping_strength=5.0).to(dev)
m.train()
eng = TrainStep(m, ema_decay=0.999, max_batch=B, domain_penalty_weight=0.01)
g = torch.Generator().manual_seed(1)
coords = torch.rand(B, 2, generator=g).to(dev); t = torch.rand(B, generator=g).to(dev); y = torch.randn(B, 1, generator=g).to(dev)
for _ in range(3):
    eng._enqueue(None, coords, t, y, B, B)
torch.cuda.synchronize()
N.profile_enable(True)
for _ in range(10):
    eng._enqueue(None, coords, t, y, B, B)
recs = N.profile_collect()
N.profile_enable(False)
agg = {}
for n, ms in recs:
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += ms
print({k: round(v[1] / v[0] * 1e3, 1) for k, v in agg.items()})
print("sum", round(sum(v[1] for v in agg.values()) / 10 * 1e3, 1))
