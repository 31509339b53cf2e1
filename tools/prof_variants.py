"""Per-kernel event times (us per step) of the step variants that are not the bench headline: the reference's
shipped YAML shape (227 GMM knots, learnable, 5 quantiles), 227 uniform knots MSE, C2 learnable knots.
usage (MI355X box): python tools/prof_variants.py"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-dadk_amd"))
from stnf.models import STInterpMLP
from stnf.engine import TrainStep
from stnf import _native as N

B = 4096
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
coords = torch.rand(B, 2, generator=g).to(dev)
t = torch.rand(B, generator=g).to(dev)
y = torch.randn(B, 1, generator=g).to(dev)
taus = [0.05, 0.25, 0.5, 0.75, 0.95]
np.random.seed(0)
site = coords.cpu().numpy()
variants = {
    "shipped_yaml": (dict(k_spatial_centers=[25, 81, 121], spatial_learnable=True, spatial_init_method="gmm",
                          train_coords=site, gradient_damping=True, damping_threshold=0.0, damping_strength=5.0,
                          output_dim=5),
                     dict(loss="pinball", quantile_levels=taus, non_crossing_weight=0.5, domain_penalty_weight=0.01)),
    "ref_default_227": (dict(k_spatial_centers=[25, 81, 121]), {}),
    "c2_learnable": (dict(k_spatial_centers=[1024, 4096, 5184], spatial_learnable=True, gradient_damping=True,
                          damping_threshold=0.0, damping_strength=5.0), dict(domain_penalty_weight=0.01)),
}
for name, (mk, ek) in variants.items():
    torch.manual_seed(0)
    m = STInterpMLP(dropout=0.1, **mk).to(dev)
    m.train()
    eng = TrainStep(m, ema_decay=0.999, max_batch=B, **ek)
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
    print(name, "window" if eng.uses_window else "materialised",
          {k: (v[0] // 10, round(v[1] / 10 * 1e3, 1)) for k, v in agg.items()},
          "sum", round(sum(v[1] for v in agg.values()) / 10 * 1e3, 1), flush=True)
