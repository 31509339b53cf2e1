"""Wall time per step against the sum of the step's kernel times (event brackets) for the split-call step paths
(learnable knots: forward/backward, norms, optimiser as separate C calls from Python)."""
import os, sys, time
import numpy as np
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
n_obs = 100000
coords, t, y = bench.synth(n_obs, 1, dev)
t = t.view(-1).contiguous()
perm = torch.randperm(n_obs, device=dev)
taus = [0.05, 0.25, 0.5, 0.75, 0.95]
np.random.seed(0)
site = coords[:20000].cpu().numpy()
CASES = {
    "c2_fixed": (dict(), dict()),
    "c2_learnable": (dict(spatial_learnable=True, gradient_damping=True, damping_threshold=0.0, damping_strength=5.0),
                     dict(domain_penalty_weight=0.01)),
    "shipped_yaml_227": (dict(k_spatial_centers=[25, 81, 121], spatial_learnable=True, spatial_init_method="gmm",
                              train_coords=site, gradient_damping=True, damping_threshold=0.0, damping_strength=5.0,
                              output_dim=5),
                         dict(loss="pinball", quantile_levels=taus, non_crossing_weight=0.5, domain_penalty_weight=0.01)),
    "ref_default_227": (dict(k_spatial_centers=[25, 81, 121]), dict()),
}
for name, (mk, ek) in CASES.items():
    torch.manual_seed(0)
    kw = dict(p=0, k_spatial_centers=wl["k_spatial_centers"], k_temporal_centers=wl["k_temporal_centers"],
              hidden_dims=wl["hidden_dims"], dropout=0.1, layernorm=True)
    kw.update(mk)
    m = STInterpMLP(**kw).to(dev); m.train()
    yy = y if kw.get("output_dim", 1) == 1 else y
    eng = TrainStep(m, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999, max_batch=B, **ek)
    nb = n_obs // B
    sl = lambda i: perm[(i % nb) * B:(i % nb) * B + B]
    for i in range(20):
        eng.step_indexed(coords, t, yy, sl(i), next_idx=sl(i + 1))
    torch.cuda.synchronize()
    k = 200
    t0 = time.perf_counter()
    for i in range(20, 20 + k):
        eng.step_indexed(coords, t, yy, sl(i), next_idx=sl(i + 1))
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    N.profile_enable(True)
    for i in range(10):
        eng.step_indexed(coords, t, yy, sl(i), next_idx=sl(i + 1))
    torch.cuda.synchronize()
    recs = N.profile_collect(); N.profile_enable(False)
    agg = {}
    for nm, ms in recs:
        a = agg.setdefault(nm.split("(")[0].replace("stdadk::", "")[:40], 0.0); agg[nm.split("(")[0].replace("stdadk::", "")[:40]] = a + ms
    ksum = sum(agg.values()) / 10 * 1e3
    side = sum(v for kk, v in agg.items() if kk.startswith(("bin_", "gather_batch", "cell_"))) / 10 * 1e3
    print(f"{name:18s} wall {wall / k * 1e6:7.1f} us/step | host enqueue {t_enq / k * 1e6:7.1f} us/step | kernels {ksum:7.1f} us "
          f"(of which batch preparation on the side stream {side:5.1f}) | whole_step={getattr(eng, '_whole_step', None)}", flush=True)
    print("      ", {kk: round(v / 10 * 1e3, 1) for kk, v in sorted(agg.items(), key=lambda kv: -kv[1])})
    del eng, m
