"""End-to-end soak: KAUST-shaped synthetic field (S sites x T times, z = smooth field + N(0, 0.1^2) noise, 10 % held
out), C2 model, the reference's optimiser settings with a cosine learning-rate schedule, `run_epoch` over a
device-resident dataset for many epochs.  Prints train loss / held-out RMSE per few epochs and the rate.
usage (MI355X box): python tools/soak_training.py [epochs]"""
import math, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-dadk_amd"))
from stnf.models import STInterpMLP
from stnf.engine import TrainStep, Predictor
from stnf.dataio.device_dataset import DeviceDataset
from stnf.utils import compute_metrics, set_seed

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
set_seed(0)
rs = np.random.RandomState(0)
S, T = 2000, 100
coords = rs.uniform(0, 1, (S, 2)).astype(np.float32)
tt = np.arange(T) / (T - 1)
field = (np.sin(4 * np.pi * coords[None, :, 0]) * np.cos(3 * np.pi * coords[None, :, 1])
         * (1 + 0.5 * np.sin(2 * np.pi * tt[:, None])))
z = (field + 0.1 * rs.standard_normal((T, S))).astype(np.float32)
mask = rs.uniform(size=(T, S)) < 0.9                       # 90 % train, 10 % held out
ds = DeviceDataset.from_mask(z, coords, mask)
dv = DeviceDataset.from_mask(z, coords, ~mask)
clean = DeviceDataset.from_mask(field.astype(np.float32), coords, ~mask)
d = torch.device("cuda:0")
m = STInterpMLP(p=0, k_spatial_centers=[1024, 4096, 5184], k_temporal_centers=[10, 15, 45], hidden_dims=[256, 256, 128],
                dropout=0.1, layernorm=True).to(d)
m.train()
B = 4096
nb = math.ceil(len(ds) / B)
eng = TrainStep(m, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=1.0 - 1.0 / (10.0 * nb), max_batch=B)
g = torch.Generator(device=d).manual_seed(0)
t0 = time.perf_counter()
for ep in range(epochs):
    eng.set_lr(2e-2 * 0.5 * (1 + math.cos(math.pi * ep / epochs)))         # CosineAnnealingLR, as the driver
    tr = eng.run_epoch(ds, B, generator=g)
    if ep % 10 == 9 or ep == epochs - 1:
        eng.swap_in_ema()
        m.eval()
        pred = Predictor(m).predict(dv.coords, dv.t)
        m.train()
        eng.swap_in_ema()
        rmse_noisy = float(((pred - dv.y) ** 2).mean().sqrt())
        rmse_clean = float(((pred - clean.y) ** 2).mean().sqrt())
        assert math.isfinite(tr) and math.isfinite(rmse_noisy)
        print(f"epoch {ep + 1:3d}  train MSE {tr:.5f}  held-out RMSE vs noisy {rmse_noisy:.4f} (noise floor 0.1000)  "
              f"vs the noise-free field {rmse_clean:.4f}", flush=True)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"{epochs} epochs x {nb} steps of {B} in {el:.2f} s incl. evaluation = {epochs * len(ds) / el / 1e6:.1f} M obs/s end to end")


# ---- the shipped configuration's shape: 227 GMM-initialised learnable knots, 5 quantiles, non-crossing penalty
taus = [0.05, 0.25, 0.5, 0.75, 0.95]
np.random.seed(0)
m2 = STInterpMLP(p=0, k_spatial_centers=[25, 81, 121], k_temporal_centers=[10, 15, 45], hidden_dims=[256, 256, 128],
                 dropout=0.1, layernorm=True, spatial_learnable=True, spatial_init_method="gmm", train_coords=coords,
                 gradient_damping=True, damping_threshold=0.0, damping_strength=5.0, output_dim=5).to(d)
m2.train()
e2 = TrainStep(m2, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=1.0 - 1.0 / (10.0 * nb), max_batch=B,
               loss="pinball", quantile_levels=taus, non_crossing_weight=0.5, domain_penalty_weight=0.01)
c0 = m2.spatial_basis.centers.detach().clone()
t0 = time.perf_counter()
for ep in range(epochs):
    e2.set_lr(2e-2 * 0.5 * (1 + math.cos(math.pi * ep / epochs)))
    e2.set_basis_lr(0.05 * 2e-2 * 0.5 * (1 + math.cos(math.pi * ep / epochs)))
    tr = e2.run_epoch(ds, B, generator=g)
torch.cuda.synchronize()
el = time.perf_counter() - t0
e2.swap_in_ema()
m2.eval()
with torch.no_grad():
    pq = m2(None, dv.coords, dv.t.view(-1, 1))
cover = [(float((dv.y[:, 0] <= pq[:, i]).float().mean())) for i in range(5)]
cross = float((pq[:, 1:] < pq[:, :-1]).float().mean())
moved = float((m2.spatial_basis.centers.detach() - c0).norm(dim=1).mean())
print(f"multi-quantile + learnable knots: objective {tr:.5f}; held-out coverage P(y <= q_tau) for tau {taus}: "
      f"{[round(c, 3) for c in cover]}; crossing fraction {cross:.4f}; mean knot displacement {moved:.4f}; "
      f"{epochs * len(ds) / el / 1e6:.1f} M obs/s")
assert all(math.isfinite(c) for c in cover) and math.isfinite(tr)
