"""Per-kernel event times of the first steps of a fresh engine against the same engine a few hundred steps later
(the first ~100 steps of a run measure ~6 % slower; which launch is it?).
usage: python tools/early_vs_late.py [gap steps, default 400]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "st-dadk_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from stnf.models import STInterpMLP
from stnf.engine import TrainStep
from stnf import _native as N

GAP = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["c2"]
B = 4096
torch.manual_seed(0)
model = STInterpMLP(p=0, k_spatial_centers=wl["k_spatial_centers"], k_temporal_centers=wl["k_temporal_centers"],
                    hidden_dims=wl["hidden_dims"], dropout=0.1, layernorm=True).to(dev)
model.train()
n_obs = wl["n_obs"]
coords, t, y = bench.synth(n_obs, 2025, dev)
bpe = n_obs // B
eng = TrainStep(model, lr=float(os.environ.get("LR", "2e-2")), weight_decay=5e-4, grad_clip=10.0,
                ema_decay=1.0 - 1.0 / (10.0 * bpe), max_batch=B)
perm = torch.randperm(n_obs, device=dev)
k = [0]


def steps(n):
    for _ in range(n):
        j, jn = k[0] % bpe, (k[0] + 1) % bpe
        eng.step_indexed(coords, t, y, perm[j * B:j * B + B], global_rows=B, next_idx=perm[jn * B:jn * B + B])
        k[0] += 1


def profiled(n, label):
    torch.cuda.synchronize()
    N.profile_enable(True)
    steps(n)
    torch.cuda.synchronize()
    recs = N.profile_collect()
    N.profile_enable(False)
    agg = {}
    for name, us in recs:   # milliseconds -> us below
        a = agg.setdefault(name.split("<")[0], [0, 0.0])
        a[0] += 1; a[1] += us * 1e3
    print(label, "loss", round(eng.mean_loss(), 4),
          {kk: round(v[1] / n, 1) for kk, v in sorted(agg.items(), key=lambda kv: -kv[1][1])})


steps(5)
profiled(20, f"steps 5-25     ")
steps(GAP)
profiled(20, f"steps {k[0]}-{k[0] + 20}")
steps(GAP)
profiled(20, f"steps {k[0]}-{k[0] + 20}")
