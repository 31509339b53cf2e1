"""Train step of the C2 model with one or two knots per wave in the per-knot gather of dW0^T
(STDADK_KNOTS_PER_WAVE=1|2), pair groups in table order or XCD-striped
(STDADK_KNOT_XCD=0|1), several batch sizes.  usage (MI355X box): python tools/bench_knot_pairs.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-dadk_amd"))
from stnf.models import STInterpMLP
from stnf.engine import TrainStep
from stnf import _native as N

d = torch.device("cuda:0")
torch.manual_seed(0)
N_OBS = 262144
coords = torch.rand(N_OBS, 2, device=d)
t = torch.randint(0, 100, (N_OBS, 1), device=d).float() / 99
y = torch.randn(N_OBS, 1, device=d)
mk = dict(p=0, k_spatial_centers=[1024, 4096, 5184], k_temporal_centers=[10, 15, 45], hidden_dims=[256, 256, 128],
          dropout=0.1, layernorm=True)
for pairs, xcd in (("1", "1"), ("2", "0"), ("2", "1")):
    os.environ["STDADK_KNOTS_PER_WAVE"] = pairs
    os.environ["STDADK_KNOT_XCD"] = xcd
    line = [f"knots_per_wave={pairs} xcd_striped={xcd}"]
    for B in (4096, 8192, 16384, 65536):
        m = STInterpMLP(**mk).to(d)
        m.train()
        eng = TrainStep(m, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999, max_batch=B)
        perm = torch.randperm(N_OBS, device=d)
        nb = N_OBS // B
        sl = lambda i: perm[(i % nb) * B:(i % nb) * B + B]
        for i in range(5):
            eng.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1))
        torch.cuda.synchronize()
        k = 200 if B <= 8192 else 40
        t0 = time.perf_counter()
        for i in range(5, 5 + k):
            eng.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        # device time of the weight-gradient launch
        N.profile_enable(True)
        for i in range(20):
            eng.step_indexed(coords, t, y, sl(i))
        torch.cuda.synchronize()
        recs = N.profile_collect(); N.profile_enable(False)
        dw = [ms for nm, ms in recs if "dw_all" in nm or "l1_window_bwd" in nm]
        line.append(f"B={B}: {dt * 1e6:.1f}us {B / dt / 1e6:.1f}M/s dw={sum(dw) / max(len(dw), 1) * 1e3:.1f}us")
        del eng, m
    print("  ".join(line), flush=True)
