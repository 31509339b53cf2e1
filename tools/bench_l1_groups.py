"""Layer-0 window forward with 1, 2 or 4 observations per wave (STDADK_L1_GROUP): forward-only calls and
train steps of the C2 model at several batch sizes.  usage (MI355X box): python tools/bench_l1_groups.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-dadk_amd"))
from stnf.models import STInterpMLP
from stnf.engine import TrainStep, Predictor

d = torch.device("cuda:0")
torch.manual_seed(0)
N_OBS = 262144
coords = torch.rand(N_OBS, 2, device=d)
t = torch.randint(0, 100, (N_OBS, 1), device=d).float() / 99
y = torch.randn(N_OBS, 1, device=d)
mk = dict(p=0, k_spatial_centers=[1024, 4096, 5184], k_temporal_centers=[10, 15, 45], hidden_dims=[256, 256, 128],
          dropout=0.1, layernorm=True)
for grp in ("1", "2", ""):
    if grp:
        os.environ["STDADK_L1_GROUP"] = grp
    else:
        os.environ.pop("STDADK_L1_GROUP", None)
    line = [f"group={grp or 'auto'}"]
    m = STInterpMLP(**mk).to(d)
    m.eval()
    pr = Predictor(m)
    for n in (16384, 65536, 262144):
        c, tt = coords[:n].contiguous(), t[:n].contiguous()
        for _ in range(3):
            pr.predict(c, tt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            pr.predict(c, tt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        line.append(f"infer{n}: {dt * 1e6:.0f}us {n / dt / 1e6:.0f}M/s")
    m.train()
    for B in (8192, 16384, 65536):
        eng = TrainStep(m, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999, max_batch=B)
        perm = torch.randperm(N_OBS, device=d)
        nb = N_OBS // B
        sl = lambda i: perm[(i % nb) * B:(i % nb) * B + B]
        for i in range(5):
            eng.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5, 45):
            eng.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
        line.append(f"train{B}: {dt * 1e6:.0f}us {B / dt / 1e6:.1f}M/s")
        del eng
        m = STInterpMLP(**mk).to(d)
        m.train()
    print("  ".join(line), flush=True)
