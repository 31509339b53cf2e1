"""Why is the bf16 step slow at 16 384 rows in bench.py's sweep?  Times step_indexed on 100 000 resident rows at several
batch sizes with the batch preparation pipelined or not, fp32 and bf16."""
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
for dtype in ("f32", "bf16"):
    for b2 in (8192, 16384, 32768, 65536):
        for pipe in (True, False):
            torch.manual_seed(0)
            m = STInterpMLP(p=0, k_spatial_centers=wl["k_spatial_centers"], k_temporal_centers=wl["k_temporal_centers"],
                            hidden_dims=wl["hidden_dims"], dropout=0.1, layernorm=True).to(dev).train()
            e = TrainStep(m, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999, max_batch=b2, dtype=dtype)
            nb = max(n_obs // b2, 1)
            sl = lambda i: perm[(i % nb) * b2:(i % nb) * b2 + b2]
            for i in range(5):
                e.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1) if pipe else None)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(5, 35):
                e.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1) if pipe else None)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 30
            print(f"{dtype} B={b2} pipelined={pipe}: {dt * 1e3:.3f} ms/step = {b2 / dt / 1e6:.1f} M obs/s", flush=True)
            del e, m
