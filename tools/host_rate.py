"""Host-side enqueue time per step vs device time (is the eager launch chain host-bound?)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "st-dadk_amd"))
import torch
from stnf.models import STInterpMLP
from stnf.engine import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = STInterpMLP(p=0, k_spatial_centers=[1024, 4096, 5184], k_temporal_centers=[10, 15, 45], hidden_dims=[256, 256, 128],
                dropout=0.1, layernorm=True).to(dev)
m.train()
n, B = 100000, 4096
coords = torch.rand(n, 2, device=dev); t = torch.rand(n, device=dev); y = torch.randn(n, 1, device=dev)
perm = torch.randperm(n, device=dev)
for pipe in (False, True):
    eng = TrainStep(m, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999, max_batch=B)
    def sl(i): return perm[(i % 24) * B:(i % 24) * B + B]
    for i in range(20): eng.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1) if pipe else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20, 220): eng.step_indexed(coords, t, y, sl(i), next_idx=sl(i + 1) if pipe else None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"pipelined={pipe}: host enqueue {1e6 * (t1 - t0) / 200:.1f} us/step, total {1e6 * (t2 - t0) / 200:.1f} us/step")
