"""Per-wave timing of the first GEMM phase of the 64-row tail forward (diagnostic build, -DSTDADK_DIAG):
usage: STDADK_EXTRA_FLAGS=-DSTDADK_DIAG bash st-dadk_amd/csrc/build.sh && python tools/diag/wave_stamps.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "st-dadk_amd")):
    sys.path.insert(0, p)
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
sf = torch.zeros((B // 16) * 16, dtype=torch.int64, device=dev)
os.environ["STDADK_TAIL_STAMPS"] = str(sf.data_ptr())
from stnf.models import STInterpMLP
from stnf.engine import TrainStep
torch.manual_seed(0)
m = STInterpMLP(k_spatial_centers=[1024, 4096, 5184], dropout=0.1).to(dev).train()
eng = TrainStep(m, ema_decay=0.999, max_batch=B)
g = torch.Generator().manual_seed(1)
coords = torch.rand(B, 2, generator=g).to(dev); t = torch.rand(B, generator=g).to(dev); y = torch.randn(B, 1, generator=g).to(dev)
for _ in range(4):
    eng._enqueue(None, coords, t, y, B, B)
torch.cuda.synchronize()
nt = B // 64
w = sf[nt * 16:nt * 16 + nt * 48].view(nt, 16, 3).cpu().double() / 100.0       # us
start = w[:, :, 0].min(1, keepdim=True).values
print("per wave (median over tiles, us since the tile's first wave entered the GEMM): start, end of MFMA loop, stores done")
for wv in range(16):
    print(f"  wave {wv:2d}: {(w[:, wv, 0] - start[:, 0]).median():6.2f} {(w[:, wv, 1] - start[:, 0]).median():6.2f} {(w[:, wv, 2] - start[:, 0]).median():6.2f}")
dur = w[:, :, 1] - w[:, :, 0]
print(f"MFMA loop per wave: median {dur.median():.2f} us, min {dur.min(1).values.median():.2f}, max {dur.max(1).values.median():.2f}; "
      f"last wave's end - first wave's start: {(w[:, :, 2].max(1).values - start[:, 0]).median():.2f} us")
