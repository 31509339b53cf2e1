import sys, time
sys.path.insert(0, "/root/repo/st-dadk_amd")
import torch
from stnf.models import STInterpMLP
from stnf.engine import Predictor
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = STInterpMLP(k_spatial_centers=[1024, 4096, 5184], dropout=0.1).to(dev).eval()
n = 65536 * 8
coords = torch.rand(n, 2, device=dev); t = torch.rand(n, device=dev)
for chunk in (65536, 262144):
    for graph in (True, False):
        pr = Predictor(m, chunk=chunk, use_graph=graph)
        for _ in range(2): pr.predict(coords, t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): pr.predict(coords, t)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"chunk {chunk} graph {graph}: {n / dt / 1e6:.1f} M obs/s")
