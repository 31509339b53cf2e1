"""In-kernel phase timing of the fused tail kernels (diagnostic build: -DSTDADK_DIAG).
usage on the GPU box:  STDADK_EXTRA_FLAGS=-DSTDADK_DIAG bash st-dadk_amd/csrc/build.sh && python tools/stamp_tail.py [B]
Prints the median over workgroups of each stamp-to-stamp interval (wall_clock64 ticks at 100 MHz -> us)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "st-dadk_amd")):
    sys.path.insert(0, p)
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
nblk = (B + 15) // 16
sf = torch.zeros(nblk * 16, dtype=torch.int64, device=dev)
sb = torch.zeros(nblk * 16, dtype=torch.int64, device=dev)
os.environ["STDADK_TAIL_STAMPS"] = str(sf.data_ptr())
os.environ["STDADK_TAIL_BWD_STAMPS"] = str(sb.data_ptr())
from stnf.models import STInterpMLP
from stnf.engine import TrainStep
torch.manual_seed(0)
m = STInterpMLP(k_spatial_centers=[1024, 4096, 5184], dropout=0.1).to(dev)
m.train()
eng = TrainStep(m, ema_decay=0.999, max_batch=B)
g = torch.Generator().manual_seed(1)
coords = torch.rand(B, 2, generator=g).to(dev); t = torch.rand(B, generator=g).to(dev); y = torch.randn(B, 1, generator=g).to(dev)
for _ in range(5):
    eng._enqueue(None, coords, t, y, B, B)
torch.cuda.synchronize()
for name, s, labels in (("tail_fwd", sf, ["input", "gemm1", "z->lds+bar", "ln1", "bar", "gemm2", "z->lds+bar", "ln2", "bar", "", "", "", "", "head"]),
                        ("tail_bwd", sb, ["head", "ln_bwd L2", "gemm 128->256", "store+bar", "ln_bwd L1", "gemm 256->256", "store+bar", "ln_bwd L0"])):
    st = s.view(-1, 16).cpu().double()
    st = st[(st[:, 0] > 0)]
    cols = [i for i in range(16) if (st[:, i] > 0).all()]
    print(name, "stamped slots", cols, "workgroups", st.shape[0])
    prev = cols[0]
    li = 0
    for c in cols[1:]:
        d = (st[:, c] - st[:, prev]) / 100.0
        lab = labels[li] if li < len(labels) else ""
        print(f"  {prev:2d}->{c:2d} {lab:16s} median {d.median().item():6.2f} us  max {d.max().item():6.2f}")
        prev = c; li += 1
    tot = (st[:, cols[-1]] - st[:, cols[0]]) / 100.0
    print(f"  total in-kernel median {tot.median().item():.2f} us")

# fused step kernel (B <= 4096): phase boundaries in slots 10..13 of the forward stamp buffer
st = sf.view(-1, 16).cpu().double()
st = st[(st[:, 10] > 0)]
if st.shape[0]:
    for a, b, lab in ((10, 11, "layer-0 window phase"), (11, 12, "sync + tail forward"), (12, 13, "sync + tail backward")):
        d = (st[:, b] - st[:, a]) / 100.0
        print(f"  fused {lab:24s} median {d.median().item():6.2f} us  max {d.max().item():6.2f}")
    tot = (st[:, 13] - st[:, 10]) / 100.0
    print(f"  fused total in-kernel median {tot.median().item():.2f} us  max {tot.max().item():.2f}")
    print(f"  first workgroup start -> last workgroup end: {(st[:, 13].max() - st[:, 10].min()).item() / 100.0:.2f} us")
