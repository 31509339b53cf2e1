"""Per-kernel event times (us per step, stdadk_profile_*) of the fused train step at a given workload / batch / dtype.
usage (MI355X box): python tools/prof_step.py [--workload c2|c4] [--batch 4096,16384,65536] [--dtype f32|bf16]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-dadk_amd"))
sys.path.insert(0, ROOT)
from stnf.models import STInterpMLP
from stnf.engine import TrainStep
from stnf import _native as N
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c2")
ap.add_argument("--batch", default="4096,16384,65536")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--dropout", type=float, default=0.1)
args = ap.parse_args()
dev = torch.device("cuda:0")
wl = bench.WORKLOADS[args.workload]
for B in [int(b) for b in args.batch.split(",")]:
    coords, t, y = bench.synth(B, 1, dev)
    t = t.view(-1).contiguous()
    torch.manual_seed(0)
    m = STInterpMLP(p=0, k_spatial_centers=wl["k_spatial_centers"], k_temporal_centers=wl["k_temporal_centers"],
                    hidden_dims=wl["hidden_dims"], dropout=args.dropout, layernorm=True).to(dev)
    m.train()
    kw = {} if args.dtype == "f32" else dict(dtype=args.dtype)
    eng = TrainStep(m, ema_decay=0.999, max_batch=B, **kw)
    for _ in range(3):
        eng._enqueue(None, coords, t, y, B, B)
    torch.cuda.synchronize()
    n = 10
    N.profile_enable(True)
    for _ in range(n):
        eng._enqueue(None, coords, t, y, B, B)
    recs = N.profile_collect()
    N.profile_enable(False)
    agg = {}
    for nm, ms in recs:
        a = agg.setdefault(nm.split("(")[0].replace("stdadk::", "")[:60], [0, 0.0]); a[0] += 1; a[1] += ms
    tot = sum(v[1] for v in agg.values()) / n * 1e3
    print(f"{args.workload} B={B} {args.dtype}: kernel sum {tot:.1f} us/step = {B / tot:.1f} M obs/s (kernels only)")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:60s} x{v[0] // n}  {v[1] / n * 1e3:8.1f} us", flush=True)
    del eng, m
