"""Event timing of the AdamW+EMA kernel on a C2-sized flat buffer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "st-dadk_amd")):
    sys.path.insert(0, p)
import torch
from stnf import _native as N
n = 2756100
dev = torch.device("cuda:0")
p, g, m, v, e = (torch.randn(n, device=dev) for _ in range(5))
v.abs_()
parts = torch.zeros(256, device=dev)
def run():
    N.sumsq(g, parts)
    N.adamw_ema(p, g, m, v, e, 1e-3, (0.9, 0.999), 1e-8, 5e-4, 3, max_norm=10.0, sumsq_parts=parts, ema_decay=0.999)
for _ in range(5): run()
torch.cuda.synchronize()
N.profile_enable(True)
for _ in range(50): run()
recs = N.profile_collect(); N.profile_enable(False)
agg = {}
for nm, ms in recs:
    a = agg.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += ms
for k, (c, t) in agg.items():
    us = t / c * 1e3
    print(k, round(us, 2), "us", round((36 if "adamw" in k else 4) * n / us / 1e6, 2), "TB/s")
