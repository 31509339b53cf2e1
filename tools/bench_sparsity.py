"""Device time of stdadk_sparsity_f32 on the C2 first layer (10 374 x 256), both layouts.
usage (MI355X box): python tools/bench_sparsity.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "st-dadk_amd"))
from stnf import _native as N

d = torch.device("cuda:0")
Ks, Kt, H = 10304, 70, 256
for w0_t in (True, False):
    W = torch.randn((Ks + Kt, H) if w0_t else (H, Ks + Kt), device=d) * 0.05
    G = torch.zeros_like(W)
    loss = torch.zeros(1, device=d)
    sp = N.make_sparsity("sparse_group", 1e-4, 1e-3)
    for _ in range(5):
        N.sparsity(sp, W, G, w0_t, 0, Ks, Kt, loss_scale=1.0, loss_sum=loss)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        N.sparsity(sp, W, G, w0_t, 0, Ks, Kt, loss_scale=1.0, loss_sum=loss)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print(f"w0_t={w0_t}: {us:.1f} us per call, {3 * W.numel() * 4 / us / 1e3:.0f} GB/s of 3 x 10.6 MB")
