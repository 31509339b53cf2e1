"""Where a short timed region (the driver's --steps 20 --warmup 5) loses time against a long one: the GPU-side end
time of every step (HIP events on the main stream, relative to an event recorded at the host's t0) next to the host's
enqueue times.  usage: python tools/startup_latency.py [steps] [warmup]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "st-dadk_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from stnf.models import STInterpMLP
from stnf.engine import TrainStep

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["c2"]
B = 4096
torch.manual_seed(0)
model = STInterpMLP(p=0, k_spatial_centers=wl["k_spatial_centers"], k_temporal_centers=wl["k_temporal_centers"],
                    hidden_dims=wl["hidden_dims"], dropout=0.1, layernorm=True).to(dev)
model.train()
n_obs = wl["n_obs"]
coords, t, y = bench.synth(n_obs, 2025, dev)
bpe = n_obs // B
eng = TrainStep(model, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=1.0 - 1.0 / (10.0 * bpe), max_batch=B)
perm = torch.randperm(n_obs, device=dev)


PIPE = os.environ.get("PIPE", "1") == "1"


def step(i):
    j, jn = i % bpe, (i + 1) % bpe
    eng.step_indexed(coords, t, y, perm[j * B:j * B + B], global_rows=B,
                     next_idx=perm[jn * B:jn * B + B] if PIPE else None)


PRE = os.environ.get("PRE", "none")
PRE_MS = float(os.environ.get("PRE_MS", "150"))
PRE_PIPE = os.environ.get("PRE_PIPE", "1") == "1"
if PRE != "none":
    import copy
    from stnf import _native as N
    from stnf.engine import Predictor
    tw = time.perf_counter()
    if PRE == "rbf":
        feats = torch.empty(4096, (model.input_dim + 31) // 32 * 32, device=dev)
        c4, t4 = coords[:4096].contiguous(), t[:4096].contiguous().view(-1)
        while (time.perf_counter() - tw) * 1e3 < PRE_MS:
            for _ in range(50):
                N.rbf_build(c4, t4, None, model.spatial_basis.centers, model.spatial_basis._bandwidths, "wendland",
                            model.temporal_basis.centers, model.temporal_basis.bandwidths, feats)
            torch.cuda.synchronize()
    elif PRE == "fwd":
        m2 = copy.deepcopy(model).eval()
        pr = Predictor(m2)
        ci, ti = coords.contiguous(), t.contiguous()
        while (time.perf_counter() - tw) * 1e3 < PRE_MS:
            for _ in range(10):
                pr.predict(ci, ti)
            torch.cuda.synchronize()
    elif PRE == "train":
        m2 = copy.deepcopy(model).train()
        e2 = TrainStep(m2, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999, max_batch=B)
        while (time.perf_counter() - tw) * 1e3 < PRE_MS:
            for i in range(50):
                j = i % bpe
                jn = (i + 1) % bpe
                e2.step_indexed(coords, t, y, perm[j * B:j * B + B], global_rows=B,
                                next_idx=perm[jn * B:jn * B + B] if PRE_PIPE else None)
            torch.cuda.synchronize()
    elif PRE == "self":
        while (time.perf_counter() - tw) * 1e3 < PRE_MS:
            for i in range(50):
                step(i)
            torch.cuda.synchronize()
    if "SLEEP_MS" in os.environ:
        time.sleep(float(os.environ["SLEEP_MS"]) * 1e-3)
    print(f"pre-warm {PRE}: {(time.perf_counter() - tw) * 1e3:.0f} ms")
for rep in range(int(os.environ.get("REPS", "3"))):
    for i in range(W):
        step(i)
    torch.cuda.synchronize()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(K):
        step(W + i)
        ev[i + 1].record()
        host.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) * 1e6
    gpu = [ev[0].elapsed_time(e) * 1e3 for e in ev[1:]]
    print(f"rep {rep}: host wall {el:.1f} us for {K} steps = {el / K:.1f} us/step; last step's GPU end {gpu[-1]:.1f} us after the "
          f"first event; host wall - that = {el - gpu[-1]:.1f} us")
    if K <= 40:
        print("  step: host-enqueued-at / gpu-done-at / gpu step time (us)")
        prev = 0.0
        for i in range(K):
            print(f"  {i:3d}  {host[i]:8.1f}  {gpu[i]:8.1f}  {gpu[i] - prev:7.1f}")
            prev = gpu[i]
    else:
        print("  mean gpu step time per block of 10 steps (us): " +
              " ".join(f"{(gpu[i + 9] - (gpu[i - 1] if i else 0.0)) / 10:.1f}" for i in range(0, K - 9, 10)))
