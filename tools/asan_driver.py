"""Host-side sanitizer pass of libstdadk (runs in the build container, no GPU): the library built by tools/build_asan.sh
(AddressSanitizer + UBSan on the HOST code), loaded with STDADK_DRY_RUN=1 so that every entry point runs its argument
validation, workspace planning and job-table construction and launches nothing.  The calls come from the package's own
host code (TrainStep / Predictor / the model class on host tensors standing in for device buffers), so the descriptors
are the ones real runs build.  Any sanitizer report aborts the process (halt_on_error / -fno-sanitize-recover).
Started by tests/test_host_sanitizer.py with the ASan runtime preloaded."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "st-dadk_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
assert os.environ.get("STDADK_DRY_RUN") == "1" and "asan" in os.environ.get("STDADK_LIB", "")
import numpy as np
import torch
from stnf import _native as N
from stnf.models import STInterpMLP
from stnf.engine import TrainStep, Predictor

calls = 0


def model(ks, hidden=(256, 256, 128), **kw):
    torch.manual_seed(0)
    return STInterpMLP(p=kw.pop("p", 0), k_spatial_centers=ks, k_temporal_centers=[10, 15, 45],
                       hidden_dims=list(hidden), dropout=kw.pop("dropout", 0.1), layernorm=kw.pop("layernorm", True), **kw)


def data(n, p=0, q=1):
    g = torch.Generator().manual_seed(1)
    return (torch.randn(n, p, generator=g) if p else None, torch.rand(n, 2, generator=g), torch.rand(n, 1, generator=g),
            torch.randn(n, q, generator=g))


def steps(m, B, eng_kw=None, q_cols=1, indexed=True):
    global calls
    X, c, t, y = data(max(B, 8), m.p, q_cols)
    m.train()
    eng = TrainStep(m, max_batch=B, ema_decay=0.99, **(eng_kw or {}))
    eng.step(X, c[:B], t[:B], y[:B])
    eng.step(X, c[:B - 3], t[:B - 3], y[:B - 3])          # ragged against every tile size
    if indexed and m.p == 0:
        idx = torch.arange(B - 1, -1, -1)
        eng.step_indexed(c, t, y, idx)
        if eng._whole_step and eng.uses_window:
            # the step whose optimiser launch also bins the next batch (stdadk_train_step_next_f32): planning of both
            eng._enqueue(None, c, t.view(-1), y, B, B, idx=idx, ws=eng.ws,
                         nxt=(idx[: max(B // 2, 1)].contiguous(), torch.empty_like(eng.ws)))
            calls += 1
    calls += 3
    if eng.ema is not None:
        eng.swap_in_ema(); eng.swap_in_ema()
    m.eval()
    if m.p == 0:
        pr = Predictor(m, chunk=1024)
        pr.predict(c, t)
        pr.predict_grid(c[:37], torch.linspace(0, 1, 5))
        calls += 2
    return eng


# ---- the configurations of BASELINE.json and of the reference's shipped YAML, every planner branch
for ks, B in (([25, 81, 121], 300), ([1024], 4096), ([1024, 4096, 5184], 4096), ([1024, 4096, 5184], 20000),
              ([1024, 4096, 16384, 28224], 16384)):
    steps(model(ks), B)
steps(model([1024, 4096, 5184]), 4096, dict(dtype="bf16"))
steps(model([1024, 4096, 5184]), 20000, dict(dtype="bf16"))
steps(model([1024, 4096, 5184]), 4096, dict(force_dense=True))
steps(model([25, 81, 121], layernorm=False, dropout=0.0), 777)
steps(model([25, 81, 121], p=3), 300)
steps(model([25, 81, 121], hidden=(128, 64)), 129)
steps(model([1024, 4096, 5184], spatial_learnable=True, gradient_damping=True), 4096,
      dict(domain_penalty_weight=0.01, movement_penalty_weight=0.02))
steps(model([25, 81, 121], spatial_learnable=True), 300, dict(domain_penalty_weight=0.01, world_size=2))
taus = [0.05, 0.25, 0.5, 0.75, 0.95]
steps(model([1024, 4096, 5184], output_dim=5), 4096, dict(loss="pinball", quantile_levels=taus, non_crossing_weight=0.5))
steps(model([25, 81, 121], output_dim=5, use_delta_reparameterization=True), 300,
      dict(loss="pinball", quantile_levels=taus, non_crossing_lambda=0.05))
steps(model([1024, 4096, 5184]), 4096, dict(sparsity_penalty_type="sparse_group"))
# sharded optimiser, two virtual ranks driven by hand (the collectives are the caller's in this mode)
m = model([1024, 4096, 5184], spatial_learnable=True).train()
eng = TrainStep(m, max_batch=4096, ema_decay=0.99, world_size=2, shard_optimizer=True, domain_penalty_weight=0.01)
X, c, t, y = data(4096)
for r in range(2):
    eng.set_virtual_rank(r)
    eng._enqueue_grads(None, c[:2048], t[:2048].view(-1), y[:2048], 2048, 4096)
    eng._shard_sumsq()
    eng._shard_adamw()
eng.swap_in_ema(); eng.swap_in_ema()
calls += 6
np.random.seed(0)
site = np.random.rand(20000, 2).astype(np.float32)
steps(model([1024, 4096], spatial_learnable=True, spatial_init_method="random_site", train_coords=site), 4096,
      dict(domain_penalty_weight=0.01))
steps(model([25, 81, 121], spatial_init_method="gmm", train_coords=site[:3000], spatial_learnable=True), 300)

# ---- autograd entry points of the model class (separate forward / backward calls)
m = model([1024, 4096, 5184], dropout=0.0).train()
_, c, t, y = data(300)
torch.nn.MSELoss()(m(None, c, t), y).backward()
m = model([25, 81, 121], dropout=0.0, spatial_basis_function="gaussian").train()
torch.nn.MSELoss()(m(None, c, t), y).backward()
calls += 4

# ---- argument errors: every one must come back as a negative code with a message, none may crash
lib = N.lib()
bad = 0
d = N.make_desc(297, [256, 256, 128], 1, True, 0.1)
for B in (-5, 0, 1, 7, 1 << 40):
    lib.stdadk_mlp_workspace_bytes(ctypes.byref(d), B)
for hidden in ([2000], [256] * 8, [16], [100, 36]):
    dd = N.make_desc(297, hidden, 1, True, 0.0)
    lib.stdadk_mlp_workspace_bytes(ctypes.byref(dd), 4096)
f32 = torch.zeros(4096)
for n in (-1, 0, 5):
    rc = lib.stdadk_sumsq_f32(f32.data_ptr(), n, None, None, None)
    bad += rc < 0
rc = lib.stdadk_adamw_ema_f32(f32.data_ptr(), f32.data_ptr(), None, None, None, 16, 0.1, None, 0.9, 0.999, 1e-8, 0.0, 1,
                              None, 0.0, None, 0, 1.0, 0.0, None, None, None, None)
bad += rc < 0
rc = lib.stdadk_adamw_ema_f32(f32.data_ptr(), f32.data_ptr(), f32.data_ptr(), f32.data_ptr(), None, 16, 0.1, None, 0.9,
                              0.999, 1e-8, 0.0, 1, None, 0.0, None, 0, 1.0, 0.0, None, f32.data_ptr(), None, None)
bad += rc < 0                                            # loss_watch without nonfinite_step
sh = N.make_bf16_shadow([(2, 16, 16, torch.zeros(256, dtype=torch.bfloat16), torch.zeros(256, dtype=torch.bfloat16))])
rc = lib.stdadk_bf16_shadow_refresh(f32.data_ptr(), ctypes.byref(sh), None)
bad += rc < 0                                            # region offset not a multiple of 4
assert bad == 6, bad
assert lib.stdadk_last_error()
print(f"asan driver: {calls} step-level calls through the sanitized host code, {bad} argument errors returned cleanly")
