"""GPU parity tests added in round 2 (all through the C ABI, checked against the float64 oracle / the goldens
made from the real reference):

  * BASELINE config C1's shape (one 32 x 32 level, 3-layer MLP) on both kernel paths;
  * BASELINE config C4 (4 levels, 49 728 knots) at B = 4096 on the window path: EVERY gradient against the oracle;
  * the data-parallel arithmetic of TrainStep with two virtual ranks on ragged shards: plain MSE, learnable knots
    with penalties and damping, the delta head with P_nc(delta), first-layer sparsity penalties -- against the
    union-batch goldens of the reference;
  * rank-offset dropout streams, seeds drawn from torch's generator;
  * the pipelined batch preparation right behind a device randperm (stream ordering), both event kinds;
  * Predictor / ModelEMA staying coherent with an engine that trains through raw pointers;
  * SpatialBasisEmbedding.forward differentiable w.r.t. learnable knots.
"""
import os

import numpy as np
import pytest
import torch

from golden import cases
from oracle import stdadk_oracle as orc

import test_gpu_parity as T      # model builders / digest checks of the round-1 tests

pytestmark = pytest.mark.gpu
TOL = 1e-5


def dev():
    return T.dev()


# ------------------------------------------------------------------ BASELINE config C1 (single resolution)
@pytest.mark.parametrize("dense", [False, True])
def test_c1_single_resolution_1024_knots(dense):
    """KAUST 1a shape: one 32 x 32 Wendland level, 70 temporal knots, hidden 256/256/128.
    The 1a file has no time column, so every t is 0 (T = 1, train_st_interp.py:439); half the rows here take t = 0,
    the rest random times.  y, loss and every gradient against the float64 oracle."""
    cfg = dict(p=0, k_spatial_centers=[1024], k_temporal_centers=[10, 15, 45], hidden_dims=[256, 256, 128],
               layernorm=True, basis="wendland", output_dim=1, B=1500, seed=211)
    X, coords, t, y = cases.make_inputs(cfg)
    t[: cfg["B"] // 2] = 0.0
    d = dev()
    m = T.build_model(cfg)
    m.force_dense_path = dense
    m.train()
    yp = m(*(torch.from_numpy(a).to(d) for a in (X, coords, t)))
    loss = torch.nn.MSELoss()(yp, torch.from_numpy(y).to(d))
    loss.backward()
    params = {k: v.astype(np.float64) for k, v in cases.make_state(cfg).items()}
    yo, loss_o, go = orc.train_step_grads(X, coords, t, y, params, cfg)
    assert np.abs(yp.detach().cpu().numpy() - yo).max() <= TOL * max(1.0, np.abs(yo).max())
    assert abs(loss.item() - loss_o) <= TOL * loss_o
    for k, p in m.named_parameters():
        assert T.rel_l2(p.grad.cpu().numpy(), go[k]) <= TOL, k


def test_c1_engine_step_on_1024_knots():
    """The fused engine on the C1 shape (1 024 knots: the smallest table that takes the window path by
    default): one optimiser step against the oracle's clip + AdamW + EMA."""
    from stnf.engine import TrainStep
    cfg = dict(p=0, k_spatial_centers=[1024], k_temporal_centers=[10, 15, 45], hidden_dims=[256, 256, 128],
               layernorm=True, basis="wendland", output_dim=1, B=2000, seed=212)
    X, coords, t, y = cases.make_inputs(cfg)
    d = dev()
    m = T.build_model(cfg)
    m.force_window_path = False          # the library's own choice for this table size
    m.train()
    o = cases.OPT
    eng = TrainStep(m, lr=o["lr"], weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"],
                    grad_clip=o["grad_clip"], ema_decay=o["ema_decay"], max_batch=cfg["B"])
    assert eng.uses_window
    eng.step(None, *(torch.from_numpy(a).to(d) for a in (coords, t, y)))
    loss = eng.mean_loss()
    params = {k: v.astype(np.float64) for k, v in cases.make_state(cfg).items()}
    _, loss_o, grads = orc.train_step_grads(X, coords, t, y, params, cfg)
    mm = {k: np.zeros_like(v) for k, v in params.items()}
    vv = {k: np.zeros_like(v) for k, v in params.items()}
    sh = {k: v.copy() for k, v in params.items()}
    orc.adamw_ema_step(params, grads, mm, vv, sh, 1, o["lr"], o["weight_decay"], o["betas"], o["eps"],
                       o["grad_clip"], o["ema_decay"])
    assert abs(loss - loss_o) <= TOL * loss_o
    for k, p in m.named_parameters():
        assert T.rel_l2(p.detach().cpu().numpy(), params[k]) <= 5e-5, k


# ------------------------------------------------------------------ BASELINE config C4 at B = 4096
def test_c4_b4096_window_all_gradients_vs_oracle():
    """Config C4's model (32^2 + 64^2 + 128^2 + 168^2 = 49 728 knots, 12.85 M parameters) on the window path at
    the bench batch: y, loss and ALL gradients (the 12.75 M entries of dW0 included) against the float64 oracle,
    not against the dense kernels."""
    cfg = dict(p=0, k_spatial_centers=[1024, 4096, 16384, 28224], k_temporal_centers=[10, 15, 45],
               hidden_dims=[256, 256, 128], layernorm=True, basis="wendland", output_dim=1, B=4096, seed=62)
    X, coords, t, y = cases.make_inputs(cfg)
    d = dev()
    m = T.build_model(cfg)
    m.train()
    yp = m(*(torch.from_numpy(a).to(d) for a in (X, coords, t)))
    loss = torch.nn.MSELoss()(yp, torch.from_numpy(y).to(d))
    loss.backward()
    params = {k: v.astype(np.float64) for k, v in cases.make_state(cfg).items()}
    yo, loss_o, go = orc.train_step_grads(X, coords, t, y, params, cfg)
    assert np.abs(yp.detach().cpu().numpy() - yo).max() <= TOL * max(1.0, np.abs(yo).max())
    assert abs(loss.item() - loss_o) <= TOL * loss_o
    worst = 0.0
    for k, p in m.named_parameters():
        e = T.rel_l2(p.grad.cpu().numpy(), go[k])
        worst = max(worst, e)
        assert e <= TOL, (k, e)
    # knots no observation reaches have an exactly zero column of dW0 in both
    g0 = m._body[0].weight.grad.cpu().numpy()
    assert g0.shape == go["mlp.0.weight"].shape == (256, 49798)
    zk, zo = np.abs(g0).sum(0) == 0, np.abs(go["mlp.0.weight"]).sum(0) == 0
    assert np.all(zk[zo])                                          # oracle zero => kernel zero
    # (the fp32 kernel may flush a (1-r)^6 of ~1e-40 the float64 oracle still carries: negligible columns only)
    assert np.abs(go["mlp.0.weight"][:, zk & ~zo]).max(initial=0.0) <= 1e-25
    print(f"C4 B=4096 worst gradient rel-L2 vs oracle: {worst:.2e}")


# ------------------------------------------------------------------ data-parallel arithmetic, two virtual ranks
def _virtual_steps(eng, X, coords, t, y, cut, steps):
    """`steps` optimisation steps where the batch is split at row `cut` between two virtual ranks of ONE engine
    (the model is replicated under data parallelism): each rank's share via the split path with the GLOBAL row
    count, the two gradient buffers summed as the all-reduce would, then clip + AdamW + EMA once."""
    B = coords.shape[0]
    losses = []
    for _ in range(steps):
        acc = None
        for r, (lo, hi) in enumerate(((0, cut), (cut, B))):
            eng.set_virtual_rank(r)
            eng._enqueue_grads(X[lo:hi].contiguous() if X is not None else None, coords[lo:hi].contiguous(),
                               t[lo:hi].contiguous().view(-1), y[lo:hi].contiguous(), hi - lo, B)
            acc = eng.grad.clone() if acc is None else acc + eng.grad
        eng.grad.copy_(acc)
        eng._enqueue_optimizer()
        eng._stepped(B)
        losses.append(eng.mean_loss())
    return losses


def _check_params(m, eng, g, cfg, tol):
    for k, p in m.named_parameters():
        T.check_vs_digest(p.detach().cpu().numpy(), g, "p", k, cfg["seed"] + 7, tol=tol)
    eng.swap_in_ema()
    for k, p in m.named_parameters():
        T.check_vs_digest(p.detach().cpu().numpy(), g, "ema", k, cfg["seed"] + 7, tol=tol)
    eng.swap_in_ema()


@pytest.mark.parametrize("name", ["default227", "c2_b257"])
def test_virtual_ranks_mse_match_union_batch_golden(name):
    """global_rows != local rows, gradient summed over ragged shards (100 / B-100 rows), clip on the summed
    gradient: equal to the reference's single-process steps on the union batch (G6 goldens)."""
    from stnf.engine import TrainStep
    cfg = cases.MODEL_CASES[name]
    g = T.load(name)
    o = cases.OPT
    d = dev()
    m = T.build_model(cfg)
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, lr=o["lr"], weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"],
                    grad_clip=o["grad_clip"], ema_decay=o["ema_decay"], max_batch=cfg["B"], world_size=2)
    assert not eng._whole_step
    losses = _virtual_steps(eng, X if cfg["p"] else None, coords, t, y, 100, o["steps"])
    ref = g["opt_losses64"]
    assert np.abs(np.array(losses) - ref).max() <= 5 * TOL * max(1.0, np.abs(ref).max())
    _check_params(m, eng, g, cfg, 2e-5)


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("name", ["default227_learn", "c2_b257_learn"])
def test_virtual_ranks_learnable_knots_match_union_batch_golden(name, dense):
    """Learnable knots: each rank adds 1/world of the domain / movement penalty gradients and damps its own
    share (the same linear map as damping the sum); two AdamW groups with their own clip norms on the summed
    gradient."""
    from stnf.engine import TrainStep
    m, cfg, kn, g = T.build_learn_model(name)
    o = cases.OPT
    d = dev()
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, lr=o["lr"], weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"],
                    grad_clip=o["grad_clip"], ema_decay=o["ema_decay"], max_batch=cfg["B"],
                    basis_lr_ratio=cases.BASIS_LR_RATIO, basis_clip_ratio=cases.BASIS_CLIP_RATIO,
                    domain_penalty_weight=kn.get("domain_penalty_weight", 0.0),
                    movement_penalty_weight=kn.get("movement_penalty_weight", 0.0), force_dense=dense, world_size=2)
    losses = _virtual_steps(eng, X if cfg["p"] else None, coords, t, y, 131, o["steps"])
    ref = g["opt_losses64"]
    assert np.abs(np.array(losses) - ref).max() <= 5 * TOL * max(1.0, np.abs(ref).max()), (losses, ref)
    _check_params(m, eng, g, cfg, 1e-4)


@pytest.mark.parametrize("name", ["default227_delta5", "default227_mq5", "c2_b257_mq3"])
def test_virtual_ranks_quantile_heads_match_union_batch_golden(name):
    """Multi-quantile objectives: the prediction-level non-crossing penalty is per row (shards add up), the
    parameter-level P_nc(delta) is shared 1/world per rank."""
    from stnf.engine import TrainStep
    m, cfg, lc = T.build_quantile_model(name)
    g = T.load(name)
    o = cases.OPT
    d = dev()
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, lr=o["lr"], weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"],
                    grad_clip=o["grad_clip"], ema_decay=o["ema_decay"], max_batch=cfg["B"],
                    loss="pinball", quantile_levels=lc["taus"], non_crossing_weight=lc.get("nc_weight", 0.0),
                    non_crossing_power=lc.get("nc_power", 1), non_crossing_lambda=lc.get("nc_lambda", 0.0),
                    world_size=2)
    losses = _virtual_steps(eng, X if cfg["p"] else None, coords, t, y, 77, o["steps"])
    ref = g["opt_losses64"]
    assert np.abs(np.array(losses) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), (losses, ref)
    _check_params(m, eng, g, cfg, 1e-4)


@pytest.mark.parametrize("name", ["default227_sp_sg", "c2_b257_sp_sg"])
def test_virtual_ranks_sparsity_penalty_match_union_batch_golden(name):
    """First-layer sparsity penalties: 1/world of their gradient per rank."""
    from stnf.engine import TrainStep
    m, cfg, sp, _ = T.build_sparsity_model(name)
    g = T.load(name)
    o = cases.OPT
    d = dev()
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, lr=o["lr"], weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"],
                    grad_clip=o["grad_clip"], ema_decay=o["ema_decay"], max_batch=cfg["B"],
                    sparsity_penalty_type=sp["kind"], sparsity_lambda_l1=sp["lambda_l1"],
                    sparsity_lambda_group=sp["lambda_group"],
                    sparsity_apply_to_spatial=sp.get("apply_spatial", True),
                    sparsity_apply_to_temporal=sp.get("apply_temporal", True), world_size=2)
    losses = _virtual_steps(eng, X if cfg["p"] else None, coords, t, y, 60, o["steps"])
    ref = g["opt_losses64"]
    assert np.abs(np.array(losses) - ref).max() <= 5 * TOL * max(1.0, np.abs(ref).max()), (losses, ref)
    _check_params(m, eng, g, cfg, 1e-4)


def test_empty_local_batch_contributes_a_zero_gradient():
    """A rank without rows in a step (caller-made batches) must hand the all-reduce zeros, not the previous
    step's gradient; with parameter-level penalties it raises instead of dropping its share."""
    from stnf.engine import TrainStep
    cfg = cases.MODEL_CASES["default227"]
    d = dev()
    m = T.build_model(cfg)
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, max_batch=cfg["B"], world_size=2)
    eng._enqueue_grads(None, coords, t.view(-1), y, cfg["B"], cfg["B"])
    assert float(eng.grad.abs().sum()) > 0
    eng._enqueue_grads(None, coords[:0], t[:0].view(-1), y[:0], 0, cfg["B"])
    assert float(eng.grad.abs().sum()) == 0.0
    m2, cfg2, kn, _ = T.build_learn_model("default227_learn")
    eng2 = TrainStep(m2, max_batch=cfg2["B"], world_size=2, domain_penalty_weight=0.01)
    with pytest.raises(RuntimeError, match="empty local batch"):
        eng2._enqueue_grads(None, coords[:0], t[:0].view(-1), y[:0], 0, cfg2["B"])


# ------------------------------------------------------------------ dropout streams
def test_dropout_seed_follows_torch_generator_and_rank():
    """The engine's dropout seed is drawn from torch's generator at construction (set_seed decides the masks, as
    it does for nn.Dropout in the reference), `seed=` pins it, and the rank is mixed in: two ranks that see the
    SAME rows drop different units, the same rank reproduces its masks."""
    from stnf.engine import TrainStep, _rank_seed
    cfg = cases.MODEL_CASES["default227"]
    d = dev()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))

    def grads(seed=None, manual=None, rank=0):
        m = T.build_model(cfg, dropout=0.3)
        m.train()
        if manual is not None:
            torch.manual_seed(manual)
        eng = TrainStep(m, max_batch=cfg["B"], world_size=2, seed=seed)
        eng.set_virtual_rank(rank)
        eng._enqueue_grads(None, coords, t.view(-1), y, cfg["B"], cfg["B"])
        return eng.grad.clone(), eng.seed, eng.base_seed

    a, sa, ba = grads(manual=5)
    b, sb, bb = grads(manual=5)
    c, sc, bc = grads(manual=6)
    assert ba == bb != bc and torch.equal(a, b) and not torch.equal(a, c)
    e, se, _ = grads(seed=1234, rank=0)
    f, sf, _ = grads(seed=1234, rank=1)
    e2, _, _ = grads(seed=1234, rank=0)
    assert se != sf and se == _rank_seed(1234, 0) == 1234 and sf == _rank_seed(1234, 1)
    assert torch.equal(e, e2) and not torch.equal(e, f)


# ------------------------------------------------------------------ pipelined batch preparation
@pytest.mark.parametrize("torch_events", [False, True])
def test_pipelined_preparation_behind_a_device_randperm(torch_events, monkeypatch):
    """run_epoch on a set large enough that the device randperm is a multi-kernel sort: the side stream's
    gather + binning of batch 1 must be ordered behind it (it used to wait only for an event of an
    earlier step).  Pipelined epochs == the same batches stepped one by one, bit for bit; also on the
    torch.cuda.Event branch of the event factory."""
    from stnf import engine as E
    from stnf.dataio.device_dataset import DeviceDataset
    monkeypatch.setattr(E, "_FORCE_TORCH_EVENTS", torch_events)
    cfg = cases.MODEL_CASES["c2_b257"]
    d = dev()
    n, B = 300_000, 4096
    g = torch.Generator().manual_seed(9)
    coords = torch.rand(n, 2, generator=g).to(d)
    t = torch.rand(n, 1, generator=g).to(d)
    y = torch.randn(n, 1, generator=g).to(d)
    ds = DeviceDataset(coords, t, y)
    res = []
    for pipelined in (True, False):
        m = T.build_model(cfg)
        m.train()
        eng = E.TrainStep(m, lr=1e-3, ema_decay=0.99, max_batch=B, seed=3)
        gen = torch.Generator(device=d).manual_seed(17)
        if pipelined:
            for _ in range(2):
                # burn the allocator's cached index block so that every epoch's randperm is fresh device work
                loss = eng.run_epoch(ds, B, generator=gen)
        else:
            for _ in range(2):
                for idx in ds.epoch_batches(B, generator=gen):
                    eng.step_indexed(ds.coords, ds.t, ds.y, idx)
                loss = eng.mean_loss()
        res.append((loss, eng.flat.clone()))
        if pipelined:
            kind = type(eng._pipe["announce"]).__name__
            assert kind == ("Event" if torch_events else "_LightEvent"), kind
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[1][0])
    assert torch.equal(res[0][1], res[1][1])


# ------------------------------------------------------------------ Predictor / EMA coherence
def test_predictor_follows_training_and_ema_swaps():
    """A Predictor kept across optimiser steps, an EMA swap and a later-built engine predicts with the CURRENT
    weights (delta head included: its derived output layer is refreshed)."""
    from stnf.engine import TrainStep, Predictor
    m, cfg, lc = T.build_quantile_model("default227_delta5")
    d = dev()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    m.eval()
    pr = Predictor(m, chunk=1024)               # built BEFORE the engine re-points the parameter storage
    y0 = pr.predict(coords, t).clone()
    with torch.no_grad():
        assert torch.equal(y0, m(None, coords, t))
    m.train()
    eng = TrainStep(m, lr=1e-2, ema_decay=0.9, max_batch=cfg["B"], loss="pinball", quantile_levels=lc["taus"],
                    non_crossing_lambda=lc.get("nc_lambda", 0.0))
    for _ in range(3):
        eng.step(None, coords, t, y)
    m.eval()
    with torch.no_grad():
        want = m(None, coords, t)
    got = pr.predict(coords, t)
    assert not torch.equal(got, y0) and torch.equal(got, want)
    eng.swap_in_ema()
    with torch.no_grad():
        want_ema = m(None, coords, t)
    assert not torch.equal(want_ema, want) and torch.equal(pr.predict(coords, t), want_ema)
    eng.swap_in_ema()
    assert torch.equal(pr.predict(coords, t), want)


def test_model_ema_keeps_engine_storage():
    """ModelEMA.apply_shadow / restore copy INTO the parameters' storage: after a validation pass under EMA
    weights the parameters are still views of the engine's flat buffer and training continues on them."""
    from stnf.engine import TrainStep
    from stnf.utils import ModelEMA
    cfg = cases.MODEL_CASES["default227"]
    d = dev()
    m = T.build_model(cfg)
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, lr=1e-2, max_batch=cfg["B"])
    ema = ModelEMA(m, decay=0.5)
    eng.step(None, coords, t, y)
    ema.update(m)
    ptrs = [p.data_ptr() for p in m.parameters()]
    before = eng.flat.clone()
    ema.apply_shadow()
    assert [p.data_ptr() for p in m.parameters()] == ptrs and not torch.equal(eng.flat, before)
    for n, p in m.named_parameters():
        assert torch.equal(p.data, ema.shadow[n])
    ema.restore()
    assert [p.data_ptr() for p in m.parameters()] == ptrs and torch.equal(eng.flat, before)
    eng.step(None, coords, t, y)
    sd = m.state_dict()
    for n, p in m.named_parameters():
        assert torch.equal(sd[n], p.data) and not torch.equal(p.data.reshape(-1), torch.zeros_like(p.data).reshape(-1))
    assert not torch.equal(eng.flat, before)


# ------------------------------------------------------------------ module-level autograd of the embedding
@pytest.mark.parametrize("basis", ["wendland", "gaussian", "triangular"])
def test_spatial_embedding_is_differentiable_wrt_learnable_knots(basis):
    """model.spatial_basis(coords) with learnable knots carries gradients into centres and log-bandwidths
    (stdadk_knot_grad_f32), as the reference's module does; checked against float64 autograd of the same
    formulas (st_interp.py:433-491)."""
    from stnf.models.st_interp import SpatialBasisEmbedding
    d = dev()
    rs = np.random.RandomState(3)
    sb = SpatialBasisEmbedding(n_centers=[25, 81], learnable=True, basis_function=basis).to(d)
    with torch.no_grad():
        sb.centers.add_(torch.from_numpy(rs.normal(0, 0.03, (sb.k, 2)).astype(np.float32)).to(d))
        sb.log_bandwidths.add_(torch.from_numpy(rs.normal(0, 0.2, sb.k).astype(np.float32)).to(d))
    B = 301
    coords = torch.from_numpy(rs.uniform(0, 1, (B, 2)).astype(np.float32)).to(d)
    with torch.no_grad():
        coords[0] = sb.centers[3]                          # zero distance: no pull (cdist's backward)
    w = torch.from_numpy(rs.standard_normal((B, sb.k)).astype(np.float32)).to(d)
    phi = sb(coords)
    assert phi.requires_grad
    (phi * w).sum().backward()
    # float64 restatement on the host
    c = sb.centers.detach().cpu().double().requires_grad_(True)
    lb = sb.log_bandwidths.detach().cpu().double().requires_grad_(True)
    x = coords.cpu().double()
    dist = torch.cdist(x, c, compute_mode="donot_use_mm_for_euclid_dist")    # zero gradient at zero distance
    cal = SpatialBasisEmbedding.CALIBRATION_FACTORS[basis]
    r = dist / (torch.exp(lb)[None, :] * cal)
    if basis == "wendland":
        rc = torch.clamp(r, max=1.0)
        ref = (1 - rc) ** 6 * (35 * rc ** 2 + 18 * rc + 3) / 3
    elif basis == "gaussian":
        ref = torch.exp(-0.5 * r ** 2)
    else:
        ref = torch.clamp(1 - r, min=0.0)
    assert (phi.detach().cpu().double() - ref.detach()).abs().max().item() <= TOL
    (ref * w.cpu().double()).sum().backward()
    assert T.rel_l2(sb.centers.grad.cpu().numpy(), c.grad.numpy()) <= 2e-5
    assert T.rel_l2(sb.log_bandwidths.grad.cpu().numpy(), lb.grad.numpy()) <= 2e-5
    # without grad mode the forward is the plain kernel and returns the same values
    with torch.no_grad():
        assert torch.equal(sb(coords), phi.detach())


# ------------------------------------------------------------------ two real ranks on the one GPU (gloo)
# Adam's eps in the two-process comparisons.  With the default 1e-8 an entry whose gradient is at rounding level gets a
# full lr-sized step whose SIGN is decided by the summation order (the two ranks' partial sums against one process's):
# a few dozen of the 2.76 M entries then differ by 2 lr after a handful of steps, whatever the arithmetic.  1e-3 keeps
# the update linear in such gradients, so that the comparison measures the data-parallel arithmetic and not that.
_DP_EPS = 1e-3


def _dp_worker(rank, world, port, sizes, B, learn, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p_ in (root, os.path.join(root, "st-dadk_amd"), os.path.join(root, "tests")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stnf.engine import TrainStep
        from stnf.dataio.device_dataset import DeviceDataset
        import test_gpu_parity as TT
        d = torch.device("cuda:0")
        torch.cuda.set_device(d)
        coords, t, y = _dp_data(sum(sizes), d)
        lo = sum(sizes[:rank])
        ds = DeviceDataset(coords[lo:lo + sizes[rank]].contiguous(), t[lo:lo + sizes[rank]].contiguous(),
                           y[lo:lo + sizes[rank]].contiguous())
        m, kw = _dp_model(learn)
        eng = TrainStep(m, lr=1e-3, eps=_DP_EPS, ema_decay=0.9, max_batch=B, seed=5, **kw)
        assert eng.distributed and eng.world == world and eng.rank == rank
        losses = [eng.run_epoch(ds, B, shuffle=False) for _ in range(2)]
        if rank == 0:
            q.put((losses, eng.flat.cpu().numpy(), eng.step_count))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _dp_data(n, d):
    g = torch.Generator().manual_seed(1234)
    coords = torch.rand(n, 2, generator=g)
    t = torch.rand(n, 1, generator=g)
    y = torch.sin(5 * coords[:, :1]) + 0.1 * torch.randn(n, 1, generator=g)
    return coords.to(d), t.to(d), y.to(d)


def _dp_model(learn):
    if learn:
        m, cfg, kn, _ = T.build_learn_model("c2_b257_learn")
        return m.train(), dict(domain_penalty_weight=0.01, movement_penalty_weight=0.02)
    return T.build_model(cases.MODEL_CASES["c2_b257"]).train(), {}


@pytest.mark.parametrize("learn", [False, True])
@pytest.mark.parametrize("sizes,B", [((2048, 2047), 1024), ((1025, 1024), 1024)])
def test_two_rank_run_epoch_equals_single_process_union(sizes, B, learn):
    """TrainStep.run_epoch under torch.distributed with RAGGED shards (the cases of ADVICE r1: a rank one row
    short, a rank one batch short), two processes on this GPU over gloo: no hang (every rank enters the same
    collectives), and after two epochs the replicated parameters equal a single-process run whose step i takes
    the union of the ranks' i-th batches -- the count-weighted reduction, the 1/world penalty shares and the
    post-reduce clip are all in that comparison."""
    import socket
    import torch.multiprocessing as mp
    from stnf import distributed as D
    from stnf.engine import TrainStep
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, sizes, B, learn, q)) for r in range(2)]
    for p in procs:
        p.start()
    losses, flat, steps = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # single process, union batches
    d = dev()
    coords, t, y = _dp_data(sum(sizes), d)
    table = D.epoch_schedule(list(sizes), B)
    assert steps == 2 * len(table)
    m, kw = _dp_model(learn)
    eng = TrainStep(m, lr=1e-3, eps=_DP_EPS, ema_decay=0.9, max_batch=2 * B, seed=5, **kw)
    ref_losses = []
    for _ in range(2):
        off = [0, sizes[0]]
        for row in table:
            idx = torch.cat([torch.arange(off[r], off[r] + row[r], device=d) for r in range(2)])
            off = [off[r] + row[r] for r in range(2)]
            eng.step(None, coords[idx], t[idx], y[idx])
        ref_losses.append(eng.mean_loss())
    # every rank reports the mean objective of ITS rows; with near-equal shards that is close to the union mean
    assert abs(losses[-1] - ref_losses[-1]) <= 0.05 * abs(ref_losses[-1])
    # (Adam's m/sqrt(v) turns rounding-level differences of near-zero gradient entries into lr-sized updates)
    assert T.rel_l2(flat, eng.flat.cpu().numpy()) <= 1e-4


# ------------------------------------------------------------------ scattered (non-grid) knots on the window path
def _scattered_model(sizes, basis, learnable, seed, hidden=(256, 128)):
    """Knots drawn from scattered sites (random_site initialiser, st_interp.py:266-343) plus a few pushed outside
    the unit square and a few with doubled bandwidths; golden-style deterministic weights."""
    from stnf.models import STInterpMLP
    rs = np.random.RandomState(seed)
    pts = np.concatenate([rs.uniform(0, 1, (12000, 2)), 0.3 + 0.05 * rs.standard_normal((6000, 2))]).astype(np.float32)
    np.random.seed(seed)
    cfg = dict(p=0, k_spatial_centers=list(sizes), k_temporal_centers=[10, 15], hidden_dims=list(hidden),
               layernorm=True, basis=basis, output_dim=1, B=0, seed=seed)
    m = STInterpMLP(p=0, k_spatial_centers=cfg["k_spatial_centers"], k_temporal_centers=cfg["k_temporal_centers"],
                    hidden_dims=cfg["hidden_dims"], dropout=0.0, layernorm=True, spatial_learnable=learnable,
                    spatial_init_method="random_site", spatial_basis_function=basis, train_coords=pts)
    st = cases.make_state(cfg)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k in st:
                p.copy_(torch.from_numpy(st[k].copy()))
        sb = m.spatial_basis
        c = sb.centers
        c[3] = torch.tensor([-0.04, 0.5]); c[7] = torch.tensor([1.03, -0.02]); c[11] = torch.tensor([0.5, 1.06])
        if learnable:
            sb.log_bandwidths[5] += math.log(2.0); sb.log_bandwidths[100] += math.log(1.7)
        else:
            sb._bandwidths[5] *= 2.0; sb._bandwidths[100] *= 1.7
    return m.to(dev()), cfg, st


import math  # noqa: E402


@pytest.mark.parametrize("basis,learnable,B", [("wendland", True, 300), ("wendland", True, 4096), ("wendland", True, 9000),
                                               ("wendland", False, 300), ("wendland", False, 4096), ("wendland", False, 9000),
                                               ("triangular", False, 4096), ("triangular", True, 300)])
def test_scattered_knots_window_path_matches_materialised_and_oracle(basis, learnable, B):
    """5 120 non-grid knots in two levels (1024 + 4096): the window kernels over per-level knot cell lists
    (STDADK_FLAG_SCATTERED) against the materialising kernels and the float64 oracle -- y, loss, every gradient
    (knot gradients included when learnable).  Knots outside the unit square and knots with inflated bandwidths
    are in the table; the three batch sizes take the one-launch step kernel, the separate kernels and the
    64-row-tile kernels.  (Learnable TRIANGULAR knots only at 300 rows: phi' jumps from -1 to 0 at r = 1, so among
    the ~half a million supported pairs of a 9 000-row batch the one pair that fp32 and float64 put on different
    sides of r = 1 moves the whole centre gradient by 7e-4 -- a property of that basis, not of a kernel.)"""
    d = dev()
    res = {}
    for path in ("window", "dense"):
        m, cfg, st = _scattered_model([1024, 4096], basis, learnable, 77)
        cfg["B"] = B
        X, coords, t, y = cases.make_inputs(cfg)
        m.force_dense_path = path == "dense"
        desc = m._basis_desc()
        assert (desc.n_levels == 2 and desc.side[0] == 1024 and desc.side[1] == 4096) if path == "window" else desc.n_levels == 0
        st_ = m._step_state(d, force_dense=m.force_dense_path)
        from stnf import _native as N
        assert N.step_uses_window(st_.basis, st_.desc, st_.flags) == (path == "window")
        m.train()
        yp = m(None, torch.from_numpy(coords).to(d), torch.from_numpy(t).to(d))
        loss = torch.nn.functional.mse_loss(yp, torch.from_numpy(y).to(d))
        loss.backward()
        res[path] = (yp.detach().cpu().numpy(), loss.item(), {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}, m)
    yw, lw, gw, m = res["window"]
    yd, ld, gd, _ = res["dense"]
    assert np.abs(yw - yd).max() <= 2e-6 * max(1.0, np.abs(yd).max()) and abs(lw - ld) <= 2e-6 * ld
    for k in gd:
        # (knot gradients: sum_b q_b (dZ_b . w_k) taken as (sum_b q_b dZ_b) . w_k on the window path, per row on the
        #  materialising path -- fp32 cancellation differs; the triangular q does not taper towards the support's edge)
        assert T.rel_l2(gw[k], gd[k]) <= (5e-5 if k.startswith("spatial_basis.") else 5e-6), k
    sb = m.spatial_basis
    cen = sb.centers.detach().cpu().numpy().astype(np.float64)
    if learnable:
        params = dict(st)
        params["spatial_basis.centers"] = sb.centers.detach().cpu().numpy()
        params["spatial_basis.log_bandwidths"] = sb.log_bandwidths.detach().cpu().numpy()
        yo, lo, go = orc.learnable_step_grads(X, coords, t, y, params, cfg, {}, params["spatial_basis.centers"])
    else:
        tc = m.temporal_basis.centers.cpu().numpy(); tb = m.temporal_basis.bandwidths.cpu().numpy()
        feat = orc.features(X, orc.spatial_basis(coords, cen, sb.bandwidths.cpu().numpy(), basis),
                            orc.temporal_basis(t, tc, tb), 0)
        yo, cache = orc.mlp_forward(feat, st, len(cfg["hidden_dims"]), True)
        go = orc.mlp_mse_backward(yo, y, cache, st, len(cfg["hidden_dims"]), True)
        lo = orc.mse(yo, y)
    assert np.abs(yw - yo).max() <= TOL * max(1.0, np.abs(yo).max()) and abs(lw - lo) <= TOL * lo
    for k in gw:
        assert T.rel_l2(gw[k], go[k]) <= 2e-5, (k, T.rel_l2(gw[k], go[k]))


def test_scattered_knots_engine_and_grid_inference():
    """The fused engine on scattered learnable knots (knot cell lists rebuilt every step as the knots move): three
    steps on the window path == the materialising path; the default path choice takes the window kernels for this
    table (small supports) and the materialising ones when the bandwidths are inflated to cover the domain; and
    fixed scattered knots through Predictor.predict_grid (per-site half of layer 0 on the window path) == row by row."""
    from stnf.engine import TrainStep, Predictor
    d = dev()
    flats = []
    for dense in (False, True):
        m, cfg, st = _scattered_model([1024, 4096], "wendland", True, 78)
        cfg["B"] = 4096
        X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
        m.train()
        eng = TrainStep(m, lr=1e-3, ema_decay=0.99, max_batch=4096, force_dense=dense, domain_penalty_weight=0.01,
                        movement_penalty_weight=0.02)
        assert eng.uses_window == (not dense)
        for _ in range(3):
            eng.step(None, coords, t, y)
        flats.append((eng.mean_loss(), eng.flat.clone()))
    assert abs(flats[0][0] - flats[1][0]) <= 1e-5 * abs(flats[1][0])
    assert T.rel_l2(flats[0][1].cpu().numpy(), flats[1][1].cpu().numpy()) <= 5e-5
    # supports as wide as the domain: the estimate sends the table to the materialising kernels
    m2, _, _ = _scattered_model([1024, 4096], "wendland", True, 79)
    with torch.no_grad():
        m2.spatial_basis.log_bandwidths.fill_(0.0)
    assert m2._basis_desc().n_levels == 0 and not TrainStep(m2, max_batch=256).uses_window
    # fixed scattered knots: site x time grid
    m3, cfg3, _ = _scattered_model([1024, 4096], "triangular", False, 80)
    m3.eval()
    g = torch.Generator().manual_seed(4)
    S, Tn = 5001, 3
    sites = torch.rand(S, 2, generator=g).to(d)
    tv = torch.linspace(0, 1, Tn).to(d)
    pr = Predictor(m3, chunk=2048)
    yg = pr.predict_grid(sites, tv)
    yr = pr.predict(sites.repeat(Tn, 1), tv.repeat_interleave(S)).view(Tn, S, 1)
    assert (yg - yr).abs().max().item() <= 2e-6 * max(1.0, yr.abs().max().item())


# ------------------------------------------------------------------ Q > 1 heads on the 32- and 64-row tiles
@pytest.mark.gpu
@pytest.mark.parametrize("B", [9000, 20000])
def test_quantile_step_on_large_row_tiles(B):
    """The fused step with a 5-quantile head (check loss + non-crossing penalty inside the launch) at batch sizes
    whose tail launches carry 32 (B = 9000) and 64 (B = 20000) rows per workgroup: the head of the backward sums
    the output layer's weight-gradient partials over row groups that meet in LDS there (one group per 16-row tile
    below 8192 rows, which the golden cases cover).  lr = 0, so the engine's gradient buffer is the gradient of
    the reference's objective at the golden weights: against the float64 oracle."""
    from golden import cases
    from oracle import stdadk_oracle as orc
    from stnf.engine import TrainStep
    from stnf.models import STInterpMLP
    cfg, lc = cases.quantile_cfg("default227_mq5")
    cfg = dict(cfg, B=B, seed=500 + B % 97)
    X, coords, t, y = cases.make_inputs(cfg)
    st = cases.make_state(cfg)
    d = torch.device("cuda", 0)
    m = STInterpMLP(p=cfg["p"], k_spatial_centers=cfg["k_spatial_centers"], k_temporal_centers=cfg["k_temporal_centers"],
                    hidden_dims=cfg["hidden_dims"], dropout=0.0, layernorm=cfg["layernorm"],
                    spatial_basis_function=cfg["basis"], output_dim=cfg["output_dim"])
    with torch.no_grad():
        for (_, p), (k, v) in zip(m.named_parameters(), st.items()):
            p.copy_(torch.from_numpy(v.copy()))
    m.force_window_path = True
    m = m.to(d).train()
    eng = TrainStep(m, lr=0.0, weight_decay=0.0, grad_clip=0.0, ema_decay=0.99, max_batch=B, loss="pinball",
                    quantile_levels=lc["taus"], non_crossing_weight=lc["nc_weight"], non_crossing_power=lc["nc_power"])
    assert eng.uses_window
    eng.step(None, *(torch.from_numpy(a).to(d) for a in (coords, t, y)))
    loss = eng.mean_loss()
    yo, lo, go = orc.quantile_step_grads(X, coords, t, y, st, cfg, lc)
    assert abs(loss - lo) <= 1e-5 * max(1.0, abs(lo)), (loss, lo)
    first_w = m._body[0].weight
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    assert len(names) == len(eng.grad_views)
    for n, gv in zip(names, eng.grad_views):
        got = gv.detach().cpu().numpy()
        ref = go[n]
        if dict(m.named_parameters())[n] is first_w:
            ref = ref.T                                   # the engine keeps W0 (and dW0) transposed
        err = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
        assert err <= 2e-5, (n, err)


# ------------------------------------------------------------------ dense-grid predictions (the driver's A10 caller)
@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 5])
def test_predict_all_times_matches_the_per_slice_loop(q):
    """stnf.utils.predictions.predict_all_times (one site x time grid call) against the reference driver's loop of
    one model call per time slice (scripts/train_st_interp.py:1228-1248), median column for multi-quantile models."""
    from stnf.models import STInterpMLP
    from stnf.utils.predictions import predict_all_times
    d = torch.device("cuda", 0)
    torch.manual_seed(11 + q)
    rs = np.random.RandomState(11 + q)
    S, Tn = 1501, 7
    coords = rs.uniform(-0.02, 1.02, (S, 2)).astype(np.float32)
    m = STInterpMLP(p=0, k_spatial_centers=[1024, 4096], k_temporal_centers=[10, 15], hidden_dims=[256, 256, 128],
                    dropout=0.1, layernorm=True, output_dim=q).to(d)
    m.train()
    got = predict_all_times(m, coords, Tn)
    assert m.training and got.shape == (Tn, S) and got.dtype == np.float64
    m.eval()
    c = torch.from_numpy(coords).to(d)
    with torch.no_grad():
        for ti in range(Tn):
            ref = m(torch.zeros(S, 0, device=d), c, torch.full((S, 1), ti / (Tn - 1), device=d)).cpu().numpy()
            assert np.abs(got[ti] - ref[:, q // 2]).max() <= 2e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
def test_evaluate_model_on_the_device_equals_the_host_path():
    """evaluate_model through Predictor on the GPU == the same model evaluated on host tensors (whose forward is
    pinned to the goldens in tests/test_cpu_module_path.py), for the mean and the multi-quantile metrics."""
    import copy
    from stnf.dataio.device_dataset import DeviceDataset
    from stnf.models import STInterpMLP
    from stnf.utils.predictions import evaluate_model
    d = torch.device("cuda", 0)
    torch.manual_seed(21)
    rs = np.random.RandomState(21)
    n = 40000                                   # several tail tiles of 64 rows
    c = torch.from_numpy(rs.uniform(0, 1, (n, 2)).astype(np.float32))
    t = torch.from_numpy((rs.randint(0, 100, (n, 1)) / 99.0).astype(np.float32))
    y = torch.from_numpy(rs.standard_normal((n, 1)).astype(np.float32))
    taus = [0.1, 0.5, 0.9]
    for q, cfg in ((1, None), (3, dict(regression_type="multi-quantile", quantile_levels=taus))):
        mh = STInterpMLP(p=0, k_spatial_centers=[1024, 4096], k_temporal_centers=[10, 15], hidden_dims=[256, 256, 128],
                         dropout=0.1, layernorm=True, output_dim=q)
        md = copy.deepcopy(mh).to(d)
        got = evaluate_model(md, DeviceDataset(c.to(d), t.to(d), y.to(d)), cfg)
        ref = evaluate_model(mh, DeviceDataset(c, t, y), cfg)
        assert set(got) == set(ref)
        for k in ref:
            assert got[k] == pytest.approx(ref[k], rel=2e-5), k
