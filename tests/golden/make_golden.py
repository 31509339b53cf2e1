#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REAL reference.

Runs only in the build container (needs /root/reference on disk); the GPU box never runs it and
never sees the reference.  The committed .npz files hold data only: the reference's outputs on
the deterministic inputs of cases.py (inputs/weights are rebuilt from seeds on the test side).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

What is captured:
  G1 knots.npz        knot tables of SpatialBasisEmbedding._init_uniform / TemporalBasisEmbedding
  G2/G3/G4 <case>.npz phi, psi, features (fp32 as the reference computes them, fp32 with cdist's
                      direct mode, and float64 "truth" from the same module under .double())
  G5 <case>.npz       y, MSE loss, every parameter gradient (fp32 + float64)
  G6 <case>.npz       parameters + EMA shadow after OPT['steps'] x (fwd, MSE, bwd, clip, AdamW, EMA)
"""
import copy
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import cases  # noqa: E402
from stnf.models.st_interp import (STInterpMLP, SpatialBasisEmbedding,  # noqa: E402
                                   TemporalBasisEmbedding)
from stnf.utils.ema import ModelEMA  # noqa: E402

torch.set_num_threads(8)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def store(out, prefix, key, a64, a32, seed, keep32):
    """Full float64 truth for small tensors, a digest (samples + row/column sums + norm) for
    big ones; the fp32 reference run is kept only as its error against the truth."""
    d = a32.astype(np.float64) - a64
    out[f"{prefix}err32_maxabs/{key}"] = np.float64(np.abs(d).max())
    out[f"{prefix}err32_rell2/{key}"] = np.float64(np.linalg.norm(d.ravel())
                                                   / max(np.linalg.norm(a64.ravel()), 1e-300))
    out[f"{prefix}norm64/{key}"] = np.float64(np.linalg.norm(a64.ravel()))
    if a64.size <= cases.DIGEST_ABOVE:
        out[f"{prefix}64/{key}"] = a64
        if keep32:
            out[f"{prefix}32/{key}"] = a32
    else:
        pos = cases.digest_positions(a64.shape, 2048, seed)
        out[f"{prefix}64s/{key}"] = a64.ravel()[pos]
        if a64.ndim == 2:
            out[f"{prefix}colsum64/{key}"] = a64.sum(0)
            out[f"{prefix}rowsum64/{key}"] = a64.sum(1)


def gen_knots():
    out = {}
    for side in cases.KNOT_SIDES:
        m = SpatialBasisEmbedding(n_centers=[side * side])
        c = m.centers.numpy()
        bw = m.bandwidths.numpy()
        out[f"s{side}_lin"] = torch.linspace(0, 1, side).numpy()
        out[f"s{side}_bw"] = bw[:1].copy()
        out[f"s{side}_centers_sha"] = np.frombuffer(bytes.fromhex(sha(c)), dtype=np.uint8)
        out[f"s{side}_bw_sha"] = np.frombuffer(bytes.fromhex(sha(bw)), dtype=np.uint8)
        if side <= 11:
            out[f"s{side}_centers"] = c.copy()
    # multi-level concatenation order (level offsets)
    m = SpatialBasisEmbedding(n_centers=[25, 81, 121])
    out["ml_centers"] = m.centers.numpy().copy()
    out["ml_bw"] = m.bandwidths.numpy().copy()
    for n in cases.TEMPORAL_NS:
        tm = TemporalBasisEmbedding(n_centers=[n])
        out[f"t{n}_centers"] = tm.centers.numpy().copy()
        out[f"t{n}_bw"] = tm.bandwidths.numpy().copy()
    tm = TemporalBasisEmbedding(n_centers=[10, 15, 45])
    out["tml_centers"] = tm.centers.numpy().copy()
    out["tml_bw"] = tm.bandwidths.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "knots.npz"), **out)
    print("knots.npz", len(out), "arrays")


def build(cfg):
    model = STInterpMLP(p=cfg["p"], k_spatial_centers=cfg["k_spatial_centers"],
                        k_temporal_centers=cfg["k_temporal_centers"],
                        hidden_dims=cfg["hidden_dims"], dropout=0.0, layernorm=cfg["layernorm"],
                        spatial_learnable=False, spatial_init_method="uniform",
                        spatial_basis_function=cfg["basis"], output_dim=cfg["output_dim"],
                        use_delta_reparameterization=bool(cfg.get("delta")))
    st = cases.make_state(cfg)
    sd = model.state_dict()
    keys = [k for k in sd if k.startswith(("mlp.", "mlp_trunk.", "delta_params."))]
    assert sorted(keys) == sorted(st.keys()), (keys, list(st))
    for k, v in st.items():
        assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(v.copy())
    model.load_state_dict(sd)
    return model


def run_once(model, X, coords, t, y):
    model.train()
    model.zero_grad()
    phi = model.spatial_basis(coords)
    psi = model.temporal_basis(t)
    yp = model(X, coords, t)
    loss = torch.nn.MSELoss()(yp, y)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    return phi.detach(), psi.detach(), yp.detach(), loss.detach(), grads


def opt_steps(model, X, coords, t, y):
    o = cases.OPT
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=o["lr"],
                            weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"])
    ema = ModelEMA(model, decay=o["ema_decay"])
    crit = torch.nn.MSELoss()
    losses = []
    model.train()
    for _ in range(o["steps"]):
        opt.zero_grad()
        loss = crit(model(X, coords, t), y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), o["grad_clip"])
        opt.step()
        ema.update(model)
        losses.append(float(loss))
    params = {n: p.detach().clone() for n, p in model.named_parameters()}
    shadow = {n: v.detach().clone() for n, v in ema.shadow.items()}
    return params, shadow, np.array(losses, dtype=np.float64)


def gen_case(name):
    cfg = cases.MODEL_CASES[name]
    X, coords, t, y = (torch.from_numpy(a) for a in cases.make_inputs(cfg))
    full = name in cases.FULL_CASES
    out = {}

    m32 = build(cfg)
    phi32, psi32, y32, loss32, g32 = run_once(m32, X, coords, t, y)

    m64 = build(cfg).double()
    phi64, psi64, y64, loss64, g64 = run_once(m64, X.double(), coords.double(), t.double(),
                                              y.double())

    # cdist direct mode + the reference's own basis function (SURVEY §7 parity protocol (ii))
    sb = m32.spatial_basis
    dist = torch.cdist(coords, sb.centers, compute_mode="donot_use_mm_for_euclid_dist")
    r = dist / (sb.bandwidths * sb.CALIBRATION_FACTORS[sb.basis_function])
    phi32d = {"wendland": sb._wendland, "gaussian": sb._gaussian,
              "triangular": sb._triangular}[sb.basis_function](r)

    out["y32"], out["y64"] = y32.numpy(), y64.numpy()
    out["loss32"], out["loss64"] = np.float32(loss32.item()), np.float64(loss64.item())
    out["psi64"] = psi64.numpy()
    out["psi_err32_maxabs"] = np.float64((psi32.double() - psi64).abs().max().item())
    out["phi_rowsum64"] = phi64.sum(1).numpy()
    out["phi_err32_maxabs"] = np.float64((phi32.double() - phi64).abs().max().item())
    out["phi_err32d_maxabs"] = np.float64((phi32d.double() - phi64).abs().max().item())
    out["y_err32_maxabs"] = np.float64((y32.double() - y64).abs().max().item())
    out["phi64_nnz_per_row"] = (phi64 != 0).sum(1).numpy().astype(np.int32)
    if phi64.numel() <= 50000:
        out["phi64"] = phi64.numpy()
        if full:
            out["phi32"], out["phi32d"] = phi32.numpy(), phi32d.numpy()
    else:
        # sparse truth (row, col, value) for the first 64 rows
        sub = phi64[:min(64, cfg["B"])]
        nz = (sub != 0).nonzero()
        out["phi64_nz_rc"] = nz.numpy().astype(np.int32)
        out["phi64_nz_val"] = sub[nz[:, 0], nz[:, 1]].numpy()
    for k in g64:
        store(out, "g", k, g64[k].numpy(), g32[k].numpy(), cfg["seed"] + 7, full)

    # G6 optimiser steps (float64 truth + fp32 as the reference runs it)
    p32, s32, l32 = opt_steps(build(cfg), X, coords, t, y)
    p64, s64, l64 = opt_steps(build(cfg).double(), X.double(), coords.double(), t.double(),
                              y.double())
    out["opt_losses32"], out["opt_losses64"] = l32, l64
    for k in p64:
        store(out, "p", k, p64[k].numpy(), p32[k].numpy(), cfg["seed"] + 7, full)
        store(out, "ema", k, s64[k].numpy(), s32[k].numpy(), cfg["seed"] + 7, full)

    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB  phi_err32={out['phi_err32_maxabs']:.2e} "
          f"phi_err32d={out['phi_err32d_maxabs']:.2e} y_err32={out['y_err32_maxabs']:.2e}")


# ---------------------------------------------------------------------------------------------
# N3: quantile objectives + delta head, driven exactly as the batch body does
# (train_st_interp.py:617-658) with the reference's own loss functions
# ---------------------------------------------------------------------------------------------
def n3_loss(model, yp, y, lc):
    taus = lc["taus"]
    if len(taus) == 1:
        return ref_train.quantile_loss(yp, y, taus[0])
    losses = [ref_train.quantile_loss(yp[:, i:i + 1], y, q) for i, q in enumerate(taus)]
    loss = torch.mean(torch.stack(losses))
    if lc.get("delta"):
        if lc.get("nc_lambda", 0.0) > 0:
            loss = loss + lc["nc_lambda"] * ref_train.compute_p_nc_delta_penalty(
                model.get_delta_parameters())
    elif lc.get("nc_weight", 0.0) > 0:
        loss = loss + lc["nc_weight"] * ref_train.non_crossing_penalty(
            yp, reduction="mean", power=int(lc.get("nc_power", 1)))
    return loss


def n3_run(model, X, coords, t, y, lc):
    model.train()
    model.zero_grad()
    yp = model(X, coords, t)
    loss = n3_loss(model, yp, y, lc)
    loss.backward()
    return yp.detach(), loss.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}


def n3_opt(model, X, coords, t, y, lc):
    o = cases.OPT
    opt = torch.optim.AdamW(model.parameters(), lr=o["lr"], weight_decay=o["weight_decay"],
                            betas=o["betas"], eps=o["eps"])
    ema = ModelEMA(model, decay=o["ema_decay"])
    losses = []
    model.train()
    for _ in range(o["steps"]):
        opt.zero_grad()
        loss = n3_loss(model, model(X, coords, t), y, lc)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), o["grad_clip"])
        opt.step()
        ema.update(model)
        losses.append(float(loss))
    return ({n: p.detach().clone() for n, p in model.named_parameters()},
            {n: v.detach().clone() for n, v in ema.shadow.items()}, np.array(losses, dtype=np.float64))


def gen_quantile_case(name):
    cfg, lc = cases.quantile_cfg(name)
    X, coords, t, y = (torch.from_numpy(a) for a in cases.make_inputs(cfg))
    d = (X.double(), coords.double(), t.double(), y.double())
    out = {}
    y32, l32, g32 = n3_run(build(cfg), X, coords, t, y, lc)
    y64, l64, g64 = n3_run(build(cfg).double(), *d, lc)
    out["y32"], out["y64"] = y32.numpy(), y64.numpy()
    out["loss32"], out["loss64"] = np.float32(l32.item()), np.float64(l64.item())
    # the pieces of the objective on the float64 predictions, from the reference's own functions
    out["check64"] = np.array([float(ref_train.quantile_loss(y64[:, i:i + 1], d[3], q))
                               for i, q in enumerate(lc["taus"])])
    out["nc1_64"] = np.float64(ref_train.non_crossing_penalty(y64, "mean", 1).item())
    out["nc2_64"] = np.float64(ref_train.non_crossing_penalty(y64, "mean", 2).item())
    out["crps64"] = np.float64(ref_train.compute_crps_multi_quantile(
        y64.numpy(), d[3].numpy(), lc["taus"]))
    out["crossing_rows"] = np.int64(((y64[:, :-1] - y64[:, 1:]) > 0).any(1).sum().item()
                                    if y64.shape[1] > 1 else 0)
    for k in g64:
        store(out, "g", k, g64[k].numpy(), g32[k].numpy(), cfg["seed"] + 7, False)
    p32, s32, ls32 = n3_opt(build(cfg), X, coords, t, y, lc)
    p64, s64, ls64 = n3_opt(build(cfg).double(), *d, lc)
    out["opt_losses32"], out["opt_losses64"] = ls32, ls64
    for k in p64:
        store(out, "p", k, p64[k].numpy(), p32[k].numpy(), cfg["seed"] + 7, False)
        store(out, "ema", k, s64[k].numpy(), s32[k].numpy(), cfg["seed"] + 7, False)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB  loss64={out['loss64']:.6f} "
          f"crossing_rows={int(out['crossing_rows'])}")


def gen_n3_known_answers():
    """Known answers of the loss / penalty / CRPS functions on fixed inputs: the vectors of the
    reference's own tests (tests/stnf/models/test_crps_eq_4_6.py, test_p_nc_delta_penalty.py) plus
    seeded random ones, evaluated by the reference's functions."""
    out = {}
    rs = np.random.RandomState(77)
    yp = rs.standard_normal((64, 5))
    y = rs.standard_normal((64, 1))
    out["ka_yp"], out["ka_y"] = yp, y
    for i, q in enumerate(cases.TAUS5):
        out[f"ka_check_{i}"] = np.float64(ref_train.check_loss_numpy(yp[:, i], y[:, 0], q))
        out[f"ka_qloss_{i}"] = np.float64(ref_train.quantile_loss(
            torch.from_numpy(yp[:, i:i + 1]), torch.from_numpy(y), q).item())
    out["ka_nc1"] = np.float64(ref_train.non_crossing_penalty(torch.from_numpy(yp), "mean", 1).item())
    out["ka_nc2_sum"] = np.float64(ref_train.non_crossing_penalty(torch.from_numpy(yp), "sum", 2).item())
    out["ka_crps"] = np.float64(ref_train.compute_crps_multi_quantile(yp, y, cases.TAUS5))
    out["ka_crps_w"] = np.float64(ref_train.compute_crps_multi_quantile(
        yp, y, cases.TAUS5, weights=np.array([1.0, 2.0, 3.0, 2.0, 1.0])))
    # test_crps_eq_4_6.py vectors
    yt = np.array([2.0, 3.0, 4.0, 5.0])
    pd = {0.05: yt - 1.0, 0.25: yt - 0.5, 0.5: yt.copy(), 0.75: yt + 0.5, 0.95: yt + 1.0}
    out["ka_crps_thesis"] = np.float64(ref_train.compute_crps(pd, yt))
    out["ka_crps_single"] = np.float64(ref_train.compute_crps({0.5: np.array([2.5])}, np.array([2.0])))
    # P_nc(delta): the worked example of test_p_nc_delta_penalty.py + random + tie rows
    delta = rs.standard_normal((5, 9)) * 0.5
    delta[2, 0] = 5.0                                    # J = 0 branch
    delta[3, 1:] = np.abs(delta[3, 1:]); delta[3, 0] = 0.0   # tie: S == d_k0 == 0
    delta[4, 3] = 0.0                                    # clamp boundary
    ps = [torch.nn.Parameter(torch.from_numpy(delta[k].copy())) for k in range(5)]
    P = ref_train.compute_p_nc_delta_penalty(ps)
    P.backward()
    out["ka_delta"] = delta
    out["ka_pnc"] = np.float64(P.item())
    out["ka_pnc_grad"] = np.stack([p.grad.numpy() if p.grad is not None else np.zeros(9) for p in ps])
    ex = [torch.nn.Parameter(torch.tensor([0.3, 0.1, 0.2, 0.3, 0.4], dtype=torch.float64)),
          torch.nn.Parameter(torch.tensor([2.0, 1.0, -0.5, 0.3, -0.2], dtype=torch.float64)),
          torch.nn.Parameter(torch.tensor([0.1, 1.0, -0.5, 0.3, -0.2], dtype=torch.float64))]
    out["ka_pnc_example"] = np.float64(ref_train.compute_p_nc_delta_penalty(ex).item())
    np.savez_compressed(os.path.join(HERE, "n3_known_answers.npz"), **out)
    print("n3_known_answers.npz", len(out), "arrays; P_nc =", out["ka_pnc"])


# ---------------------------------------------------------------------------------------------
# N2: learnable knots, driven as the batch body does (train_st_interp.py:470-483,617-621,660-672,
# 693-712): MSE + domain/movement penalties, damping hook, per-group clipping, two-group AdamW, EMA
# ---------------------------------------------------------------------------------------------
def build_learn(cfg, kn):
    model = STInterpMLP(p=cfg["p"], k_spatial_centers=cfg["k_spatial_centers"],
                        k_temporal_centers=cfg["k_temporal_centers"],
                        hidden_dims=cfg["hidden_dims"], dropout=0.0, layernorm=cfg["layernorm"],
                        spatial_learnable=True, spatial_init_method="uniform",
                        spatial_basis_function=cfg["basis"], output_dim=cfg["output_dim"],
                        gradient_damping=kn.get("gradient_damping", False),
                        damping_threshold=kn.get("damping_threshold", 0.3),
                        damping_strength=kn.get("damping_strength", 1.0))
    sb = model.spatial_basis
    dc, dlb = cases.knot_perturbation(cfg)
    c0 = (sb.centers.detach().numpy() + dc).astype(np.float32)
    lb0 = (sb.log_bandwidths.detach().numpy() + dlb).astype(np.float32)
    st = cases.make_state(cfg)
    sd = model.state_dict()
    for k, v in st.items():
        assert tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v.copy())
    sd["spatial_basis.centers"] = torch.from_numpy(c0.copy())
    sd["spatial_basis.log_bandwidths"] = torch.from_numpy(lb0.copy())
    model.load_state_dict(sd)
    return model, c0, lb0


def learn_loss(model, X, coords, t, y, kn):
    loss = torch.nn.MSELoss()(model(X, coords, t), y)
    if kn.get("domain_penalty_weight", 0.0) > 0:
        loss = loss + kn["domain_penalty_weight"] * model.compute_domain_penalty()
    if kn.get("movement_penalty_weight", 0.0) > 0:
        loss = loss + kn["movement_penalty_weight"] * model.compute_movement_penalty()
    return loss


def learn_run(model, X, coords, t, y, kn):
    model.train()
    model.zero_grad()
    yp = model(X, coords, t)
    loss = learn_loss(model, X, coords, t, y, kn)
    loss.backward()
    return yp.detach(), loss.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}


def learn_opt(model, X, coords, t, y, kn):
    o = cases.OPT
    basis_params = list(model.spatial_basis.parameters())
    mlp_params = [p for p in model.parameters() if not any(p is bp for bp in basis_params)]
    opt = torch.optim.AdamW([{"params": mlp_params, "lr": o["lr"]},
                             {"params": basis_params, "lr": o["lr"] * cases.BASIS_LR_RATIO}],
                            weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"])
    ema = ModelEMA(model, decay=o["ema_decay"])
    losses = []
    model.train()
    for _ in range(o["steps"]):
        opt.zero_grad()
        loss = learn_loss(model, X, coords, t, y, kn)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(basis_params, o["grad_clip"] * cases.BASIS_CLIP_RATIO)
        torch.nn.utils.clip_grad_norm_(mlp_params, o["grad_clip"])
        opt.step()
        ema.update(model)
        losses.append(float(loss.detach()))
    return ({n: p.detach().clone() for n, p in model.named_parameters()},
            {n: v.detach().clone() for n, v in ema.shadow.items()}, np.array(losses, dtype=np.float64))


def gen_learn_case(name):
    cfg, kn = cases.learn_cfg(name)
    X, coords, t, y = (torch.from_numpy(a) for a in cases.make_inputs(cfg))
    d = (X.double(), coords.double(), t.double(), y.double())
    out = {}
    m32, c0, lb0 = build_learn(cfg, kn)
    out["in_centers"], out["in_log_bw"] = c0, lb0           # fp32 INPUTS of the case (knot state)
    out["in_centers_init"] = m32.spatial_basis.centers_init.numpy().copy()
    y32, l32, g32 = learn_run(m32, X, coords, t, y, kn)
    m64 = build_learn(cfg, kn)[0].double()
    y64, l64, g64 = learn_run(m64, *d, kn)
    out["y32"], out["y64"] = y32.numpy(), y64.numpy()
    out["loss32"], out["loss64"] = np.float32(l32.item()), np.float64(l64.item())
    out["domain_pen64"] = np.float64(m64.compute_domain_penalty().item())
    out["movement_pen64"] = np.float64(m64.compute_movement_penalty().item())
    for k in g64:
        store(out, "g", k, g64[k].numpy(), g32[k].numpy(), cfg["seed"] + 7, False)
    p32, s32, ls32 = learn_opt(build_learn(cfg, kn)[0], X, coords, t, y, kn)
    p64, s64, ls64 = learn_opt(build_learn(cfg, kn)[0].double(), *d, kn)
    out["opt_losses32"], out["opt_losses64"] = ls32, ls64
    for k in p64:
        store(out, "p", k, p64[k].numpy(), p32[k].numpy(), cfg["seed"] + 7, False)
        store(out, "ema", k, s64[k].numpy(), s32[k].numpy(), cfg["seed"] + 7, False)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    gc = out["gerr32_rell2/spatial_basis.centers"]
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB  loss64={out['loss64']:.6f} "
          f"domain={out['domain_pen64']:.4g} movement={out['movement_pen64']:.4g} ref fp32 err d_centers={gc:.2e}")


# ---------------------------------------------------------------------------------------------
# Sparsity penalties on the first layer, added to the loss as the batch body does
# (train_st_interp.py:674-691) with the reference's own compute_sparsity_penalty
# ---------------------------------------------------------------------------------------------
def sp_loss(model, X, coords, t, y, sp):
    loss = torch.nn.MSELoss()(model(X, coords, t), y)
    pen = model.compute_sparsity_penalty(penalty_type=sp["kind"], lambda_l1=sp["lambda_l1"],
                                         lambda_group=sp["lambda_group"])
    if sp.get("apply_spatial", True):
        loss = loss + pen["spatial_penalty"]
    if sp.get("apply_temporal", True):
        loss = loss + pen["temporal_penalty"]
    return loss, pen


def sp_build(cfg, zero_rows):
    model = build(cfg)
    sd = model.state_dict()
    for k, v in cases.sparsity_state(cfg, zero_rows).items():
        sd[k] = torch.from_numpy(v.copy())
    model.load_state_dict(sd)
    return model


def sp_run(model, X, coords, t, y, sp):
    model.train()
    model.zero_grad()
    loss, pen = sp_loss(model, X, coords, t, y, sp)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    return loss.detach(), {k: v.detach() for k, v in pen.items()}, grads


def sp_opt(model, X, coords, t, y, sp):
    o = cases.OPT
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=o["lr"],
                            weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"])
    ema = ModelEMA(model, decay=o["ema_decay"])
    losses = []
    model.train()
    for _ in range(o["steps"]):
        opt.zero_grad()
        loss, _ = sp_loss(model, X, coords, t, y, sp)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), o["grad_clip"])
        opt.step()
        ema.update(model)
        losses.append(float(loss))
    params = {n: p.detach().clone() for n, p in model.named_parameters()}
    shadow = {n: v.detach().clone() for n, v in ema.shadow.items()}
    return params, shadow, np.array(losses, dtype=np.float64)


def gen_sparsity_case(name):
    cfg, sp, zero_rows = cases.sparsity_cfg(name)
    X, coords, t, y = (torch.from_numpy(a) for a in cases.make_inputs(cfg))
    d = lambda a: a.double()
    out = {}
    l32, pen32, g32 = sp_run(sp_build(cfg, zero_rows), X, coords, t, y, sp)
    l64, pen64, g64 = sp_run(sp_build(cfg, zero_rows).double(), d(X), d(coords), d(t), d(y), sp)
    out["loss32"], out["loss64"] = np.float32(l32.item()), np.float64(l64.item())
    for k in ("spatial_penalty", "temporal_penalty"):
        out[k + "32"], out[k + "64"] = np.float32(pen32[k].item()), np.float64(pen64[k].item())
    for k in g64:
        store(out, "g", k, g64[k].numpy(), g32[k].numpy(), cfg["seed"] + 7, False)
    p32, s32, lo32 = sp_opt(sp_build(cfg, zero_rows), X, coords, t, y, sp)
    p64, s64, lo64 = sp_opt(sp_build(cfg, zero_rows).double(), d(X), d(coords), d(t), d(y), sp)
    out["opt_losses32"], out["opt_losses64"] = lo32, lo64
    for k in p64:
        store(out, "p", k, p64[k].numpy(), p32[k].numpy(), cfg["seed"] + 7, False)
        store(out, "ema", k, s64[k].numpy(), s32[k].numpy(), cfg["seed"] + 7, False)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB  loss64={out['loss64']:.6f} "
          f"pen=({out['spatial_penalty64']:.5f}, {out['temporal_penalty64']:.5f})")


def gen_init_known_answers():
    """Knot tables of the reference's gmm / random_site initialisers (st_interp.py:187-343) on fixed
    points with a fixed global numpy seed."""
    out = {}
    pts = cases.init_points()
    for method in ("gmm", "random_site"):
        np.random.seed(7)
        m = SpatialBasisEmbedding(n_centers=[9, 25], init_method=method, train_coords=pts)
        out[f"{method}_centers"] = m.centers.numpy().copy()
        out[f"{method}_bw"] = m.bandwidths.numpy().copy()
    np.random.seed(7)
    big = np.concatenate([pts] * 4)                       # > 10 000 rows: the GMM subsamples
    m = SpatialBasisEmbedding(n_centers=[16], init_method="gmm", train_coords=big)
    out["gmm_big_centers"], out["gmm_big_bw"] = m.centers.numpy().copy(), m.bandwidths.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "init_known_answers.npz"), **out)
    print("init_known_answers.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["knots", "model", "n3", "n2", "init", "sparsity"]
    if "init" in which:
        gen_init_known_answers()
    if "n2" in which:
        for nm in cases.LEARN_CASES:
            gen_learn_case(nm)
    if "knots" in which:
        gen_knots()
    if "model" in which:
        for nm in cases.MODEL_CASES:
            gen_case(nm)
    if "sparsity" in which:
        for name in cases.SPARSITY_CASES:
            gen_sparsity_case(name)
    if "n3" in which:
        import scripts.train_st_interp as ref_train  # noqa: E402  (the reference's loss functions)
        gen_n3_known_answers()
        for nm in cases.QUANTILE_CASES:
            gen_quantile_case(nm)
