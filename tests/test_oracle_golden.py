"""Pins the CPU oracle (oracle/stdadk_oracle.py, oracle/torch_port.py) against golden vectors
generated from the real reference (tests/golden/make_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest

from golden import cases
from oracle import stdadk_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


# ------------------------------------------------------------------ G1 knot tables (bit-exact)
@pytest.mark.parametrize("side", cases.KNOT_SIDES)
def test_knot_table_bit_exact(side):
    g = load("knots")
    lin = orc.linspace_f32(0.0, 1.0, side)
    assert lin.dtype == np.float32 and np.array_equal(lin.view(np.uint32), g[f"s{side}_lin"].view(np.uint32))
    c, bw, sides = orc.uniform_knots([side * side])
    assert sides == [side]
    assert np.array_equal(sha(c), g[f"s{side}_centers_sha"])
    assert np.array_equal(sha(bw), g[f"s{side}_bw_sha"])
    assert np.array_equal(bw[:1].view(np.uint32), g[f"s{side}_bw"].view(np.uint32))
    if side <= 11:
        assert np.array_equal(c, g[f"s{side}_centers"])
    # knot index k = ix*side + iy
    k = 1 * side + 2
    assert c[k, 0] == lin[1] and c[k, 1] == lin[2]


def test_knot_table_multilevel_and_temporal():
    g = load("knots")
    c, bw, sides = orc.uniform_knots([25, 81, 121])
    assert np.array_equal(c, g["ml_centers"]) and np.array_equal(bw, g["ml_bw"]) and sides == [5, 9, 11]
    for n in cases.TEMPORAL_NS:
        tc, tb = orc.temporal_knots([n])
        assert np.array_equal(tc, g[f"t{n}_centers"]) and np.array_equal(tb, g[f"t{n}_bw"])
    tc, tb = orc.temporal_knots([10, 15, 45])
    assert np.array_equal(tc, g["tml_centers"]) and np.array_equal(tb, g["tml_bw"])


def test_knot_errors():
    with pytest.raises(AssertionError):
        orc.uniform_knots([10])
    with pytest.raises(ValueError):
        orc.spatial_basis(np.zeros((1, 2)), np.zeros((1, 2)), np.ones(1), basis="nope")


# ------------------------------------------------------------------ G2-G5 float64 truth
def _digest_check(got, g, prefix, key, seed, rtol):
    got = np.asarray(got, dtype=np.float64)
    norm = float(g[f"{prefix}norm64/{key}"])
    if f"{prefix}64/{key}" in g:
        ref = g[f"{prefix}64/{key}"]
        assert got.shape == ref.shape
        assert np.linalg.norm((got - ref).ravel()) <= rtol * max(norm, 1e-30), (prefix, key)
    else:
        pos = cases.digest_positions(got.shape, 2048, seed)
        ref = g[f"{prefix}64s/{key}"]
        scale = max(np.abs(ref).max(), 1e-30)
        assert np.abs(got.ravel()[pos] - ref).max() <= rtol * scale * 10, (prefix, key)
        assert abs(np.linalg.norm(got.ravel()) - norm) <= rtol * norm
        if got.ndim == 2:
            cs, rsum = g[f"{prefix}colsum64/{key}"], g[f"{prefix}rowsum64/{key}"]
            # (sums over the LayerNorm-ed output dim cancel to ~0, so scale by the tensor norm)
            assert np.abs(got.sum(0) - cs).max() <= rtol * norm * 10
            assert np.abs(got.sum(1) - rsum).max() <= rtol * norm * 10


@pytest.mark.parametrize("name", list(cases.MODEL_CASES))
def test_oracle_matches_reference_float64(name):
    cfg = cases.MODEL_CASES[name]
    g = load(name)
    X, coords, t, y = cases.make_inputs(cfg)
    params = cases.make_state(cfg)
    yp, cache, phi, psi, feat = orc.model_forward(X, coords, t, params, cfg, np.float64)
    D = cfg["p"] + sum(cfg["k_spatial_centers"]) + sum(cfg["k_temporal_centers"])
    assert feat.shape == (cfg["B"], D)
    # G4 column order [X | phi | psi]
    p, Ks = cfg["p"], phi.shape[1]
    if p:
        assert np.array_equal(feat[:, :p], X.astype(np.float64))
    assert np.array_equal(feat[:, p:p + Ks], phi) and np.array_equal(feat[:, p + Ks:], psi)
    assert np.abs(psi - g["psi64"]).max() < 1e-14
    assert np.abs(phi.sum(1) - g["phi_rowsum64"]).max() < 1e-11
    if "phi64" in g:
        assert np.abs(phi - g["phi64"]).max() < 1e-12   # ref f64 uses cdist's expansion
    else:
        rc, val = g["phi64_nz_rc"], g["phi64_nz_val"]
        assert np.abs(phi[rc[:, 0], rc[:, 1]] - val).max() < 1e-12
    assert np.array_equal((phi != 0).sum(1).astype(np.int32), g["phi64_nnz_per_row"])
    assert np.abs(yp - g["y64"]).max() < 1e-11
    loss = orc.mse(yp, y)
    assert abs(loss - float(g["loss64"])) < 1e-12 * max(1.0, abs(loss))
    grads = orc.mlp_mse_backward(yp, y, cache, params, len(cfg["hidden_dims"]), cfg["layernorm"])
    for k in params:
        _digest_check(grads[k], g, "g", k, cfg["seed"] + 7, 1e-10)


@pytest.mark.parametrize("name", list(cases.MODEL_CASES))
def test_oracle_optimizer_steps(name):
    """G6: OPT['steps'] x (fwd, MSE, bwd, clip, AdamW, EMA) in float64."""
    cfg = cases.MODEL_CASES[name]
    g = load(name)
    o = cases.OPT
    X, coords, t, y = cases.make_inputs(cfg)
    params = {k: v.astype(np.float64) for k, v in cases.make_state(cfg).items()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v2 = {k: np.zeros_like(v) for k, v in params.items()}
    shadow = {k: v.copy() for k, v in params.items()}
    losses = []
    for s in range(1, o["steps"] + 1):
        yp, loss, grads = orc.train_step_grads(X, coords, t, y, params, cfg)
        losses.append(loss)
        orc.adamw_ema_step(params, grads, m, v2, shadow, s, o["lr"], o["weight_decay"], o["betas"],
                           o["eps"], o["grad_clip"], o["ema_decay"])
    assert np.abs(np.array(losses) - g["opt_losses64"]).max() < 1e-9
    for k in params:
        _digest_check(params[k], g, "p", k, cfg["seed"] + 7, 1e-9)
        _digest_check(shadow[k], g, "ema", k, cfg["seed"] + 7, 1e-9)


# ------------------------------------------------------------------ integer bookkeeping
@pytest.mark.parametrize("name", ["default227", "c2_b257"])
def test_knot_windows_cover_support(name):
    """The WINDOW x WINDOW block of knot_windows() contains every non-zero knot of the truth."""
    cfg = cases.MODEL_CASES[name]
    _, coords, _, _ = cases.make_inputs(cfg)
    centers, bw, sides = orc.uniform_knots(cfg["k_spatial_centers"])
    phi = orc.spatial_basis(coords, centers, bw, "wendland")
    ix0, iy0, col0, offs = orc.knot_windows(coords, sides, cfg["p"])
    covered = np.zeros_like(phi, dtype=bool)
    for l, side in enumerate(sides):
        w = min(orc.WINDOW, side)
        for b in range(coords.shape[0]):
            for dx in range(w):
                k0 = offs[l] + (ix0[b, l] + dx) * side + iy0[b, l]
                covered[b, k0:k0 + w] = True
        assert np.array_equal(col0[:, l], cfg["p"] + offs[l] + ix0[:, l] * side + iy0[:, l])
    assert not np.any((phi != 0) & ~covered)


# ------------------------------------------------------------------ torch port (timed baseline)
@pytest.mark.parametrize("name", ["tiny9", "tiny9_ln_p3", "default227", "default227_gauss",
                                  "default227_tri", "c2_b257_noln"])
def test_torch_port_matches_reference_fp32(name):
    import torch
    from oracle import torch_port as tp
    torch.set_num_threads(4)
    cfg = cases.MODEL_CASES[name]
    g = load(name)
    X, coords, t, y = (torch.from_numpy(a) for a in cases.make_inputs(cfg))
    model = tp.PortModel(cfg, state=cases.make_state(cfg))
    assert sorted(model.params) == sorted(cases.make_state(cfg))
    yp = model.forward(X, coords, t)
    loss = torch.nn.functional.mse_loss(yp, y)
    # same op sequence as the reference => agrees with the reference's own fp32 run to rounding
    scale = max(float(np.abs(g["y64"]).max()), 1.0)
    assert np.abs(yp.detach().numpy() - g["y32"]).max() <= 2e-6 * scale
    assert abs(float(loss.detach()) - float(g["loss32"])) <= 2e-6 * max(1.0, float(g["loss32"]))
    loss.backward()
    for k, pten in model.params.items():
        if f"g64/{k}" in g:
            ref = g[f"g64/{k}"]
            err = np.abs(pten.grad.numpy() - ref).max()
            assert err <= max(2 * float(g[f"gerr32_maxabs/{k}"]), 1e-6 * np.abs(ref).max() + 1e-9), k


# ------------------------------------------------------------------ N3 quantile objectives, delta head, CRPS
def test_n3_known_answers():
    """check loss / non-crossing / CRPS / P_nc(delta) against values the reference's own functions
    produced (incl. the vectors of its tests test_crps_eq_4_6.py, test_p_nc_delta_penalty.py)."""
    g = load("n3_known_answers")
    yp, y = g["ka_yp"], g["ka_y"]
    for i, q in enumerate(cases.TAUS5):
        assert abs(orc.check_loss(yp[:, i:i + 1], y, q) - float(g[f"ka_check_{i}"])) < 1e-15
        assert abs(orc.check_loss(yp[:, i:i + 1], y, q) - float(g[f"ka_qloss_{i}"])) < 1e-15
    assert abs(orc.non_crossing_penalty(yp, 1, "mean") - float(g["ka_nc1"])) < 1e-14
    assert abs(orc.non_crossing_penalty(yp, 2, "sum") - float(g["ka_nc2_sum"])) < 1e-12
    assert orc.non_crossing_penalty(yp[:, :1]) == 0.0
    assert abs(orc.crps(yp, y, cases.TAUS5) - float(g["ka_crps"])) < 1e-14
    assert abs(orc.crps(yp, y, cases.TAUS5, [1, 2, 3, 2, 1]) - float(g["ka_crps_w"])) < 1e-14
    yt = np.array([2.0, 3.0, 4.0, 5.0])
    preds = np.stack([yt - 1.0, yt - 0.5, yt, yt + 0.5, yt + 1.0], axis=1)
    assert abs(orc.crps(preds, yt, cases.TAUS5) - float(g["ka_crps_thesis"])) < 1e-15
    assert abs(orc.crps(np.array([[2.5]]), np.array([2.0]), [0.5]) - float(g["ka_crps_single"])) < 1e-15
    # reference test vectors: check loss 0.25 / 0.05 for e = 0.5 at tau 0.5 / 0.1
    assert abs(orc.check_loss(np.array([1.0, 2.0, 3.0]), np.array([1.5, 2.5, 3.5]), 0.5) - 0.25) < 1e-15
    assert abs(orc.check_loss(np.array([1.0, 2.0, 3.0]), np.array([1.5, 2.5, 3.5]), 0.1) - 0.05) < 1e-15
    P, gP = orc.p_nc_delta(g["ka_delta"])
    assert abs(P - float(g["ka_pnc"])) < 1e-14
    assert np.abs(gP - g["ka_pnc_grad"]).max() < 1e-15     # incl. the tie and clamp-boundary rows
    ex = np.array([[0.3, 0.1, 0.2, 0.3, 0.4], [2.0, 1.0, -0.5, 0.3, -0.2], [0.1, 1.0, -0.5, 0.3, -0.2]])
    Pe, _ = orc.p_nc_delta(ex)
    assert abs(Pe - float(g["ka_pnc_example"])) < 1e-15 and abs(Pe - (0.1 - 0.7)) < 1e-15
    assert orc.p_nc_delta(ex[:1])[0] == 0.0


def test_n3_objective_gradient_is_the_derivative():
    """Finite differences of quantile_objective / p_nc_delta away from the kinks."""
    rs = np.random.RandomState(5)
    yp, y = rs.standard_normal((9, 4)), rs.standard_normal((9, 1))
    taus = [0.1, 0.4, 0.6, 0.9]
    for pw in (1, 2):
        L0, dY = orc.quantile_objective(yp, y, taus, 0.7, pw)
        for (b, q) in [(0, 0), (3, 2), (8, 3)]:
            e = np.zeros_like(yp); e[b, q] = 1e-7
            L1, _ = orc.quantile_objective(yp + e, y, taus, 0.7, pw)
            assert abs((L1 - L0) / 1e-7 - dY[b, q]) < 1e-5
    d = rs.standard_normal((4, 6))
    P0, gP = orc.p_nc_delta(d)
    for (k, j) in [(1, 0), (2, 3), (3, 5), (0, 2)]:
        e = np.zeros_like(d); e[k, j] = 1e-7
        assert abs((orc.p_nc_delta(d + e)[0] - P0) / 1e-7 - gP[k, j]) < 1e-6
    dW, db = rs.standard_normal((4, 5)), rs.standard_normal(4)
    Wo, bo = orc.delta_head(d)
    # <delta_head(d), (dW,db)> == <d, delta_head_backward(dW,db)>  (adjoint)
    assert abs((Wo * dW).sum() + (bo * db).sum() - (d * orc.delta_head_backward(dW, db)).sum()) < 1e-12


@pytest.mark.parametrize("name", list(cases.QUANTILE_CASES))
def test_n3_oracle_matches_reference_float64(name):
    cfg, lc = cases.quantile_cfg(name)
    g = load(name)
    X, coords, t, y = cases.make_inputs(cfg)
    params = cases.make_state(cfg)
    yp, loss, grads = orc.quantile_step_grads(X, coords, t, y, params, cfg, lc)
    assert yp.shape == (cfg["B"], cfg["output_dim"])
    assert np.abs(yp - g["y64"]).max() < 1e-11
    assert abs(loss - float(g["loss64"])) < 1e-12
    for i, q in enumerate(lc["taus"]):
        assert abs(orc.check_loss(yp[:, i:i + 1], y, q) - g["check64"][i]) < 1e-12
    assert abs(orc.non_crossing_penalty(yp, 1) - float(g["nc1_64"])) < 1e-12
    assert abs(orc.non_crossing_penalty(yp, 2) - float(g["nc2_64"])) < 1e-12
    assert abs(orc.crps(yp, y, lc["taus"]) - float(g["crps64"])) < 1e-12
    assert set(grads) == set(params)
    for k in params:
        _digest_check(grads[k], g, "g", k, cfg["seed"] + 7, 1e-10)


@pytest.mark.parametrize("name", list(cases.QUANTILE_CASES))
def test_n3_oracle_optimizer_steps(name):
    cfg, lc = cases.quantile_cfg(name)
    g = load(name)
    o = cases.OPT
    X, coords, t, y = cases.make_inputs(cfg)
    params = {k: v.astype(np.float64) for k, v in cases.make_state(cfg).items()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v2 = {k: np.zeros_like(v) for k, v in params.items()}
    shadow = {k: v.copy() for k, v in params.items()}
    losses = []
    for s in range(1, o["steps"] + 1):
        _, loss, grads = orc.quantile_step_grads(X, coords, t, y, params, cfg, lc)
        losses.append(loss)
        orc.adamw_ema_step(params, grads, m, v2, shadow, s, o["lr"], o["weight_decay"], o["betas"],
                           o["eps"], o["grad_clip"], o["ema_decay"])
    assert np.abs(np.array(losses) - g["opt_losses64"]).max() < 1e-9
    for k in params:
        _digest_check(params[k], g, "p", k, cfg["seed"] + 7, 1e-8)
        _digest_check(shadow[k], g, "ema", k, cfg["seed"] + 7, 1e-8)


# ------------------------------------------------------------------ N2 learnable knots
def learn_params(name):
    cfg, kn = cases.learn_cfg(name)
    g = load(name)
    params = dict(cases.make_state(cfg))
    params["spatial_basis.centers"] = g["in_centers"]
    params["spatial_basis.log_bandwidths"] = g["in_log_bw"]
    return cfg, kn, g, params


def test_n2_knot_gradient_is_the_derivative():
    """knot_backward against finite differences of sum(G * phi) for each basis."""
    rs = np.random.RandomState(9)
    x = rs.uniform(0, 1, (40, 2))
    c = rs.uniform(0, 1, (7, 2))
    lb = np.log(rs.uniform(0.3, 0.6, 7))
    G = rs.standard_normal((40, 7))
    for basis in ("wendland", "gaussian", "triangular"):
        f = lambda cc, ll: (G * orc.spatial_basis(x, cc, np.exp(ll), basis)).sum()      # noqa: E731
        dc, dlb = orc.knot_backward(x, c, lb, G, basis)
        for k, a in [(0, 0), (3, 1), (6, 0)]:
            e = np.zeros_like(c); e[k, a] = 1e-6
            assert abs((f(c + e, lb) - f(c - e, lb)) / 2e-6 - dc[k, a]) < 1e-5 * max(1.0, abs(dc[k, a])), basis
        for k in (1, 5):
            e = np.zeros_like(lb); e[k] = 1e-6
            assert abs((f(c, lb + e) - f(c, lb - e)) / 2e-6 - dlb[k]) < 1e-5 * max(1.0, abs(dlb[k])), basis
    # an observation exactly on a knot contributes nothing to that knot's centre gradient
    x[0] = c[2]
    dc1, _ = orc.knot_backward(x, c, lb, G, "triangular")
    assert np.isfinite(dc1).all()


@pytest.mark.parametrize("name", list(cases.LEARN_CASES))
def test_n2_oracle_matches_reference_float64(name):
    cfg, kn, g, params = learn_params(name)
    X, coords, t, y = cases.make_inputs(cfg)
    # the knot state of the case = grid knots (bit-exact with the reference) + the case's perturbation
    c_grid, bw_grid, _ = orc.uniform_knots(cfg["k_spatial_centers"])
    assert np.array_equal(g["in_centers_init"], c_grid)
    dc, dlb = cases.knot_perturbation(cfg)
    assert np.array_equal(g["in_centers"], (c_grid + dc).astype(np.float32))
    assert np.abs(g["in_log_bw"] - (np.log(bw_grid.astype(np.float64)) + dlb)).max() < 1e-6
    yp, loss, grads = orc.learnable_step_grads(X, coords, t, y, params, cfg, kn, g["in_centers_init"])
    assert np.abs(yp - g["y64"]).max() < 1e-10
    assert abs(orc.domain_penalty(params["spatial_basis.centers"])[0] - float(g["domain_pen64"])) < 1e-12
    assert abs(orc.movement_penalty(params["spatial_basis.centers"], g["in_centers_init"])[0]
               - float(g["movement_pen64"])) < 1e-12
    assert abs(loss - float(g["loss64"])) < 1e-11
    assert set(grads) == set(params)
    for k in params:
        _digest_check(grads[k], g, "g", k, cfg["seed"] + 7, 1e-9)


@pytest.mark.parametrize("name", list(cases.LEARN_CASES))
def test_n2_oracle_optimizer_steps(name):
    cfg, kn, g, params = learn_params(name)
    o = cases.OPT
    X, coords, t, y = cases.make_inputs(cfg)
    params = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v2 = {k: np.zeros_like(v) for k, v in params.items()}
    shadow = {k: v.copy() for k, v in params.items()}
    basis_keys = ["spatial_basis.centers", "spatial_basis.log_bandwidths"]
    mlp_keys = [k for k in params if k not in basis_keys]
    groups = [(mlp_keys, o["lr"], o["grad_clip"]),
              (basis_keys, o["lr"] * cases.BASIS_LR_RATIO, o["grad_clip"] * cases.BASIS_CLIP_RATIO)]
    losses = []
    for s in range(1, o["steps"] + 1):
        _, loss, grads = orc.learnable_step_grads(X, coords, t, y, params, cfg, kn, g["in_centers_init"])
        losses.append(loss)
        orc.adamw_ema_groups(params, grads, m, v2, shadow, s, groups, o["weight_decay"], o["betas"], o["eps"],
                             o["ema_decay"])
    assert np.abs(np.array(losses) - g["opt_losses64"]).max() < 1e-8
    for k in params:
        _digest_check(params[k], g, "p", k, cfg["seed"] + 7, 1e-8)
        _digest_check(shadow[k], g, "ema", k, cfg["seed"] + 7, 1e-8)


# ------------------------------------------------------------------ sparsity penalties on the first layer
def _sparsity_step(X, coords, t, y, params, cfg, sp):
    """MSE + the applied penalties and their gradient (train_st_interp.py:617-621,674-693)."""
    yp, loss, grads = orc.train_step_grads(X, coords, t, y, params, cfg)
    k0 = next(iter(params))
    ps, pt, dW = orc.sparsity_penalty(params[k0], cfg["p"], sum(cfg["k_spatial_centers"]),
                                      sum(cfg["k_temporal_centers"]), sp["kind"], sp["lambda_l1"],
                                      sp["lambda_group"], sp.get("apply_spatial", True),
                                      sp.get("apply_temporal", True))
    loss += (ps if sp.get("apply_spatial", True) else 0.0) + (pt if sp.get("apply_temporal", True) else 0.0)
    grads = dict(grads)
    grads[k0] = grads[k0] + dW
    return loss, ps, pt, grads


@pytest.mark.parametrize("name", list(cases.SPARSITY_CASES))
def test_sparsity_oracle_matches_reference_float64(name):
    cfg, sp, zero_rows = cases.sparsity_cfg(name)
    g = load(name)
    X, coords, t, y = cases.make_inputs(cfg)
    params = {k: v.astype(np.float64) for k, v in cases.sparsity_state(cfg, zero_rows).items()}
    loss, ps, pt, grads = _sparsity_step(X, coords, t, y, params, cfg, sp)
    assert abs(ps - float(g["spatial_penalty64"])) < 1e-11 and abs(pt - float(g["temporal_penalty64"])) < 1e-11
    assert abs(loss - float(g["loss64"])) < 1e-10
    for k in params:
        _digest_check(grads[k], g, "g", k, cfg["seed"] + 7, 1e-9)


@pytest.mark.parametrize("name", list(cases.SPARSITY_CASES))
def test_sparsity_oracle_optimizer_steps(name):
    cfg, sp, zero_rows = cases.sparsity_cfg(name)
    g = load(name)
    o = cases.OPT
    X, coords, t, y = cases.make_inputs(cfg)
    params = {k: v.astype(np.float64) for k, v in cases.sparsity_state(cfg, zero_rows).items()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v2 = {k: np.zeros_like(v) for k, v in params.items()}
    shadow = {k: v.copy() for k, v in params.items()}
    losses = []
    for s in range(1, o["steps"] + 1):
        loss, _, _, grads = _sparsity_step(X, coords, t, y, params, cfg, sp)
        losses.append(loss)
        orc.adamw_ema_step(params, grads, m, v2, shadow, s, o["lr"], o["weight_decay"], o["betas"],
                           o["eps"], o["grad_clip"], o["ema_decay"])
    assert np.abs(np.array(losses) - g["opt_losses64"]).max() < 1e-8
    for k in params:
        _digest_check(params[k], g, "p", k, cfg["seed"] + 7, 1e-8)
        _digest_check(shadow[k], g, "ema", k, cfg["seed"] + 7, 1e-8)


def test_sparsity_oracle_rejects_unknown_kind():
    with pytest.raises(ValueError):
        orc.sparsity_penalty(np.zeros((2, 3)), 0, 2, 1, "lasso")
    assert orc.sparsity_penalty(np.ones((2, 3)), 0, 2, 1, "none")[:2] == (0.0, 0.0)
