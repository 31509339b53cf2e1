"""The product's own host path (plain torch inside stnf.models, selected for host tensors; `device: cpu` is the
reference's shipped default, configs/config_st_interp.yaml:85) against the SAME goldens the HIP path is pinned to:
y / loss / every gradient vs the float64 run of the real reference.  It never touches the oracle package or the
native library."""
import os

import numpy as np
import pytest
import torch

from golden import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def check_vs_digest(got, g, prefix, key, seed, tol=TOL):
    got = np.asarray(got, dtype=np.float64)
    norm = float(g[f"{prefix}norm64/{key}"])
    if f"{prefix}64/{key}" in g:
        ref = g[f"{prefix}64/{key}"]
        assert got.shape == ref.shape
        assert np.linalg.norm((got - ref).ravel()) / max(norm, 1e-30) <= tol, (prefix, key)
        return
    pos = cases.digest_positions(got.shape, 2048, seed)
    ref = g[f"{prefix}64s/{key}"]
    assert np.linalg.norm(got.ravel()[pos] - ref) / max(np.linalg.norm(ref), 1e-30) <= tol, (prefix, key)
    assert abs(np.linalg.norm(got.ravel()) - norm) <= tol * norm


def build(cfg, **kw):
    from stnf.models import STInterpMLP
    m = STInterpMLP(p=cfg["p"], k_spatial_centers=cfg["k_spatial_centers"],
                    k_temporal_centers=cfg["k_temporal_centers"], hidden_dims=cfg["hidden_dims"], dropout=0.0,
                    layernorm=cfg["layernorm"], spatial_basis_function=cfg["basis"], output_dim=cfg["output_dim"], **kw)
    return m


@pytest.mark.parametrize("name", list(cases.MODEL_CASES))
def test_host_forward_backward_matches_reference_goldens(name):
    cfg = cases.MODEL_CASES[name]
    g = load(name)
    m = build(cfg)
    st = cases.make_state(cfg)
    with torch.no_grad():
        for (_, p), (k, v) in zip(m.named_parameters(), st.items()):
            p.copy_(torch.from_numpy(v.copy()))
    m.train()
    X, coords, t, y = (torch.from_numpy(a) for a in cases.make_inputs(cfg))
    phi = m.spatial_basis(coords)
    if "phi64" in g:            # (the large cases keep row sums of phi only)
        assert np.abs(phi.detach().numpy() - g["phi64"]).max() <= TOL
    else:
        assert np.abs(phi.detach().double().sum(1).numpy() - g["phi_rowsum64"]).max() <= TOL * max(1.0, g["phi_rowsum64"].max())
    assert np.abs(m.temporal_basis(t).detach().numpy() - g["psi64"]).max() <= TOL
    yp = m(X, coords, t)
    loss = torch.nn.MSELoss()(yp, y)
    loss.backward()
    assert np.abs(yp.detach().numpy() - g["y64"]).max() <= TOL * max(1.0, np.abs(g["y64"]).max())
    assert abs(loss.item() - float(g["loss64"])) <= TOL * float(g["loss64"])
    for k, p in m.named_parameters():
        check_vs_digest(p.grad.numpy(), g, "g", k, cfg["seed"] + 7)
    # eval / no_grad gives the same numbers (dropout = 0), and (B, N, 2) coordinates are accepted like the reference's
    m.eval()
    with torch.no_grad():
        assert torch.allclose(m(X, coords, t), yp.detach(), rtol=0, atol=1e-6)
        assert torch.allclose(m.spatial_basis(coords.view(1, -1, 2)).view(phi.shape), phi.detach(), rtol=0, atol=1e-7)


@pytest.mark.parametrize("name", ["tiny9_delta5", "default227_delta5", "default227_mq5"])
def test_host_quantile_heads_match_reference_goldens(name):
    """Multi-quantile output and the delta-reparameterised head on the host path, through stnf.losses' check loss and
    penalties (the driver's functions), against the goldens made with the reference's own loss functions."""
    from stnf import losses as Ls
    cfg, lc = cases.quantile_cfg(name)
    g = load(name)
    m = build(cfg, use_delta_reparameterization=cfg["delta"])
    st = cases.make_state(cfg)
    assert [k for k, _ in m.named_parameters()] == list(st.keys())
    with torch.no_grad():
        for (_, p), (k, v) in zip(m.named_parameters(), st.items()):
            p.copy_(torch.from_numpy(v.copy()))
    m.train()
    X, coords, t, y = (torch.from_numpy(a) for a in cases.make_inputs(cfg))
    yp = m(X, coords, t)
    assert yp.shape == (cfg["B"], cfg["output_dim"])
    assert np.abs(yp.detach().numpy() - g["y64"]).max() <= TOL * max(1.0, np.abs(g["y64"]).max())


@pytest.mark.parametrize("name", ["tiny9_learn", "default227_learn", "default227_learn_gauss"])
def test_host_learnable_knots_match_reference_goldens(name):
    """Learnable knots on the host path: gradients flow into centres and log-bandwidths by plain autograd."""
    cfg, kn = cases.learn_cfg(name)
    g = load(name)
    m = build(cfg, spatial_learnable=True, gradient_damping=False)
    st = cases.make_state(cfg)
    with torch.no_grad():
        m.spatial_basis.centers.copy_(torch.from_numpy(g["in_centers"]))
        m.spatial_basis.log_bandwidths.copy_(torch.from_numpy(g["in_log_bw"]))
        for (k, p) in list(m.named_parameters())[2:]:
            p.copy_(torch.from_numpy(st[k].copy()))
    m.train()
    X, coords, t, y = (torch.from_numpy(a) for a in cases.make_inputs(cfg))
    yp = m(X, coords, t)
    assert np.abs(yp.detach().numpy() - g["y64"]).max() <= TOL * max(1.0, np.abs(g["y64"]).max())
    torch.nn.MSELoss()(yp, y).backward()
    assert m.spatial_basis.centers.grad is not None and m.spatial_basis.log_bandwidths.grad is not None
    assert float(m.spatial_basis.centers.grad.abs().sum()) > 0


def test_host_model_runs_the_reference_style_batch_body():
    """optimizer / clip / EMA exactly as the reference's driver strings them together (train_st_interp.py:608-721),
    on host tensors: the loss goes down and ModelEMA's in-place swap restores the training weights."""
    from stnf.models import create_model
    from stnf.utils import ModelEMA, set_seed
    set_seed(3)
    m = create_model(dict(k_spatial_centers=[25, 81], k_temporal_centers=[5, 9], hidden_dims=[32, 16], dropout=0.1))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2, weight_decay=1e-4)
    ema = ModelEMA(m, decay=0.9)
    g = torch.Generator().manual_seed(0)
    coords, t = torch.rand(512, 2, generator=g), torch.rand(512, 1, generator=g)
    y = torch.sin(4 * coords[:, :1]) + 0.05 * torch.randn(512, 1, generator=g)
    first = last = None
    for _ in range(60):
        opt.zero_grad()
        loss = torch.nn.MSELoss()(m(torch.zeros(512, 0), coords, t), y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 10.0)
        opt.step()
        ema.update(m)
        first = loss.item() if first is None else first
        last = loss.item()
    assert last < 0.5 * first
    before = [p.detach().clone() for p in m.parameters()]
    ema.apply_shadow()
    assert any(not torch.equal(a, p) for a, p in zip(before, m.parameters()))
    ema.restore()
    assert all(torch.equal(a, p) for a, p in zip(before, m.parameters()))


# ------------------------------------------------------------------ dense-grid predictions + predictions.npz
def test_predict_all_times_host_and_npz_record(tmp_path):
    """The driver's dense-grid loop (scripts/train_st_interp.py:1228-1248) and its predictions.npz
    (:2551-2560): time slices t_idx / (T - 1), the median column of a multi-quantile model, float64 (T, S);
    the record carries the reference's six keys with the arrays unchanged."""
    from stnf.models import STInterpMLP
    from stnf.utils.predictions import NPZ_KEYS, predict_all_times, save_predictions_npz
    torch.manual_seed(3)
    rs = np.random.RandomState(3)
    S, T = 37, 5
    coords = rs.uniform(0, 1, (S, 2)).astype(np.float32)
    for q in (1, 5):
        m = STInterpMLP(p=0, k_spatial_centers=[9, 25], k_temporal_centers=[4], hidden_dims=[32, 16], dropout=0.1,
                        layernorm=True, output_dim=q)
        m.train()
        got = predict_all_times(m, coords, T)
        assert m.training                                   # the mode is restored
        assert got.shape == (T, S) and got.dtype == np.float64
        m.eval()
        with torch.no_grad():
            for ti in range(T):
                ref = m(torch.zeros(S, 0), torch.from_numpy(coords), torch.full((S, 1), ti / (T - 1))).numpy()
                np.testing.assert_array_equal(got[ti], ref[:, q // 2].astype(np.float64))
        one = predict_all_times(m, coords, 1)               # T == 1: t = 0.0
        with torch.no_grad():
            ref0 = m(torch.zeros(S, 0), torch.from_numpy(coords), torch.zeros(S, 1)).numpy()
        np.testing.assert_array_equal(one[0], ref0[:, q // 2].astype(np.float64))
    z = rs.standard_normal((T, S))
    z[0, 3] = np.nan
    masks = [rs.uniform(size=(T, S)) < p for p in (0.6, 0.2, 0.2)]
    path = save_predictions_npz(tmp_path / "run", got, z, coords, *masks)
    with np.load(path) as f:
        assert tuple(f.files) == NPZ_KEYS
        np.testing.assert_array_equal(f["predictions"], got)
        np.testing.assert_array_equal(f["true"], z)
        np.testing.assert_array_equal(f["coords"], coords)
        for k, mk in zip(NPZ_KEYS[3:], masks):
            assert f[k].dtype == np.bool_
            np.testing.assert_array_equal(f[k], mk)
    with pytest.raises(ValueError):
        save_predictions_npz(tmp_path / "bad", got, z[:, :-1], coords, *masks)
    with pytest.raises(ValueError):
        predict_all_times(STInterpMLP(p=2, k_spatial_centers=[9], k_temporal_centers=[4], hidden_dims=[16]), coords, T)


def test_evaluate_model_metrics_host():
    """stnf.utils.predictions.evaluate_model against the formulas of scripts/train_st_interp.py:884-961 written
    out in numpy: mse / mae / rmse on the median column, the check loss of a single quantile, CRPS = 2 x the mean
    check loss over the levels (equation 4.6 with uniform weights) and its aliases."""
    from stnf.dataio.device_dataset import DeviceDataset
    from stnf.models import STInterpMLP
    from stnf.utils.predictions import evaluate_model
    torch.manual_seed(5)
    rs = np.random.RandomState(5)
    n = 211
    ds = DeviceDataset(torch.from_numpy(rs.uniform(0, 1, (n, 2)).astype(np.float32)),
                       torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)),
                       torch.from_numpy(rs.standard_normal((n, 1)).astype(np.float32)))
    taus = [0.05, 0.25, 0.5, 0.75, 0.95]

    def rho(p, y, q):
        e = y - p
        return np.mean(np.maximum((q - 1) * e, q * e))

    for q, cfg in ((1, None), (1, dict(regression_type="quantile", current_quantile=0.9)),
                   (5, dict(regression_type="multi-quantile", quantile_levels=taus))):
        m = STInterpMLP(p=0, k_spatial_centers=[9, 25], k_temporal_centers=[4], hidden_dims=[32, 16], dropout=0.1,
                        layernorm=True, output_dim=q)
        m.train()
        got = evaluate_model(m, ds, cfg)
        assert m.training
        m.eval()
        with torch.no_grad():
            pr = m(torch.zeros(n, 0), ds.coords, ds.t).numpy()
        y = ds.y.numpy()
        col = pr[:, 2:3] if q == 5 else pr
        mse = np.mean((col - y) ** 2)
        assert got["mse"] == pytest.approx(float(mse), rel=1e-6) and got["rmse"] == pytest.approx(float(np.sqrt(mse)), rel=1e-6)
        assert got["mae"] == pytest.approx(float(np.mean(np.abs(col - y))), rel=1e-6)
        if cfg is None:
            assert set(got) == {"mse", "mae", "rmse"}
        elif q == 1:
            assert got["check_loss"] == pytest.approx(float(rho(pr, y, 0.9)), rel=1e-6)
        else:
            cl = [rho(pr[:, i:i + 1], y, t) for i, t in enumerate(taus)]
            assert got["mean_check_loss"] == pytest.approx(float(np.mean(cl)), rel=1e-6)
            assert got["check_loss"] == got["mean_check_loss"]
            assert got["crps"] == pytest.approx(2.0 * float(np.mean(cl)), rel=1e-6)
