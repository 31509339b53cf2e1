"""GPU parity tests: the HIP path (through the C ABI, via the stnf package) against the oracle and
the golden vectors of the real reference.  Run with `pytest -m gpu` on an MI355X.

Tolerances (BASELINE.json north_star: 1e-5 relative fp32; bit-exact for index bookkeeping):
  phi / psi / features : max-abs error <= 1e-5 * max|.| (= 1e-5, the bases peak at 1) vs float64 truth
  y_pred               : max-abs error <= 1e-5 * max(1, max|y|)
  loss                 : relative error <= 1e-5
  gradients / params   : rel-L2 error <= 1e-5 per tensor (accumulation order differs from MKL)
and the HIP result must be at least as close to the float64 truth as the reference's own fp32 run
where that run is limited by cdist's matmul expansion.
"""
import os

import numpy as np
import pytest
import torch

from golden import cases
from oracle import stdadk_oracle as orc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def dev():
    assert torch.cuda.is_available(), "needs an MI355X"
    return torch.device("cuda:0")


def build_model(cfg, dropout=0.0):
    from stnf.models import STInterpMLP
    m = STInterpMLP(p=cfg["p"], k_spatial_centers=cfg["k_spatial_centers"],
                    k_temporal_centers=cfg["k_temporal_centers"], hidden_dims=cfg["hidden_dims"],
                    dropout=dropout, layernorm=cfg["layernorm"], spatial_basis_function=cfg["basis"],
                    output_dim=cfg["output_dim"])
    # assign in nn.Sequential order (Dropout layers shift the indices of the state_dict keys)
    st = cases.make_state(cfg)
    with torch.no_grad():
        for (_, p), (k, v) in zip(m.named_parameters(), st.items()):
            assert tuple(p.shape) == v.shape, k
            p.copy_(torch.from_numpy(v.copy()))
    m.force_window_path = True      # the small golden models exercise the window kernels too
    return m.to(dev())


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def check_vs_digest(got, g, prefix, key, seed, tol=TOL):
    got = np.asarray(got, dtype=np.float64)
    norm = float(g[f"{prefix}norm64/{key}"])
    if f"{prefix}64/{key}" in g:
        ref = g[f"{prefix}64/{key}"]
        assert got.shape == ref.shape
        err = np.linalg.norm((got - ref).ravel()) / max(norm, 1e-30)
        assert err <= tol, (prefix, key, err)
        return err
    pos = cases.digest_positions(got.shape, 2048, seed)
    ref = g[f"{prefix}64s/{key}"]
    err = np.linalg.norm(got.ravel()[pos] - ref) / max(np.linalg.norm(ref), 1e-30)
    assert err <= tol, (prefix, key, "samples", err)
    assert abs(np.linalg.norm(got.ravel()) - norm) <= tol * norm
    if got.ndim == 2:
        cs, rs = g[f"{prefix}colsum64/{key}"], g[f"{prefix}rowsum64/{key}"]
        # sums over a LayerNorm-ed dim cancel to ~0: scale by the tensor norm
        assert np.abs(got.sum(0) - cs).max() <= tol * norm * 4
        assert np.abs(got.sum(1) - rs).max() <= tol * norm * 4
    return err


# ------------------------------------------------------------------ feature builder (A2-A5)
@pytest.mark.parametrize("name", ["tiny9", "tiny9_ln_p3", "default227", "default227_gauss",
                                  "default227_tri", "c2_b257"])
def test_rbf_build_matches_truth(name):
    cfg = cases.MODEL_CASES[name]
    g = load(name)
    X, coords, t, y = cases.make_inputs(cfg)
    m = build_model(cfg)
    d = dev()
    Xd, cd, td = (torch.from_numpy(a).to(d) for a in (X, coords, t))
    # knot buffers as the product built them (torch.linspace on this host)
    cen = m.spatial_basis.centers.cpu().numpy()
    bw = m.spatial_basis.bandwidths.cpu().numpy()
    tc, tb = m.temporal_basis.centers.cpu().numpy(), m.temporal_basis.bandwidths.cpu().numpy()
    phi_o = orc.spatial_basis(coords, cen, bw, cfg["basis"])
    psi_o = orc.temporal_basis(t, tc, tb)

    phi = m.spatial_basis(cd).cpu().numpy()            # (B, Ks) contiguous => ld = Ks
    psi = m.temporal_basis(td).cpu().numpy()
    assert phi.shape == phi_o.shape and psi.shape == psi_o.shape
    assert np.abs(phi - phi_o).max() <= TOL
    assert np.abs(psi - psi_o).max() <= TOL
    # golden truth of the reference itself (knots generated in the build container)
    assert np.abs(psi - g["psi64"]).max() <= TOL
    assert np.abs(phi.astype(np.float64).sum(1) - g["phi_rowsum64"]).max() <= TOL * phi.shape[1] ** 0.5 * 4
    if "phi64" in g:
        assert np.abs(phi - g["phi64"]).max() <= TOL
    else:
        rc, val = g["phi64_nz_rc"], g["phi64_nz_val"]
        assert np.abs(phi[rc[:, 0], rc[:, 1]] - val).max() <= TOL
    # the HIP direct-form distance beats the reference's own fp32 run (cdist expansion) at K > 25
    if float(g["phi_err32_maxabs"]) > 1e-6:
        assert np.abs(phi - phi_o).max() <= float(g["phi_err32_maxabs"])
    # exact zeros outside the support, exact one on a knot (row 0 = (0,0) sits on knot 0)
    if cfg["basis"] == "wendland":
        assert np.array_equal(phi == 0, phi_o == 0) or np.abs(phi[(phi == 0) != (phi_o == 0)]).max() < 1e-12
        assert phi[0, 0] == 1.0

    # A5: padded feature buffer, column order [X | phi | psi], zero padding
    feats = m.build_features(Xd, cd, td)
    D = cfg["p"] + phi.shape[1] + psi.shape[1]
    assert feats.shape[1] % 32 == 0 and feats.shape[1] >= D
    f = feats.cpu().numpy()
    p = cfg["p"]
    if p:
        assert np.array_equal(f[:, :p], X)
    assert np.array_equal(f[:, p:p + phi.shape[1]], phi)
    assert np.array_equal(f[:, p + phi.shape[1]:D], psi)
    assert not f[:, D:].any()


def test_rbf_build_unaligned_and_ragged():
    """ld not a multiple of 4, odd B, B == 1, empty batch."""
    from stnf import _native as N
    d = dev()
    cen, bw, _ = orc.uniform_knots([25, 81, 121])
    tc, tb = orc.temporal_knots([10, 15, 45])
    cend, bwd, tcd, tbd = (torch.from_numpy(a).to(d) for a in (cen, bw, tc, tb))
    rs = np.random.RandomState(5)
    for B in (1, 3, 67, 1025):
        coords = rs.uniform(-0.1, 1.1, (B, 2)).astype(np.float32)
        t = rs.uniform(0, 1, (B, 1)).astype(np.float32)
        X = rs.standard_normal((B, 5)).astype(np.float32)
        ref = np.concatenate([X, orc.spatial_basis(coords, cen, bw), orc.temporal_basis(t, tc, tb)], 1)
        for ld in (ref.shape[1], ref.shape[1] + 1, ref.shape[1] + 3):
            buf = torch.full((B, ld), 7.0, device=d)
            N.rbf_build(torch.from_numpy(coords).to(d), torch.from_numpy(t).to(d).view(-1),
                        torch.from_numpy(X).to(d), cend, bwd, "wendland", tcd, tbd, buf)
            got = buf.cpu().numpy()
            assert np.abs(got[:, :ref.shape[1]] - ref).max() <= TOL
            assert not got[:, ref.shape[1]:].any()
    empty = torch.empty(0, 302, device=d)
    N.rbf_build(torch.empty(0, 2, device=d), torch.empty(0, device=d), torch.empty(0, 5, device=d),
                cend, bwd, "wendland", tcd, tbd, empty)


# ------------------------------------------------------------------ forward / loss / backward
@pytest.mark.parametrize("path", ["auto", "dense"])
@pytest.mark.parametrize("name", list(cases.MODEL_CASES))
def test_forward_backward_matches_reference(name, path):
    """path 'auto' takes the index-window kernels where they apply (Wendland/triangular, uniform
    grid, hidden[0] in {128,256}) and the dense kernels elsewhere; 'dense' forces materialisation."""
    from stnf import _native as N
    cfg = cases.MODEL_CASES[name]
    g = load(name)
    X, coords, t, y = cases.make_inputs(cfg)
    d = dev()
    m = build_model(cfg)
    m.force_dense_path = path == "dense"
    st = m._step_state(d, force_dense=m.force_dense_path)
    expect_window = (path == "auto" and cfg["basis"] in ("wendland", "triangular")
                     and cfg["hidden_dims"][0] in (128, 256))
    assert N.step_uses_window(st.basis, st.desc, st.flags) == expect_window
    m.train()
    Xd, cd, td, yd = (torch.from_numpy(a).to(d) for a in (X, coords, t, y))
    yp = m(Xd, cd, td)
    assert yp.shape == (cfg["B"], cfg["output_dim"])
    loss = torch.nn.MSELoss()(yp, yd)
    loss.backward()
    y_hip = yp.detach().cpu().numpy()
    scale = max(1.0, float(np.abs(g["y64"]).max()))
    err_y = np.abs(y_hip - g["y64"]).max()
    assert err_y <= TOL * scale, err_y
    assert abs(loss.item() - float(g["loss64"])) <= TOL * max(float(g["loss64"]), 1e-12)
    # at least as close to the truth as the reference's fp32 run when that run is cdist-limited
    if float(g["y_err32_maxabs"]) > 1e-5:
        assert err_y <= float(g["y_err32_maxabs"])
    for k, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, k
        check_vs_digest(p.grad.cpu().numpy(), g, "g", k, cfg["seed"] + 7)
    # eval / no_grad path gives the same numbers (dropout = 0)
    m.eval()
    with torch.no_grad():
        y2 = m(Xd, cd, td)
    assert torch.equal(y2, yp.detach())


def test_forward_determinism_and_shapes():
    """Mirrors the reference's structural tests (tests/stnf/models/
    test_st_interp_delta_reparameterization.py:43-55,141-151): output shape (B, output_dim),
    attributes, two eval forwards identical."""
    from stnf.models import STInterpMLP
    d = dev()
    m = STInterpMLP(p=0, k_spatial_centers=[9], k_temporal_centers=[5], hidden_dims=[32, 16],
                    dropout=0.0, layernorm=False, output_dim=5).to(d)
    m.eval()
    X, c, t = torch.zeros(4, 0, device=d), torch.rand(4, 2, device=d), torch.rand(4, 1, device=d)
    with torch.no_grad():
        a, b = m(X, c, t), m(X, c, t)
    assert a.shape == (4, 5) and torch.equal(a, b) and torch.isfinite(a).all()
    assert hasattr(m, "mlp") and m.mlp_trunk is None and m.delta_params is None
    assert m.get_delta_parameters() is None
    # gradient flows to every parameter
    m.train()
    m(X, c, t).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_wide_output_head_uses_gemm():
    """output_dim > 8 takes the GEMM path of the output layer; compare with the oracle."""
    cfg = dict(cases.MODEL_CASES["default227"], output_dim=11, B=77, seed=31)
    X, coords, t, _ = cases.make_inputs(cfg)
    d = dev()
    m = build_model(cfg)
    params = cases.make_state(cfg)
    rs = np.random.RandomState(3)
    y = rs.standard_normal((cfg["B"], 11)).astype(np.float32)
    yp = m(*(torch.from_numpy(a).to(d) for a in (X, coords, t)))
    torch.nn.MSELoss()(yp, torch.from_numpy(y).to(d)).backward()
    cen = m.spatial_basis.centers.cpu().numpy(); bw = m.spatial_basis.bandwidths.cpu().numpy()
    tc = m.temporal_basis.centers.cpu().numpy(); tb = m.temporal_basis.bandwidths.cpu().numpy()
    feat = orc.features(X, orc.spatial_basis(coords, cen, bw), orc.temporal_basis(t, tc, tb), 0)
    yo, cache = orc.mlp_forward(feat, params, 3, True)
    go = orc.mlp_mse_backward(yo, y, cache, params, 3, True)
    assert np.abs(yp.detach().cpu().numpy() - yo).max() <= TOL * max(1.0, np.abs(yo).max())
    for k, p in m.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), go[k]) <= TOL, k


# ------------------------------------------------------------------ dropout (explicit masks)
def test_dropout_masks_forward_backward():
    from stnf import _native as N
    cfg = dict(cases.MODEL_CASES["default227"], B=96, seed=41)
    X, coords, t, y = cases.make_inputs(cfg)
    d = dev()
    m = build_model(cfg, dropout=0.25)
    params = cases.make_state(cfg)
    rs = np.random.RandomState(9)
    masks = [(rs.uniform(size=(cfg["B"], h)) >= 0.25).astype(np.uint8) for h in cfg["hidden_dims"]]
    feats = m.build_features(*(torch.from_numpy(a).to(d) for a in (X, coords, t)))
    desc = m._native_desc()
    B = cfg["B"]
    ws = torch.empty(N.mlp_workspace_bytes(desc, B) // 4, device=d)
    yp = torch.empty(B, 1, device=d)
    md = [torch.from_numpy(a).to(d) for a in masks]
    N.mlp_forward(desc, m._native_tensors(), feats, B, yp, ws, True, 123, md)
    dY = torch.empty(B, 1, device=d)
    lsum = torch.zeros(1, device=d)
    N.mse(yp, torch.from_numpy(y).to(d), 1.0 / B, dY, lsum)
    grads = [torch.empty_like(p) for p in m._param_list()]
    N.mlp_backward(desc, m._native_tensors(), m._pack(grads), feats, B, dY, ws, 123, md)
    cen = m.spatial_basis.centers.cpu().numpy(); bw = m.spatial_basis.bandwidths.cpu().numpy()
    tc = m.temporal_basis.centers.cpu().numpy(); tb = m.temporal_basis.bandwidths.cpu().numpy()
    feat = orc.features(X, orc.spatial_basis(coords, cen, bw), orc.temporal_basis(t, tc, tb), 0)
    yo, cache = orc.mlp_forward(feat, params, 3, True, drop_masks=masks, drop_p=0.25)
    go = orc.mlp_mse_backward(yo, y, cache, params, 3, True)
    assert np.abs(yp.cpu().numpy() - yo).max() <= TOL * max(1.0, np.abs(yo).max())
    assert abs(lsum.item() / B - orc.mse(yo, y)) <= TOL * orc.mse(yo, y)
    for k, gt in zip(params, grads):
        assert rel_l2(gt.cpu().numpy(), go[k]) <= TOL, k
    # generated masks: keep-rate ~ 1-p, train != eval, same seed reproduces
    m.train()
    a1 = torch.empty(B, 1, device=d); a2 = torch.empty(B, 1, device=d); a3 = torch.empty(B, 1, device=d)
    N.mlp_forward(desc, m._native_tensors(), feats, B, a1, ws, True, 77)
    N.mlp_forward(desc, m._native_tensors(), feats, B, a2, ws, True, 77)
    N.mlp_forward(desc, m._native_tensors(), feats, B, a3, ws, True, 78)
    assert torch.equal(a1, a2) and not torch.equal(a1, a3)


# ------------------------------------------------------------------ optimiser (A9 / G6)
@pytest.mark.parametrize("name", ["tiny9_ln_p3", "default227", "c2_b257"])
def test_optimizer_steps_match_reference(name):
    """OPT['steps'] x (fwd, MSE, bwd, clip, AdamW, EMA) with the native sumsq/adamw_ema kernels on
    per-parameter tensors, against the float64 golden of torch.optim.AdamW + clip_grad_norm_ +
    the reference's ModelEMA."""
    from stnf import _native as N
    cfg = cases.MODEL_CASES[name]
    g = load(name)
    o = cases.OPT
    d = dev()
    m = build_model(cfg)
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    plist = list(m.named_parameters())
    ms = [torch.zeros_like(p) for _, p in plist]
    vs = [torch.zeros_like(p) for _, p in plist]
    ema = [p.detach().clone() for _, p in plist]
    losses = []
    for step in range(1, o["steps"] + 1):
        for _, p in plist:
            p.grad = None
        loss = torch.nn.MSELoss()(m(X, coords, t), y)
        loss.backward()
        losses.append(loss.item())
        ss = torch.empty(len(plist) * N.SUMSQ_PARTS, device=d)
        for i, (_, p) in enumerate(plist):
            N.sumsq(p.grad.contiguous(), ss[i * N.SUMSQ_PARTS:(i + 1) * N.SUMSQ_PARTS])
        for (_, p), mm, vv, ee in zip(plist, ms, vs, ema):
            N.adamw_ema(p.data, p.grad.contiguous(), mm, vv, ee, o["lr"], o["betas"], o["eps"],
                        o["weight_decay"], step, max_norm=o["grad_clip"], sumsq_parts=ss,
                        ema_decay=o["ema_decay"])
    assert np.abs(np.array(losses) - g["opt_losses64"]).max() <= 5 * TOL * max(1.0, g["opt_losses64"].max())
    for (k, p), ee in zip(plist, ema):
        check_vs_digest(p.detach().cpu().numpy(), g, "p", k, cfg["seed"] + 7, tol=2e-5)
        check_vs_digest(ee.cpu().numpy(), g, "ema", k, cfg["seed"] + 7, tol=2e-5)


# ------------------------------------------------------------------ bench-size batch vs oracle
def test_c2_full_batch_against_oracle():
    """B = 4096 rows of the BASELINE C2 model: whole train step against the float64 oracle."""
    cfg = dict(cases.MODEL_CASES["c2_b257"], B=4096, seed=99)
    X, coords, t, y = cases.make_inputs(cfg)
    d = dev()
    m = build_model(cfg)
    m.train()
    params = cases.make_state(cfg)
    yp = m(*(torch.from_numpy(a).to(d) for a in (X, coords, t)))
    loss = torch.nn.MSELoss()(yp, torch.from_numpy(y).to(d))
    loss.backward()
    cen = m.spatial_basis.centers.cpu().numpy(); bw = m.spatial_basis.bandwidths.cpu().numpy()
    tc = m.temporal_basis.centers.cpu().numpy(); tb = m.temporal_basis.bandwidths.cpu().numpy()
    feat = orc.features(X, orc.spatial_basis(coords, cen, bw), orc.temporal_basis(t, tc, tb), 0)
    yo, cache = orc.mlp_forward(feat, params, 3, True)
    go = orc.mlp_mse_backward(yo, y, cache, params, 3, True)
    assert np.abs(yp.detach().cpu().numpy() - yo).max() <= TOL * max(1.0, np.abs(yo).max())
    assert abs(loss.item() - orc.mse(yo, y)) <= TOL * orc.mse(yo, y)
    for k, p in m.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), go[k]) <= TOL, k


def test_errors_are_loud():
    from stnf import _native as N
    from stnf.models import STInterpMLP
    d = dev()
    m = STInterpMLP(k_spatial_centers=[9], k_temporal_centers=[5], hidden_dims=[32, 16], dropout=0.0).to(d)
    with pytest.raises(RuntimeError):
        m(torch.zeros(4, 0), torch.rand(4, 2), torch.rand(4, 1))      # CPU tensors
    with pytest.raises(RuntimeError):
        N.rbf_build(torch.rand(4, 2, device=d), None, None, m.spatial_basis.centers,
                    m.spatial_basis.bandwidths, "wendland", None, None, torch.empty(4, 5, device=d))


# ------------------------------------------------------------------ the MFMA GEMM itself
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("shape", [(1, 1, 1), (33, 65, 17), (257, 256, 297), (64, 10374, 300),
                                   (300, 128, 10374), (4096, 256, 256), (256, 297, 4099)])
def test_gemm_f32_layouts(layout, shape):
    """C = op(A) op(B) + bias for the three operand layouts the MLP uses, ragged / unaligned
    shapes included, against float64 numpy.  Asymmetric random operands (a symmetric operand would
    hide a transposed C write)."""
    from stnf import _native as N
    M, Nn, K = shape
    d = dev()
    rs = np.random.RandomState(M * 7 + Nn * 3 + K)
    Aop = rs.standard_normal((M, K)).astype(np.float32)
    Bop = rs.standard_normal((K, Nn)).astype(np.float32)
    bias = rs.standard_normal(Nn).astype(np.float32)
    ref = Aop.astype(np.float64) @ Bop.astype(np.float64) + bias
    a_km = layout == "tn"
    b_km = layout in ("nn", "tn")
    A = torch.from_numpy(np.ascontiguousarray(Aop.T if a_km else Aop)).to(d)
    Bm = torch.from_numpy(np.ascontiguousarray(Bop if b_km else Bop.T)).to(d)
    out = N.gemm(A, a_km, Bm, b_km, M, Nn, K, bias=torch.from_numpy(bias).to(d))
    got = out.cpu().numpy()
    # fp32 fma chain: error ~ 1e-7 * sum|a b| ~ 1e-7 * sqrt(K) here
    assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.sqrt(K)) * 4
    assert rel_l2(got, ref) <= 2e-6


# ------------------------------------------------------------------ window path: integer bookkeeping
@pytest.mark.parametrize("B,G", [(1, 8), (257, 16), (4096, 64), (5000, 256)])
def test_bin_obs_bit_exact(B, G):
    """cell keys, cell_start and the sorted permutation are integers: bit-exact against the oracle."""
    from stnf import _native as N
    rs = np.random.RandomState(B + G)
    coords = rs.uniform(-0.05, 1.05, (B, 2)).astype(np.float32)
    coords[: min(B, 4)] = np.array([[0, 0], [1, 1], [0.5, 0.5], [np.nan, 2.0]], np.float32)[: min(B, 4)]
    if B > 600:
        coords[100:600] = coords[100]            # a crowded cell: ordering inside a cell is by index
    keys, cell_start, perm = N.bin_obs(torch.from_numpy(coords).to(dev()), G)
    ko = orc.cell_keys(coords, G)
    assert np.array_equal(keys.cpu().numpy(), ko)
    counts = np.bincount(ko, minlength=G * G)
    assert np.array_equal(cell_start.cpu().numpy(), np.concatenate([[0], np.cumsum(counts)]).astype(np.int32))
    assert np.array_equal(perm.cpu().numpy(), np.argsort(ko, kind="stable").astype(np.int32))


@pytest.mark.parametrize("sides,p", [([5, 9, 11], 0), ([32, 64, 72], 3), ([3], 0), ([1, 2, 6, 7], 1),
                                     ([32, 64, 128, 168], 0)])
def test_knot_windows_bit_exact(sides, p):
    from stnf import _native as N
    rs = np.random.RandomState(sum(sides))
    coords = rs.uniform(-0.2, 1.2, (3000, 2)).astype(np.float32)
    coords[:6] = np.array([[0, 0], [1, 1], [0.5, 0.5], [np.nan, 0.3], [np.inf, -np.inf], [1e-8, 1 - 1e-7]],
                          np.float32)
    ix0, iy0, col0 = N.knot_windows(torch.from_numpy(coords).to(dev()), sides, p)
    ox, oy, oc, _ = orc.knot_windows(coords, sides, p)
    assert np.array_equal(ix0.cpu().numpy(), ox)
    assert np.array_equal(iy0.cpu().numpy(), oy)
    assert np.array_equal(col0.cpu().numpy(), oc)


# ------------------------------------------------------------------ window path vs dense path
@pytest.mark.parametrize("name", ["default227", "default227_tri", "c2_b257", "c2_b257_noln"])
def test_window_and_dense_paths_agree(name):
    """Same model, both kernel families: predictions and every gradient agree to rounding, and
    y_pred comes back in the caller's row order."""
    cfg = cases.MODEL_CASES[name]
    X, coords, t, y = cases.make_inputs(cfg)
    d = dev()
    outs = []
    for dense in (False, True):
        m = build_model(cfg)
        m.force_dense_path = dense
        m.train()
        yp = m(*(torch.from_numpy(a).to(d) for a in (X, coords, t)))
        torch.nn.MSELoss()(yp, torch.from_numpy(y).to(d)).backward()
        outs.append((yp.detach().cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}))
    (yw, gw), (yd, gd) = outs
    assert np.abs(yw - yd).max() <= 2e-6 * max(1.0, np.abs(yd).max())
    for k in gd:
        assert rel_l2(gw[k], gd[k]) <= 2e-6, k


def test_window_path_with_covariates_and_h128():
    """p > 0 rows of W0^T and the 128-wide first layer, against the float64 oracle."""
    cfg = dict(p=3, k_spatial_centers=[25, 81, 121], k_temporal_centers=[10, 15, 45],
               hidden_dims=[128, 64], layernorm=True, basis="wendland", output_dim=1, B=300, seed=51)
    from stnf import _native as N
    X, coords, t, y = cases.make_inputs(cfg)
    d = dev()
    m = build_model(cfg)
    st = m._step_state(d)
    assert N.step_uses_window(st.basis, st.desc, st.flags)
    params = cases.make_state(cfg)
    m.train()
    yp = m(*(torch.from_numpy(a).to(d) for a in (X, coords, t)))
    loss = torch.nn.MSELoss()(yp, torch.from_numpy(y).to(d))
    loss.backward()
    yo, lo, go = orc.train_step_grads(X, coords, t, y, params, cfg)
    assert np.abs(yp.detach().cpu().numpy() - yo).max() <= TOL * max(1.0, np.abs(yo).max())
    assert abs(loss.item() - lo) <= TOL * lo
    for k, p in m.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), go[k]) <= TOL, k


def test_window_full_batch_sizes():
    """B = 4096 and 20000 rows of the C2 model on the window path: loss and the gradient's column
    sums (one per knot: which knots an observation touches) against the float64 oracle."""
    for B, seed in ((4096, 99), (20000, 98)):
        cfg = dict(cases.MODEL_CASES["c2_b257"], B=B, seed=seed)
        X, coords, t, y = cases.make_inputs(cfg)
        d = dev()
        m = build_model(cfg)
        m.train()
        params = cases.make_state(cfg)
        yp = m(*(torch.from_numpy(a).to(d) for a in (X, coords, t)))
        loss = torch.nn.MSELoss()(yp, torch.from_numpy(y).to(d))
        loss.backward()
        # 20 000 rows x 640 hidden units: a unit whose ReLU input the float64 run puts within 1e-6 of the kink can
        # fall on the other side in ANY fp32 evaluation (y moves by <= 1e-6, that unit's gradient contribution by
        # its full size: ~1e-5 of the gradient norms here; seed 98 has one within 1e-7).  Either side is accepted
        # for exactly those units (fitted from the residual), nothing else.
        yo, lo, go, alts = orc.train_step_grads(X, coords, t, y, params, cfg, kink_tol=1e-6)
        assert np.abs(yp.detach().cpu().numpy() - yo).max() <= TOL * max(1.0, np.abs(yo).max())
        assert abs(loss.item() - lo) <= TOL * lo
        got = {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}
        flipped, adj = orc.fit_kink_sides(got, go, alts)
        worst = max(rel_l2(got[k], adj[k]) for k in got)
        print(f"B={B}: {len(alts)} hidden units within 1e-6 of a ReLU kink, on the other side in fp32: {flipped}; "
              f"worst gradient rel-L2 {worst:.2e} (float64 sides: {max(rel_l2(got[k], go[k]) for k in got):.2e})")
        assert worst <= TOL, (worst, flipped)


# ------------------------------------------------------------------ fused engine (TrainStep / Predictor)
@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("name", ["default227", "c2_b257"])
def test_engine_steps_match_reference(name, dense):
    """OPT['steps'] fused steps (window or dense kernels, flat AdamW+clip+EMA, transposed W0
    storage) against the float64 golden of the reference's batch body; the nn.Module's parameters
    remain valid views (state_dict shapes unchanged)."""
    from stnf.engine import TrainStep
    cfg = cases.MODEL_CASES[name]
    g = load(name)
    o = cases.OPT
    d = dev()
    m = build_model(cfg)
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, lr=o["lr"], weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"],
                    grad_clip=o["grad_clip"], ema_decay=o["ema_decay"], max_batch=cfg["B"], force_dense=dense)
    assert eng.uses_window == (not dense)
    losses = []
    for _ in range(o["steps"]):
        eng.step(None, coords, t, y)
        losses.append(eng.mean_loss())
    assert np.abs(np.array(losses) - g["opt_losses64"]).max() <= 5 * TOL * max(1.0, g["opt_losses64"].max())
    sd = m.state_dict()
    for k, p in m.named_parameters():
        assert tuple(sd[k].shape) == tuple(p.shape)
        check_vs_digest(p.detach().cpu().numpy(), g, "p", k, cfg["seed"] + 7, tol=2e-5)
    # EMA shadow: swap it in and read it through the module's parameters
    eng.swap_in_ema()
    for k, p in m.named_parameters():
        check_vs_digest(p.detach().cpu().numpy(), g, "ema", k, cfg["seed"] + 7, tol=2e-5)
    eng.swap_in_ema()
    # module-level eval forward still works on the engine-owned (transposed) storage
    m.eval()
    with torch.no_grad():
        y1 = m(None, coords, t)
    assert torch.isfinite(y1).all() and y1.shape == (cfg["B"], 1)


def test_engine_graph_replay_equals_eager():
    from stnf.engine import TrainStep
    cfg = cases.MODEL_CASES["default227"]
    d = dev()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    res = []
    for graph in (False, True):
        m = build_model(cfg)
        eng = TrainStep(m, ema_decay=0.99, max_batch=cfg["B"], use_graph=graph)
        for _ in range(4):
            eng.step(None, coords, t, y)
        res.append((eng.mean_loss(), eng.flat.clone()))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0])
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-6)


def test_predictor_matches_module_forward():
    from stnf.engine import Predictor
    cfg = cases.MODEL_CASES["c2_b257"]
    d = dev()
    m = build_model(cfg)
    m.eval()
    rs = np.random.RandomState(4)
    n = 5000
    coords = torch.from_numpy(rs.uniform(0, 1, (n, 2)).astype(np.float32)).to(d)
    t = torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)).to(d)
    with torch.no_grad():
        ref = m(None, coords, t)
    for dense in (False, True):
        pr = Predictor(m, chunk=2048, use_graph=True, force_dense=dense)
        out = pr.predict(coords, t)
        assert torch.allclose(out, ref, rtol=1e-5, atol=2e-6)


def test_step_indexed_equals_step():
    """The library's batch gather (A0) + step == step on torch-indexed tensors, eager and graph."""
    from stnf.engine import TrainStep
    from stnf import _native as N
    cfg = cases.MODEL_CASES["default227"]
    d = dev()
    rs = np.random.RandomState(8)
    n = 1000
    coords = torch.from_numpy(rs.uniform(0, 1, (n, 2)).astype(np.float32)).to(d)
    t = torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)).to(d)
    y = torch.from_numpy(rs.standard_normal((n, 1)).astype(np.float32)).to(d)
    idx = torch.from_numpy(rs.permutation(n)[:193].astype(np.int64)).to(d)
    # the gather itself is exact
    co, to, yo = torch.empty(193, 2, device=d), torch.empty(193, device=d), torch.empty(193, 1, device=d)
    N.gather_batch(coords, t.view(-1), y, None, idx, co, to, yo, None)
    assert torch.equal(co, coords[idx]) and torch.equal(to, t.view(-1)[idx]) and torch.equal(yo, y[idx])
    res = []
    for mode in ("plain", "indexed", "indexed_graph"):
        m = build_model(cfg)
        eng = TrainStep(m, ema_decay=0.99, max_batch=193, use_graph=(mode == "indexed_graph"))
        for _ in range(3):
            if mode == "plain":
                eng.step(None, coords[idx], t[idx], y[idx])
            else:
                eng.step_indexed(coords, t, y, idx)
        res.append((eng.mean_loss(), eng.flat.clone()))
    for other in res[1:]:
        assert abs(res[0][0] - other[0]) <= 1e-6 * abs(res[0][0])
        assert torch.allclose(res[0][1], other[1], rtol=1e-5, atol=1e-6)


def test_c4_four_levels_window_vs_dense_and_oracle():
    """BASELINE config C4 model (4 resolutions, 49 728 knots, D = 49 798): the window path walks more
    than three levels; compare with the dense path and, for the prediction, with the float64 oracle."""
    cfg = dict(p=0, k_spatial_centers=[1024, 4096, 16384, 28224], k_temporal_centers=[10, 15, 45],
               hidden_dims=[256, 256, 128], layernorm=True, basis="wendland", output_dim=1, B=300, seed=61)
    X, coords, t, y = cases.make_inputs(cfg)
    d = dev()
    outs = []
    for dense in (False, True):
        m = build_model(cfg)
        m.force_dense_path = dense
        m.train()
        yp = m(*(torch.from_numpy(a).to(d) for a in (X, coords, t)))
        torch.nn.MSELoss()(yp, torch.from_numpy(y).to(d)).backward()
        outs.append((yp.detach().cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}))
    (yw, gw), (yd, gd) = outs
    assert np.abs(yw - yd).max() <= 2e-6 * max(1.0, np.abs(yd).max())
    for k in gd:
        assert rel_l2(gw[k], gd[k]) <= 2e-6, k
    yo, _, _, _, _ = orc.model_forward(X, coords, t, cases.make_state(cfg), cfg)
    assert np.abs(yw - yo).max() <= TOL * max(1.0, np.abs(yo).max())


def test_predictor_large_grid_chunks():
    """Dense-grid inference (config C5 shape, scaled): 200 000 points through 65 536-row chunks with a
    ragged tail, window vs dense kernels."""
    from stnf.engine import Predictor
    cfg = cases.MODEL_CASES["c2_b257"]
    d = dev()
    m = build_model(cfg)
    m.eval()
    g = torch.Generator().manual_seed(3)
    n = 200_000
    coords = torch.rand(n, 2, generator=g).to(d)
    t = (torch.randint(0, 100, (n,), generator=g).float() / 99.0).to(d)
    a = Predictor(m, chunk=65536, use_graph=True).predict(coords, t)
    b = Predictor(m, chunk=50000, use_graph=False, force_dense=True).predict(coords[:60000], t[:60000])
    assert a.shape == (n, 1) and torch.isfinite(a).all()
    assert torch.allclose(a[:60000], b, rtol=1e-5, atol=2e-6)


def test_engine_two_streams_equals_one():
    """The optional auxiliary stream (fork/join inside the library) must not change the numbers."""
    from stnf.engine import TrainStep
    cfg = cases.MODEL_CASES["c2_b257"]
    d = dev()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    res = []
    for two in (False, True):
        m = build_model(cfg)
        eng = TrainStep(m, ema_decay=0.99, max_batch=cfg["B"], two_streams=two)
        for _ in range(3):
            eng.step(None, coords, t, y)
        torch.cuda.synchronize()
        res.append((eng.mean_loss(), eng.flat.clone()))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0])   # the loss is summed with float atomics
    # Same kernels and the same order of every gradient sum; only the clip norm differs in its last bits
    # (one stream: partials out of the weight-gradient and reductions launches, one per workgroup; two streams:
    # stdadk_sumsq_f32, 256).  Adam (eps 1e-8) turns that into up to 1e-6 on a handful of near-zero entries after
    # three steps.
    torch.testing.assert_close(res[0][1], res[1][1], rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("name,dense,clip", [("c2_b257", False, 1.0), ("c2_b257", False, 0.0),
                                             ("default227", True, 1.0), ("c2_b257_mq3", False, 1.0),
                                             ("tiny9_mq5_nc2", False, 1.0)])
def test_one_call_step_equals_split_calls(name, dense, clip):
    """stdadk_train_step_f32 (forward, objective, backward, clip + AdamW + EMA in one call, the clip norm
    taken from the reductions launch) against the same step issued as stdadk_train_fwd_bwd_f32 +
    stdadk_sumsq_f32 + stdadk_adamw_ema_f32.  The dense path has gradients the reductions launch does not
    produce, so the entry falls back to the separate norm pass there."""
    from stnf.engine import TrainStep
    d = dev()
    res = []
    for whole in (True, False):
        kw = {}
        if name in cases.MODEL_CASES:
            cfg = cases.MODEL_CASES[name]
            m = build_model(cfg)
        else:
            m, cfg, lc = build_quantile_model(name)
            kw = dict(loss="pinball", quantile_levels=lc["taus"], non_crossing_weight=lc.get("nc_weight", 0.0),
                      non_crossing_power=lc.get("nc_power", 1))
        X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
        eng = TrainStep(m, lr=1e-2, ema_decay=0.99, grad_clip=clip, max_batch=cfg["B"], force_dense=dense, **kw)
        assert eng._whole_step
        eng._whole_step = whole
        for _ in range(3):
            eng.step(X if cfg["p"] else None, coords, t, y)
        torch.cuda.synchronize()
        assert int(eng.step_dev.item()) == 3
        res.append((eng.mean_loss(), eng.flat.clone(), eng.ema.clone()))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0])
    if clip == 0.0:
        assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    else:   # the squared norm is summed in a different order (per-workgroup partials of the step's own launches
        # against stdadk_sumsq_f32's 256); at lr 1e-2 Adam (eps 1e-8) turns that into up to 1e-6 on a handful of
        # near-zero gradient entries after three steps
        torch.testing.assert_close(res[0][1], res[1][1], rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(res[0][2], res[1][2], rtol=2e-6, atol=2e-6)


def test_reference_style_loop_matches_engine():
    """The reference's batch body written with torch pieces on the drop-in module (forward,
    nn.MSELoss, backward, clip_grad_norm_, optim.AdamW, ModelEMA — scripts/train_st_interp.py:608-712)
    and the fused engine take the same optimisation trajectory (dropout off)."""
    from stnf.engine import TrainStep
    from stnf.utils import ModelEMA
    cfg = cases.MODEL_CASES["default227"]
    d = dev()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    lr, wd, clip, decay, steps = 2e-2, 5e-4, 10.0, 0.99, 4
    # (1) driver-style loop
    m1 = build_model(cfg); m1.train()
    opt = torch.optim.AdamW([p for p in m1.parameters() if p.requires_grad], lr=lr, weight_decay=wd)
    ema = ModelEMA(m1, decay=decay)
    crit = torch.nn.MSELoss()
    l1 = []
    for _ in range(steps):
        opt.zero_grad()
        loss = crit(m1(torch.zeros(cfg["B"], 0, device=d), coords, t), y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m1.parameters(), clip)
        opt.step()
        ema.update(m1)
        l1.append(loss.item())
    # (2) fused engine
    m2 = build_model(cfg); m2.train()
    eng = TrainStep(m2, lr=lr, weight_decay=wd, grad_clip=clip, ema_decay=decay, max_batch=cfg["B"])
    l2 = []
    for _ in range(steps):
        eng.step(None, coords, t, y)
        l2.append(eng.mean_loss())
    assert np.allclose(l1, l2, rtol=2e-5)
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert rel_l2(p2.detach().cpu().numpy(), p1.detach().cpu().numpy()) <= 5e-5, k
    # validation under EMA weights, the driver's way (apply_shadow / restore) vs the engine's swap
    ema.apply_shadow(); m1.eval()
    with torch.no_grad():
        v1 = m1(torch.zeros(cfg["B"], 0, device=d), coords, t)
    ema.restore()
    eng.swap_in_ema(); m2.eval()
    with torch.no_grad():
        v2 = m2(None, coords, t)
    eng.swap_in_ema()
    assert torch.allclose(v1, v2, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ N3: quantile objectives + delta head
def build_quantile_model(name, dropout=0.0):
    from stnf.models import STInterpMLP
    cfg, lc = cases.quantile_cfg(name)
    m = STInterpMLP(p=cfg["p"], k_spatial_centers=cfg["k_spatial_centers"],
                    k_temporal_centers=cfg["k_temporal_centers"], hidden_dims=cfg["hidden_dims"],
                    dropout=dropout, layernorm=cfg["layernorm"], spatial_basis_function=cfg["basis"],
                    output_dim=cfg["output_dim"], use_delta_reparameterization=cfg["delta"])
    st = cases.make_state(cfg)
    assert [k for k, _ in m.named_parameters()] == list(st.keys())      # reference state_dict keys
    with torch.no_grad():
        for (_, p), (k, v) in zip(m.named_parameters(), st.items()):
            assert tuple(p.shape) == v.shape, k
            p.copy_(torch.from_numpy(v.copy()))
    m.force_window_path = True
    return m.to(dev()), cfg, lc


@pytest.mark.parametrize("Q,ycols,nc_w,nc_p", [(1, 1, 0.0, 1), (5, 1, 0.0, 1), (5, 1, 0.7, 1), (5, 1, 1.3, 2),
                                                 (8, 8, 0.4, 2), (3, 1, 2.0, 1)])
def test_loss_kernel_matches_oracle(Q, ycols, nc_w, nc_p):
    """stdadk_loss_f32 (check loss + non-crossing, broadcast targets, sub-gradients at the kinks)."""
    from stnf import _native as N
    rs = np.random.RandomState(100 + Q)
    B = 1031
    yp = rs.standard_normal((B, Q)).astype(np.float32)
    y = rs.standard_normal((B, ycols)).astype(np.float32)
    yp[:7, 0] = y[:7, 0]                       # e == 0: max() tie, gradient 1/2 - tau
    if Q > 1:
        yp[7:14, 1] = yp[7:14, 0]              # q_k == q_{k+1}: relu'(0) = 0
    taus = list(np.linspace(0.05, 0.95, Q)) if Q > 1 else [0.9]
    desc = N.make_loss("pinball", Q, ycols, taus, nc_w, nc_p)
    d = dev()
    dY = torch.empty(B, Q, device=d)
    acc = torch.zeros(1, device=d)
    gs = 1.0 / (B * Q)
    N.loss(desc, torch.from_numpy(yp).to(d), torch.from_numpy(y).to(d), gs, dY, acc)
    if ycols == 1:
        L, g = orc.quantile_objective(yp, y, taus, nc_w, nc_p)
    else:   # column-wise targets: evaluate the check loss per column, penalty on the predictions
        L0, g = orc.quantile_objective(yp, np.zeros((B, 1)), taus, nc_w, nc_p)
        Lz, gz = orc.quantile_objective(yp, np.zeros((B, 1)), taus, 0.0, 1)
        Lc = np.mean([orc.check_loss(yp[:, q], y[:, q], taus[q]) for q in range(Q)])
        gc = np.stack([orc.quantile_objective(yp[:, q:q + 1], y[:, q:q + 1], [taus[q]])[1][:, 0] / Q
                       for q in range(Q)], axis=1)
        L, g = L0 - Lz + Lc, g - gz + gc
    assert abs(acc.item() * gs - L) <= 1e-5 * max(1.0, abs(L))
    # the kinks are hit exactly, so the gradient is comparable element by element
    assert np.abs(dY.cpu().numpy() - g).max() <= 1e-6 * np.abs(g).max()
    # MSE through the same entry point, broadcast targets
    if ycols == 1:
        acc.zero_()
        N.loss(N.make_loss("mse", Q, 1), torch.from_numpy(yp).to(d), torch.from_numpy(y).to(d), gs, dY, acc)
        ref = ((yp.astype(np.float64) - y) ** 2)
        assert abs(acc.item() - ref.sum()) <= 1e-5 * ref.sum()
        assert np.abs(dY.cpu().numpy() - 2 * (yp.astype(np.float64) - y) * gs).max() <= 1e-6 * gs * 10


def test_delta_head_kernels_match_oracle_and_reference():
    from stnf import _native as N
    g = load("n3_known_answers")
    delta64 = g["ka_delta"]                           # incl. J = 0, tie and clamp-boundary rows
    d = dev()
    Q, d1 = delta64.shape
    # rows embedded in a wider buffer, as TrainStep's flat storage has them (stride 12 > d+1 = 9)
    buf = torch.zeros(Q, 12, device=d)
    buf[:, :d1] = torch.from_numpy(delta64).float()
    delta = buf[:, :d1]
    Wo, bo = torch.empty(Q, d1 - 1, device=d), torch.empty(Q, device=d)
    N.delta_head(delta, Wo, bo)
    Wr, br = orc.delta_head(delta64.astype(np.float32))
    assert np.abs(Wo.cpu().numpy() - Wr).max() < 1e-6 and np.abs(bo.cpu().numpy() - br).max() < 1e-6
    rs = np.random.RandomState(3)
    dW, db = rs.standard_normal((Q, d1 - 1)).astype(np.float32), rs.standard_normal(Q).astype(np.float32)
    gbuf = torch.zeros(Q, 12, device=d)
    acc = torch.zeros(1, device=d)
    N.delta_head_backward(delta, torch.from_numpy(dW).to(d), torch.from_numpy(db).to(d), 0.3, 2.0,
                          gbuf[:, :d1], acc)
    P, gP = orc.p_nc_delta(delta64.astype(np.float32))
    ref = orc.delta_head_backward(dW, db) + 0.3 * gP
    assert np.abs(gbuf[:, :d1].cpu().numpy() - ref).max() < 1e-5
    assert (gbuf[:, d1:] == 0).all()
    assert abs(acc.item() - 2.0 * P) < 1e-5
    # the reference's own value and gradient of P_nc(delta) on the same vector
    assert abs(P - float(g["ka_pnc"])) < 1e-6
    assert np.abs(gP - g["ka_pnc_grad"]).max() < 1e-12
    from stnf import losses
    ps = [torch.from_numpy(delta64[k]).float().to(d) for k in range(Q)]
    assert abs(losses.compute_p_nc_delta_penalty(ps).item() - float(g["ka_pnc"])) < 1e-5
    assert losses.compute_p_nc_delta_penalty(ps[:1]).item() == 0.0
    assert losses.compute_p_nc_delta_penalty(None).item() == 0.0


def test_losses_module_known_answers():
    """stnf.losses on the device against values of the reference's functions (incl. the vectors of
    its tests test_crps_eq_4_6.py)."""
    from stnf import losses
    g = load("n3_known_answers")
    d = dev()
    yp, y = torch.from_numpy(g["ka_yp"]).float().to(d), torch.from_numpy(g["ka_y"]).float().to(d)
    for i, q in enumerate(cases.TAUS5):
        assert abs(losses.quantile_loss(yp[:, i:i + 1], y, q).item() - float(g[f"ka_qloss_{i}"])) < 1e-6
    assert abs(losses.non_crossing_penalty(yp, "mean", 1).item() - float(g["ka_nc1"])) < 1e-5
    assert abs(losses.non_crossing_penalty(yp, "sum", 2).item() - float(g["ka_nc2_sum"])) < 1e-3
    assert losses.non_crossing_penalty(yp[:, :1]).item() == 0.0
    with pytest.raises(ValueError, match="Unsupported power"):
        losses.non_crossing_penalty(yp, "mean", 3)
    with pytest.raises(ValueError, match="Unsupported reduction"):
        losses.non_crossing_penalty(yp, "median", 1)
    assert abs(losses.compute_crps_multi_quantile(yp, y, cases.TAUS5) - float(g["ka_crps"])) < 1e-6
    assert abs(losses.compute_crps_multi_quantile(g["ka_yp"], g["ka_y"], cases.TAUS5,
                                                  weights=np.array([1.0, 2.0, 3.0, 2.0, 1.0]))
               - float(g["ka_crps_w"])) < 1e-12
    yt = np.array([2.0, 3.0, 4.0, 5.0])
    pd = {0.05: yt - 1.0, 0.25: yt - 0.5, 0.5: yt.copy(), 0.75: yt + 0.5, 0.95: yt + 1.0}
    assert abs(losses.compute_crps(pd, yt) - float(g["ka_crps_thesis"])) < 1e-15
    assert abs(losses.compute_crps({0.5: np.array([2.5])}, np.array([2.0])) - float(g["ka_crps_single"])) < 1e-15
    with pytest.raises(ValueError, match="cannot be empty"):
        losses.compute_crps({}, yt)
    with pytest.raises(ValueError, match="weights length"):
        losses.compute_crps({0.5: yt, 0.9: yt}, yt, weights=np.array([0.5]))


def _torch_n3_loss(m, yp, y, lc):
    """The batch objective as the reference's driver composes it from torch ops
    (train_st_interp.py:622-658), on this package's differentiable forward()."""
    def ql(a, q):
        e = y - a
        return torch.mean(torch.max((q - 1) * e, q * e))
    taus = lc["taus"]
    if len(taus) == 1:
        return ql(yp, taus[0])
    loss = torch.mean(torch.stack([ql(yp[:, i:i + 1], q) for i, q in enumerate(taus)]))
    if lc.get("delta"):
        pen = torch.zeros((), device=yp.device)
        for dk in m.get_delta_parameters()[1:]:
            S = torch.clamp(-dk[1:], min=0.0).sum()
            pen = pen + dk[0] - torch.max(dk[0], S)
        return loss + lc.get("nc_lambda", 0.0) * pen
    if lc.get("nc_weight", 0.0) > 0:
        v = torch.relu(yp[:, :-1] - yp[:, 1:])
        if lc.get("nc_power", 1) == 2:
            v = v ** 2
        loss = loss + lc["nc_weight"] * v.sum(1).mean()
    return loss


@pytest.mark.parametrize("name", list(cases.QUANTILE_CASES))
def test_quantile_module_loop_matches_reference(name):
    """forward() + the driver's torch loss + backward() (autograd through the C ABI, delta head
    included) against the float64 golden of the reference."""
    m, cfg, lc = build_quantile_model(name)
    g = load(name)
    d = dev()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    m.train()
    yp = m(X, coords, t)
    assert yp.shape == (cfg["B"], cfg["output_dim"])
    assert np.abs(yp.detach().cpu().numpy() - g["y64"]).max() <= TOL * max(1.0, np.abs(g["y64"]).max())
    loss = _torch_n3_loss(m, yp, y, lc)
    assert abs(loss.item() - float(g["loss64"])) <= 2 * TOL * max(1.0, abs(float(g["loss64"])))
    loss.backward()
    for k, p in m.named_parameters():
        assert p.grad is not None and tuple(p.grad.shape) == tuple(p.shape), k
        check_vs_digest(p.grad.cpu().numpy(), g, "g", k, cfg["seed"] + 7, tol=2e-5)
    if cfg["delta"]:
        # beta_k = sum_{l<=k} delta_l through the torch trunk, as the reference's test does
        with torch.no_grad():
            m.eval()
            feats = m.build_features(X, coords, t)[:, :m.input_dim]
            h = m.mlp_trunk(feats)
            beta = torch.cumsum(torch.stack(list(m.delta_params)), dim=0)
            ref = beta[:, :1].t() + h @ beta[:, 1:].t()
            got = m(X, coords, t)
        assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("name", list(cases.QUANTILE_CASES))
def test_quantile_engine_steps_match_reference(name, dense):
    """OPT['steps'] fused steps with the check loss / non-crossing / delta head inside the step."""
    from stnf.engine import TrainStep
    m, cfg, lc = build_quantile_model(name)
    g = load(name)
    o = cases.OPT
    d = dev()
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, lr=o["lr"], weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"],
                    grad_clip=o["grad_clip"], ema_decay=o["ema_decay"], max_batch=cfg["B"], force_dense=dense,
                    loss="pinball", quantile_levels=lc["taus"], non_crossing_weight=lc.get("nc_weight", 0.0),
                    non_crossing_power=lc.get("nc_power", 1), non_crossing_lambda=lc.get("nc_lambda", 0.0))
    losses = []
    for _ in range(o["steps"]):
        eng.step(X if cfg["p"] else None, coords, t, y)
        losses.append(eng.mean_loss())
    ref = g["opt_losses64"]
    assert np.abs(np.array(losses) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), (losses, ref)
    for k, p in m.named_parameters():
        check_vs_digest(p.detach().cpu().numpy(), g, "p", k, cfg["seed"] + 7, tol=1e-4)
    eng.swap_in_ema()
    for k, p in m.named_parameters():
        check_vs_digest(p.detach().cpu().numpy(), g, "ema", k, cfg["seed"] + 7, tol=1e-4)
    eng.swap_in_ema()


def test_quantile_engine_graph_and_indexed():
    """Graph replay of the quantile step (delta head + P_nc inside the capture) == eager, and
    step_indexed gathers (N,1) targets for a (B,5) head."""
    from stnf.engine import TrainStep
    d = dev()
    res = []
    for graph in (False, True):
        m, cfg, lc = build_quantile_model("default227_delta5")
        m.train()
        X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
        eng = TrainStep(m, lr=1e-3, grad_clip=10.0, ema_decay=0.99, max_batch=64, use_graph=graph,
                        loss="pinball", quantile_levels=lc["taus"], non_crossing_lambda=0.05)
        for s in range(4):
            idx = torch.arange(64, device=d) + 16 * s
            eng.step_indexed(coords, t, y, idx)
        res.append((torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone(), eng.mean_loss()))
    assert torch.equal(res[0][0], res[1][0])
    assert abs(res[0][1] - res[1][1]) <= 1e-6 * max(1.0, abs(res[0][1]))


def test_delta_model_surface():
    """The reference's delta-head API (tests/stnf/models/test_st_interp_delta_reparameterization.py)."""
    from stnf.models import STInterpMLP, create_model
    d = dev()
    m = STInterpMLP(p=0, k_spatial_centers=[9], k_temporal_centers=[5], hidden_dims=[32, 16], output_dim=5,
                    use_delta_reparameterization=True).to(d)
    assert m.mlp_trunk is not None and len(m.delta_params) == 5 and m.last_hidden_dim == 16
    assert all(tuple(p.shape) == (17,) for p in m.get_delta_parameters())
    assert all(p.abs().max().item() < 0.1 for p in m.delta_params)          # N(0, 0.01) init
    coords, t = torch.rand(10, 2, device=d), torch.rand(10, 1, device=d)
    m.eval()
    with torch.no_grad():
        y1, y2 = m(torch.empty(10, 0, device=d), coords, t), m(None, coords, t)
    assert y1.shape == (10, 5) and torch.equal(y1, y2)
    m.train()
    m(None, coords, t).sum().backward()
    assert all(p.grad is not None for p in m.parameters())
    assert any(p.grad.abs().sum().item() > 0 for p in m.delta_params)
    pen = m.compute_sparsity_penalty('sparse_group', 0.01, 0.01)
    assert torch.isfinite(pen['total_penalty'])
    # single quantile: the flag is ignored, standard head (reference :668)
    m1 = STInterpMLP(k_spatial_centers=[9], k_temporal_centers=[5], hidden_dims=[32, 16], output_dim=1,
                     use_delta_reparameterization=True)
    assert m1.mlp_trunk is None and m1.delta_params is None and m1.get_delta_parameters() is None
    m2 = create_model({'regression_type': 'multi-quantile', 'quantile_levels': [0.05, 0.25, 0.5, 0.75, 0.95],
                       'use_delta_reparameterization': True, 'k_spatial_centers': [9],
                       'k_temporal_centers': [5], 'hidden_dims': [32, 16]})
    assert m2.use_delta_reparameterization and m2.output_dim == 5 and len(m2.delta_params) == 5


# ------------------------------------------------------------------ N2: learnable knots (DA-STDK)
def build_learn_model(name):
    from stnf.models import STInterpMLP
    cfg, kn = cases.learn_cfg(name)
    g = load(name)
    m = STInterpMLP(p=cfg["p"], k_spatial_centers=cfg["k_spatial_centers"],
                    k_temporal_centers=cfg["k_temporal_centers"], hidden_dims=cfg["hidden_dims"],
                    dropout=0.0, layernorm=cfg["layernorm"], spatial_learnable=True,
                    spatial_basis_function=cfg["basis"], output_dim=cfg["output_dim"],
                    gradient_damping=kn.get("gradient_damping", False),
                    damping_threshold=kn.get("damping_threshold", 0.3),
                    damping_strength=kn.get("damping_strength", 1.0))
    sb = m.spatial_basis
    # grid knots and log-bandwidths are bit-identical with the reference's before the perturbation
    assert np.array_equal(sb.centers_init.numpy(), g["in_centers_init"])
    dc, dlb = cases.knot_perturbation(cfg)
    assert np.array_equal((sb.centers.detach().numpy() + dc).astype(np.float32), g["in_centers"])
    # (torch.log in fp32 may differ by one ulp between host CPUs' vector paths; the case's state is loaded below)
    assert np.abs((sb.log_bandwidths.detach().numpy() + dlb).astype(np.float32) - g["in_log_bw"]).max() <= 5e-7
    names = [k for k, _ in m.named_parameters()]
    assert names[:2] == ["spatial_basis.centers", "spatial_basis.log_bandwidths"]
    st = cases.make_state(cfg)
    assert names[2:] == list(st.keys())
    with torch.no_grad():
        sb.centers.copy_(torch.from_numpy(g["in_centers"]))
        sb.log_bandwidths.copy_(torch.from_numpy(g["in_log_bw"]))
        for (k, p) in list(m.named_parameters())[2:]:
            p.copy_(torch.from_numpy(st[k].copy()))
    m.force_window_path = True
    return m.to(dev()), cfg, kn, g


def _learn_loss(m, X, coords, t, y, kn):
    loss = torch.nn.functional.mse_loss(m(X, coords, t), y)
    if kn.get("domain_penalty_weight", 0.0) > 0:
        loss = loss + kn["domain_penalty_weight"] * m.compute_domain_penalty()
    if kn.get("movement_penalty_weight", 0.0) > 0:
        loss = loss + kn["movement_penalty_weight"] * m.compute_movement_penalty()
    return loss


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("name", list(cases.LEARN_CASES))
def test_learnable_module_loop_matches_reference(name, dense):
    """forward + MSE + the driver's penalties + backward with learnable knots: gradients into the
    centres (damping hook applied) and log-bandwidths against the reference's float64 golden, on the
    window path (where it applies) and on the materialising path."""
    m, cfg, kn, g = build_learn_model(name)
    m.force_dense_path = dense
    d = dev()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    m.train()
    yp = m(X, coords, t)
    assert np.abs(yp.detach().cpu().numpy() - g["y64"]).max() <= TOL * max(1.0, np.abs(g["y64"]).max())
    assert abs(m.compute_domain_penalty().item() - float(g["domain_pen64"])) <= 1e-5 * max(1e-3, float(g["domain_pen64"]))
    assert abs(m.compute_movement_penalty().item() - float(g["movement_pen64"])) <= 1e-5 * float(g["movement_pen64"])
    loss = _learn_loss(m, X, coords, t, y, kn)
    assert abs(loss.item() - float(g["loss64"])) <= 2 * TOL * max(1.0, abs(float(g["loss64"])))
    loss.backward()
    errs = {}
    for k, p in m.named_parameters():
        assert p.grad is not None and tuple(p.grad.shape) == tuple(p.shape), k
        errs[k] = check_vs_digest(p.grad.cpu().numpy(), g, "g", k, cfg["seed"] + 7, tol=2e-5)
    # at least as close to the truth as the reference's own fp32 run (cdist's matmul expansion) where
    # that run is the limiting one
    ref_err = float(g["gerr32_rell2/spatial_basis.centers"])
    assert errs["spatial_basis.centers"] <= max(2e-5, ref_err)
    # eval-mode values of the embedding use exp(log_bw)
    with torch.no_grad():
        phi = m.spatial_basis(coords)
    truth = orc.spatial_basis(coords.cpu().numpy(), g["in_centers"], np.exp(g["in_log_bw"].astype(np.float64)),
                              cfg["basis"])
    assert np.abs(phi.cpu().numpy() - truth).max() <= TOL


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("name", list(cases.LEARN_CASES))
def test_learnable_engine_steps_match_reference(name, dense):
    """OPT['steps'] fused steps with the knot group: own lr (x0.05), own clip (x0.1), damping and
    penalties inside stdadk_knot_backward_f32; parameters and EMA against the float64 golden."""
    from stnf.engine import TrainStep
    m, cfg, kn, g = build_learn_model(name)
    o = cases.OPT
    d = dev()
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, lr=o["lr"], weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"],
                    grad_clip=o["grad_clip"], ema_decay=o["ema_decay"], max_batch=cfg["B"],
                    basis_lr_ratio=cases.BASIS_LR_RATIO, basis_clip_ratio=cases.BASIS_CLIP_RATIO,
                    domain_penalty_weight=kn.get("domain_penalty_weight", 0.0),
                    movement_penalty_weight=kn.get("movement_penalty_weight", 0.0), force_dense=dense)
    windowable = cfg["basis"] != "gaussian" and cfg["hidden_dims"][0] in (128, 256)
    assert eng.learnable and eng.uses_window == (windowable and not dense)
    losses = []
    for _ in range(o["steps"]):
        eng.step(X if cfg["p"] else None, coords, t, y)
        losses.append(eng.mean_loss())
    ref = g["opt_losses64"]
    assert np.abs(np.array(losses) - ref).max() <= 5 * TOL * max(1.0, np.abs(ref).max()), (losses, ref)
    # (Adam's m/sqrt(v) turns rounding-level differences of near-zero gradients — knot rows that few of
    #  the 257 observations touch — into lr-sized parameter differences: 1.0e-4 of the tensor norm on the default
    #  launch shapes, 1.05e-4 with another summation order (STDADK_NO_FUSED_TAIL / STDADK_KROT, round-3 env matrix),
    #  hence 2e-4 after three steps; the gradients themselves are pinned at 1e-5 by the tests above)
    for k, p in m.named_parameters():
        check_vs_digest(p.detach().cpu().numpy(), g, "p", k, cfg["seed"] + 7, tol=2e-4)
    eng.swap_in_ema()
    for k, p in m.named_parameters():
        check_vs_digest(p.detach().cpu().numpy(), g, "ema", k, cfg["seed"] + 7, tol=2e-4)
    eng.swap_in_ema()
    # the state_dict keeps the reference's keys and shapes for the knot tensors
    sd = m.state_dict()
    assert tuple(sd["spatial_basis.centers"].shape) == (m.k_spatial, 2)
    assert tuple(sd["spatial_basis.log_bandwidths"].shape) == (m.k_spatial,)
    assert "spatial_basis.centers_init" in sd


def test_learnable_engine_graph_freeze_and_errors():
    """Graph replay == eager with the knot group; basis lr 0 keeps the knots frozen (progressive
    unfreezing, train_st_interp.py:477-478,582-602); the knot entry point refuses fixed-knot state."""
    from stnf.engine import TrainStep
    from stnf import _native as N
    d = dev()
    res = []
    for graph in (False, True):
        m, cfg, kn, g = build_learn_model("default227_learn")
        m.train()
        X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
        eng = TrainStep(m, lr=1e-3, grad_clip=10.0, ema_decay=0.99, max_batch=64, use_graph=graph,
                        domain_penalty_weight=0.01)
        c0 = m.spatial_basis.centers.detach().clone()
        eng.set_basis_lr(0.0)
        eng.step_indexed(coords, t, y, torch.arange(64, device=d))
        wd_only = c0 * (1.0 - 0.0 * eng.wd)
        assert torch.equal(m.spatial_basis.centers.detach(), wd_only)        # frozen: lr 0 => no move
        eng.set_basis_lr(1e-3 * 0.05)
        for s in range(1, 5):
            eng.step_indexed(coords, t, y, torch.arange(64, device=d) + 16 * s)
        assert not torch.equal(m.spatial_basis.centers.detach(), c0)
        res.append(torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone())
    assert torch.equal(res[0], res[1])
    # fixed-knot state has no knot gradient
    m = build_model(cases.MODEL_CASES["default227"])
    st = m._step_state(d, force_dense=True)
    ws = torch.empty(N.step_workspace_bytes(st.basis, st.desc, 8, st.flags) // 4, device=d)
    with pytest.raises(RuntimeError, match="STDADK_FLAG_LOG_BW"):
        N.knot_backward(st.basis, st.desc, st.params, torch.rand(8, 2, device=d), 8, ws, st.flags, None,
                        torch.empty(227, 2, device=d), torch.empty(227, device=d))


@pytest.mark.parametrize("B", [193, 9000, 70000])
def test_indexed_step_large_batches_equals_gathered(B):
    """Batches beyond the single-workgroup binning (multi-kernel histogram / tiled scan / scatter /
    in-cell ordering, G up to 256): rows read in place through idx == the gathered batch, bit for bit
    (the binning is deterministic, so both runs do the same arithmetic)."""
    from stnf.engine import TrainStep
    cfg = dict(cases.MODEL_CASES["default227"], p=2)
    d = dev()
    rs = np.random.RandomState(B)
    n = B + 5000
    coords = torch.from_numpy(rs.uniform(-0.05, 1.05, (n, 2)).astype(np.float32)).to(d)
    t = torch.from_numpy(rs.uniform(0, 1, (n,)).astype(np.float32)).to(d)
    X = torch.from_numpy(rs.standard_normal((n, 2)).astype(np.float32)).to(d)
    y = torch.from_numpy(rs.standard_normal((n, 1)).astype(np.float32)).to(d)
    idx = torch.from_numpy(rs.permutation(n)[:B].astype(np.int64)).to(d)
    res = []
    for mode in ("gathered", "indexed"):
        m = build_model(cfg)
        eng = TrainStep(m, ema_decay=0.99, max_batch=B)
        assert eng.uses_window
        eng.indexed_min_batch = 0          # B = 193: the single-workgroup binning reading in place, too
        for _ in range(2):
            if mode == "gathered":
                eng.step(X[idx], coords[idx], t[idx], y[idx])
            else:
                eng.step_indexed(coords, t, y, idx, X_all=X)
        res.append((eng.mean_loss(), eng.flat.clone()))
    assert torch.equal(res[0][1], res[1][1])
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0])


@pytest.mark.parametrize("name,move,grow", [("default227", 0.4, 1.3), ("default227_tri", 3.0, 2.2),
                                            ("c2_b257", 1.5, 1.6), ("c2_b257", 6.0, 0.5)])
def test_learnable_window_follows_moved_knots(name, move, grow):
    """Knots displaced by up to `move` grid cells (finest level) and bandwidths scaled by up to `grow`:
    the window path widens its candidate windows on the device and must agree with the materialising
    path and with the oracle (values, dW0, knot gradients) — including overflowing candidate lists."""
    from stnf.models import STInterpMLP
    cfg = cases.MODEL_CASES[name]
    d = dev()
    rs = np.random.RandomState(int(move * 10) + 7)
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    st = cases.make_state(cfg)
    out = {}
    Ks = sum(cfg["k_spatial_centers"])
    step = 1.0 / (int(np.sqrt(cfg["k_spatial_centers"][-1])) - 1)
    dc = rs.uniform(-move * step, move * step, (Ks, 2)).astype(np.float32)
    dlb = np.log(rs.uniform(min(1.0, grow), max(1.0, grow), Ks)).astype(np.float32)
    for mode in ("window", "dense"):
        m = STInterpMLP(p=cfg["p"], k_spatial_centers=cfg["k_spatial_centers"],
                        k_temporal_centers=cfg["k_temporal_centers"], hidden_dims=cfg["hidden_dims"],
                        dropout=0.0, layernorm=cfg["layernorm"], spatial_learnable=True,
                        spatial_basis_function=cfg["basis"])
        with torch.no_grad():
            m.spatial_basis.centers.add_(torch.from_numpy(dc))
            m.spatial_basis.log_bandwidths.add_(torch.from_numpy(dlb))
            for (k, p) in list(m.named_parameters())[2:]:
                p.copy_(torch.from_numpy(st[k].copy()))
        m = m.to(d)
        m.force_dense_path = mode == "dense"
        m.force_window_path = mode == "window"
        m.train()
        yp = m(X, coords, t)
        torch.nn.functional.mse_loss(yp, y).backward()
        out[mode] = (yp.detach().cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in m.named_parameters()},
                     m.spatial_basis.centers.detach().cpu().numpy(), m.spatial_basis.log_bandwidths.detach().cpu().numpy())
    yw, gw, cen, lbw = out["window"]
    yd, gd, _, _ = out["dense"]
    params = dict(st)
    params["spatial_basis.centers"], params["spatial_basis.log_bandwidths"] = cen, lbw
    cinit, _, _ = orc.uniform_knots(cfg["k_spatial_centers"])
    yo, _, go = orc.learnable_step_grads(*cases.make_inputs(cfg), params, dict(cfg), {}, cinit)
    assert np.abs(yw - yo).max() <= TOL * max(1.0, np.abs(yo).max())
    assert np.abs(yd - yo).max() <= TOL * max(1.0, np.abs(yo).max())
    for k in go:
        assert rel_l2(gw[k], go[k]) <= 2e-5, (k, "window", rel_l2(gw[k], go[k]))
        assert rel_l2(gd[k], go[k]) <= 2e-5, (k, "dense", rel_l2(gd[k], go[k]))


@pytest.mark.parametrize("method,basis", [("random_site", "wendland"), ("gmm", "triangular")])
def test_scattered_learnable_knots_match_oracle(method, basis):
    """The shipped YAML's combination — data-adaptive (scattered) initial knots, learnable — runs the
    materialising kernels (no grid indexing): values and every gradient against the oracle."""
    from stnf.models import STInterpMLP
    d = dev()
    np.random.seed(11)
    pts = cases.init_points()
    cfg = dict(p=0, k_spatial_centers=[25, 81], k_temporal_centers=[10, 15], hidden_dims=[256, 128],
               layernorm=True, basis=basis, output_dim=1, B=300, seed=61)
    m = STInterpMLP(p=0, k_spatial_centers=cfg["k_spatial_centers"], k_temporal_centers=cfg["k_temporal_centers"],
                    hidden_dims=cfg["hidden_dims"], dropout=0.0, layernorm=True, spatial_learnable=True,
                    spatial_init_method=method, spatial_basis_function=basis, train_coords=pts)
    st = cases.make_state(cfg)
    with torch.no_grad():
        for (k, p) in list(m.named_parameters())[2:]:
            p.copy_(torch.from_numpy(st[k].copy()))
    m = m.to(d)
    assert m._basis_desc().n_levels == 0
    X, coords, t, y = cases.make_inputs(cfg)
    m.train()
    yp = m(None, torch.from_numpy(coords).to(d), torch.from_numpy(t).to(d))
    torch.nn.functional.mse_loss(yp, torch.from_numpy(y).to(d)).backward()
    params = dict(st)
    params["spatial_basis.centers"] = m.spatial_basis.centers.detach().cpu().numpy()
    params["spatial_basis.log_bandwidths"] = m.spatial_basis.log_bandwidths.detach().cpu().numpy()
    yo, _, go = orc.learnable_step_grads(X, coords, t, y, params, cfg, {}, params["spatial_basis.centers"])
    assert np.abs(yp.detach().cpu().numpy() - yo).max() <= TOL * max(1.0, np.abs(yo).max())
    for k, p in m.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), go[k]) <= 2e-5, k


def test_path_choice_small_tables_run_materialised():
    """Default path choice: knot tables under 1024 knots take the materialising kernels (faster there),
    larger compact-support grids the window kernels; STDADK_FLAG_WINDOW / FLAG_DENSE override it."""
    from stnf.engine import TrainStep
    small = build_model(cases.MODEL_CASES["default227"])
    small.force_window_path = False
    assert not TrainStep(small, max_batch=64).uses_window
    small2 = build_model(cases.MODEL_CASES["default227"])
    assert TrainStep(small2, max_batch=64).uses_window               # helper forces the window kernels
    big = build_model(cases.MODEL_CASES["c2_b257"])
    big.force_window_path = False
    assert TrainStep(big, max_batch=64).uses_window
    big2 = build_model(cases.MODEL_CASES["c2_b257"])
    assert not TrainStep(big2, max_batch=64, force_dense=True).uses_window


@pytest.mark.parametrize("name", ["default227", "default227_learn", "default227_mq5"])
def test_pipelined_batch_preparation_equals_inline(name):
    """next_idx: gather + binning of the following batch on a side stream into a second workspace;
    the parameters after several steps are bit-identical with the in-step preparation (same binning,
    same arithmetic), also when an announced batch is not the one that follows."""
    from stnf.engine import TrainStep
    d = dev()
    rs = np.random.RandomState(5)
    n = 3000
    coords = torch.from_numpy(rs.uniform(0, 1, (n, 2)).astype(np.float32)).to(d)
    t = torch.from_numpy(rs.uniform(0, 1, (n,)).astype(np.float32)).to(d)
    y = torch.from_numpy(rs.standard_normal((n, 1)).astype(np.float32)).to(d)
    perm = torch.from_numpy(rs.permutation(n).astype(np.int64)).to(d)
    res = []
    for mode in ("inline", "pipelined", "pipelined_mispredicted"):
        kw = {}
        if name in cases.LEARN_CASES:
            m, cfg, kn, g = build_learn_model(name)
            kw = dict(domain_penalty_weight=0.01)
        elif name in cases.QUANTILE_CASES:
            m, cfg, lc = build_quantile_model(name)
            kw = dict(loss="pinball", quantile_levels=lc["taus"], non_crossing_weight=0.5)
        else:
            m = build_model(cases.MODEL_CASES[name])
        m.train()
        eng = TrainStep(m, lr=1e-3, ema_decay=0.99, max_batch=256, **kw)
        assert eng.uses_window
        for s in range(6):
            idx = perm[256 * s:256 * s + 256]
            nxt = perm[256 * (s + 1):256 * (s + 1) + 256]
            if mode == "inline":
                eng.step_indexed(coords, t, y, idx)
            elif mode == "pipelined":
                eng.step_indexed(coords, t, y, idx, next_idx=nxt)
            else:       # announces a batch that is NOT used next every other step
                eng.step_indexed(coords, t, y, idx, next_idx=nxt if s % 2 else perm[:256])
        res.append((eng.flat.clone(), eng.mean_loss()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][0], res[2][0])
    assert abs(res[0][1] - res[1][1]) <= 1e-6 * abs(res[0][1])


def test_coarse_knots_learnable_large_batch():
    """B = 3000 on the 227-knot table: every knot sees hundreds to thousands of rows (long per-knot
    gathers, many list flushes); values and all gradients, knots included, against the oracle and
    bit-identical between two runs.  (Splitting such knots over several waves was measured and is NOT
    done: at B = 65 536 it made the gather 15 % slower — the kernel is L2-throughput-bound there.)"""
    from stnf.models import STInterpMLP
    cfg = dict(cases.MODEL_CASES["default227"], B=3000, seed=71)
    d = dev()
    X, coords, t, y = cases.make_inputs(cfg)
    st = cases.make_state(cfg)
    dc, dlb = cases.knot_perturbation(cfg)
    outs = []
    for rep in range(2):
        m = STInterpMLP(p=0, k_spatial_centers=cfg["k_spatial_centers"], k_temporal_centers=cfg["k_temporal_centers"],
                        hidden_dims=cfg["hidden_dims"], dropout=0.0, layernorm=True, spatial_learnable=True)
        with torch.no_grad():
            m.spatial_basis.centers.add_(torch.from_numpy(dc))
            m.spatial_basis.log_bandwidths.add_(torch.from_numpy(dlb))
            for (k, p) in list(m.named_parameters())[2:]:
                p.copy_(torch.from_numpy(st[k].copy()))
        m = m.to(d)
        m.force_window_path = True
        m.train()
        yp = m(None, torch.from_numpy(coords).to(d), torch.from_numpy(t).to(d))
        torch.nn.functional.mse_loss(yp, torch.from_numpy(y).to(d)).backward()
        outs.append({k: p.grad.clone() for k, p in m.named_parameters()})
        cen, lbw = m.spatial_basis.centers.detach().cpu().numpy(), m.spatial_basis.log_bandwidths.detach().cpu().numpy()
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    params = dict(st)
    params["spatial_basis.centers"], params["spatial_basis.log_bandwidths"] = cen, lbw
    cinit, _, _ = orc.uniform_knots(cfg["k_spatial_centers"])
    yo, _, go = orc.learnable_step_grads(X, coords, t, y, params, cfg, {}, cinit)
    assert np.abs(yp.detach().cpu().numpy() - yo).max() <= TOL * max(1.0, np.abs(yo).max())
    for k in go:
        assert rel_l2(outs[0][k].cpu().numpy(), go[k]) <= 2e-5, (k, rel_l2(outs[0][k].cpu().numpy(), go[k]))


def test_run_epoch_equals_manual_loop():
    """TrainStep.run_epoch over a DeviceDataset (pipelined batch preparation, ragged last batch) ==
    the plain step_indexed loop over the same index slices, bit for bit."""
    from stnf.engine import TrainStep
    from stnf.dataio.device_dataset import DeviceDataset
    d = dev()
    rs = np.random.RandomState(21)
    n = 1000
    ds = DeviceDataset(torch.from_numpy(rs.uniform(0, 1, (n, 2)).astype(np.float32)).to(d),
                       torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)).to(d),
                       torch.from_numpy(rs.standard_normal((n, 1)).astype(np.float32)).to(d))
    res = []
    for mode in ("epoch", "manual"):
        m = build_model(cases.MODEL_CASES["default227"])
        m.train()
        eng = TrainStep(m, lr=1e-3, ema_decay=0.99, max_batch=256)
        gen = torch.Generator(device=d).manual_seed(3)
        if mode == "epoch":
            loss = eng.run_epoch(ds, 256, generator=gen)
        else:
            for idx in ds.epoch_batches(256, generator=gen):
                eng.step_indexed(ds.coords, ds.t, ds.y, idx)
            loss = eng.mean_loss()
        res.append((eng.flat.clone(), loss))
    assert torch.equal(res[0][0], res[1][0])
    assert abs(res[0][1] - res[1][1]) <= 1e-6 * abs(res[0][1])


@pytest.mark.parametrize("variant", ["four_levels", "h128_noln", "tri_covariates", "learnable_tri", "q3_pinball",
                                     "b4096", "b33"])
def test_fused_step_kernels_window_vs_materialised(variant):
    """The one-launch step kernels (layer 0 + tail forward/backward; all weight gradients) across their
    template variants — level chunking, H = 128, no LayerNorm, triangular basis, covariates, learnable
    knots, several outputs, a full 4096-row batch, a ragged 33-row one: two engine steps on the window
    path against the same two steps on the materialising kernels."""
    from stnf.models import STInterpMLP
    from stnf.engine import TrainStep
    d = dev()
    kw = dict(p=0, k_spatial_centers=[1024, 1600], k_temporal_centers=[10, 15], hidden_dims=[256, 128],
              dropout=0.0, layernorm=True, spatial_basis_function="wendland")
    ekw, B, ycols = {}, 700, 1
    if variant == "four_levels":
        kw.update(k_spatial_centers=[64, 256, 576, 1600])
    elif variant == "h128_noln":
        kw.update(hidden_dims=[128, 64], layernorm=False)
    elif variant == "tri_covariates":
        kw.update(p=3, spatial_basis_function="triangular")
    elif variant == "learnable_tri":
        kw.update(spatial_learnable=True, spatial_basis_function="triangular", gradient_damping=True)
        ekw.update(domain_penalty_weight=0.05, movement_penalty_weight=0.02)
    elif variant == "q3_pinball":
        kw.update(output_dim=3)
        ekw.update(loss="pinball", quantile_levels=[0.1, 0.5, 0.9], non_crossing_weight=0.3)
    elif variant == "b4096":
        B = 4096
    elif variant == "b33":
        B = 33
    rs = np.random.RandomState(sum(map(ord, variant)))       # deterministic per variant
    coords = torch.from_numpy(rs.uniform(-0.02, 1.02, (B, 2)).astype(np.float32)).to(d)
    t = torch.from_numpy(rs.uniform(0, 1, (B,)).astype(np.float32)).to(d)
    y = torch.from_numpy(rs.standard_normal((B, ycols)).astype(np.float32)).to(d)
    X = torch.from_numpy(rs.standard_normal((B, kw["p"])).astype(np.float32)).to(d) if kw["p"] else None
    res = []
    for dense in (False, True):
        torch.manual_seed(4)
        m = STInterpMLP(**kw).to(d)
        if kw.get("spatial_learnable"):
            with torch.no_grad():
                m.spatial_basis.centers.add_(0.004 * torch.randn_like(m.spatial_basis.centers))
        m.train()
        eng = TrainStep(m, lr=1e-3, grad_clip=10.0, ema_decay=0.99, max_batch=B, force_dense=dense, **ekw)
        assert eng.uses_window == (not dense)
        for _ in range(2):
            eng.step(X, coords, t, y)
        res.append((eng.flat.clone().cpu().numpy(), eng.mean_loss()))
    assert abs(res[0][1] - res[1][1]) <= 2e-5 * max(1.0, abs(res[1][1])), (res[0][1], res[1][1])
    err = rel_l2(res[0][0], res[1][0])
    assert err <= 5e-5, err


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_configs_window_vs_materialised(seed):
    """Seeded random model / batch shapes (levels, widths, covariates, outputs, LayerNorm, dropout off,
    learnable or fixed knots, batch size incl. > 4096): two engine steps on the window path against the
    materialising kernels."""
    from stnf.models import STInterpMLP
    from stnf.engine import TrainStep
    d = dev()
    rs = np.random.RandomState(1000 + seed)
    n_lev = int(rs.randint(1, 5))
    sides = sorted(int(s) for s in rs.choice([12, 16, 20, 24, 33, 40, 48], size=n_lev, replace=False))
    if sum(s * s for s in sides) < 1024:
        sides[-1] = 40
    h0 = int(rs.choice([128, 256]))
    hidden = [h0] + [int(h) for h in rs.choice([32, 64, 128, 256], size=int(rs.randint(0, 3)))]
    Q = int(rs.choice([1, 1, 2, 5]))
    kw = dict(p=int(rs.choice([0, 0, 2])), k_spatial_centers=[s * s for s in sides],
              k_temporal_centers=[int(k) for k in rs.choice([4, 7, 10, 15], size=int(rs.randint(1, 3)))],
              hidden_dims=hidden, dropout=0.0, layernorm=bool(rs.randint(0, 2)),
              spatial_basis_function=str(rs.choice(["wendland", "triangular"])), output_dim=Q,
              spatial_learnable=bool(rs.randint(0, 2)))
    B = int(rs.choice([17, 300, 1500, 4096, 5000, 9000]))
    coords = torch.from_numpy(rs.uniform(-0.01, 1.01, (B, 2)).astype(np.float32)).to(d)
    t = torch.from_numpy(rs.uniform(0, 1, (B,)).astype(np.float32)).to(d)
    y = torch.from_numpy(rs.standard_normal((B, Q)).astype(np.float32)).to(d)
    X = torch.from_numpy(rs.standard_normal((B, kw["p"])).astype(np.float32)).to(d) if kw["p"] else None
    res = []
    for dense in (False, True):
        torch.manual_seed(seed)
        m = STInterpMLP(**kw).to(d)
        m.train()
        eng = TrainStep(m, lr=1e-3, grad_clip=5.0, ema_decay=0.9, max_batch=B, force_dense=dense)
        assert eng.uses_window == (not dense), (kw, B)
        for _ in range(2):
            eng.step(X, coords, t, y)
        res.append((eng.flat.clone().cpu().numpy(), eng.mean_loss()))
    assert abs(res[0][1] - res[1][1]) <= 2e-5 * max(1.0, abs(res[1][1])), (kw, B, res[0][1], res[1][1])
    err = rel_l2(res[0][0], res[1][0])
    assert err <= 1e-4, (kw, B, err)


@pytest.mark.parametrize("switch", ["STDADK_NO_L1_TAIL", "STDADK_NO_TAIL_FWD_BWD", "STDADK_NO_DW_ALL"])
def test_launch_fusions_are_bitwise_neutral(switch, monkeypatch):
    """The fused launches run the same kernel bodies as the separate ones: switching a fusion off
    (library environment switch, read per call) leaves the parameters after three steps bit-identical."""
    from stnf.engine import TrainStep
    d = dev()
    cfg = cases.MODEL_CASES["c2_b257"]
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setenv(switch, "1")
        m = build_model(cfg, dropout=0.1)
        m.train()
        # no clipping: where the clip norm's partials are summed (inside the reductions launch, or a
        # separate pass when that launch is split up) changes its last bits, which is not the point here
        eng = TrainStep(m, lr=1e-3, ema_decay=0.99, max_batch=cfg["B"], grad_clip=0.0, seed=77)
        for _ in range(3):
            eng.step(None, coords, t, y)
        res.append(eng.flat.clone())
    monkeypatch.delenv(switch, raising=False)
    assert torch.equal(res[0], res[1])


# ------------------------------------------------------------------ sparsity penalties on the first layer
def build_sparsity_model(name):
    cfg, sp, zero_rows = cases.sparsity_cfg(name)
    m = build_model(cfg)
    with torch.no_grad():
        w = m._body[0].weight
        for r in zero_rows:
            w[:, cfg["p"] + r] = 0.0
    return m, cfg, sp, zero_rows


@pytest.mark.parametrize("w0_t", [True, False])
@pytest.mark.parametrize("kind,H,p,Ks,Kt,ap_s,ap_t", [
    ("element", 32, 0, 9, 5, True, True), ("group", 256, 3, 227, 70, True, False),
    ("sparse_group", 256, 0, 1031, 70, True, True), ("sparse_group", 128, 2, 70, 9, False, True),
    ("group", 48, 1, 5, 0, True, True), ("sparse_group", 30, 1, 9, 5, True, True),
    ("sparse_group", 512, 0, 300, 9, True, True), ("element", 1024, 1, 40, 7, True, True)])
def test_sparsity_kernel_matches_oracle(kind, H, p, Ks, Kt, ap_s, ap_t, w0_t):
    """stdadk_sparsity_f32 in both weight layouts: penalty values, gradient added INTO dW0 (with its scale),
    the loss accumulator; zero weights and an all-zero group take torch's sub-gradient 0; covariate
    columns untouched."""
    from stnf import _native as N
    rs = np.random.RandomState(H + Ks + Kt)
    D = p + Ks + Kt
    W = rs.standard_normal((H, D)).astype(np.float32) * 0.1
    W[:, p + 1] = 0.0                    # an all-zero group
    W[rs.randint(0, H, 40), rs.randint(p, D, 40)] = 0.0
    G0 = rs.standard_normal((H, D)).astype(np.float32)
    l1, lg, gs, ls = 0.013, 0.021, 0.5, 7.0
    ps, pt, dW = orc.sparsity_penalty(W, p, Ks, Kt, kind, l1, lg, ap_s, ap_t)
    d = dev()
    Wd = torch.from_numpy(W.T.copy() if w0_t else W).to(d)
    Gd = torch.from_numpy(G0.T.copy() if w0_t else G0.copy()).to(d)
    loss = torch.full((1,), 2.0, device=d)
    pen = torch.zeros(2, device=d)
    N.sparsity(N.make_sparsity(kind, l1, lg, ap_s, ap_t), Wd, Gd, w0_t, p, Ks, Kt, grad_scale=gs, loss_scale=ls,
               loss_sum=loss, penalties=pen)
    got = Gd.cpu().numpy().astype(np.float64)
    got = got.T if w0_t else got
    assert np.abs(got - (G0 + gs * dW)).max() <= 2e-6
    assert np.array_equal(got[:, :p], G0[:, :p].astype(np.float64))
    assert abs(pen[0].item() - ps) <= 1e-5 * max(1.0, ps) and abs(pen[1].item() - pt) <= 1e-5 * max(1.0, pt)
    applied = (ps if ap_s else 0.0) + (pt if ap_t else 0.0)
    assert abs(loss.item() - (2.0 + ls * applied)) <= 1e-5 * max(1.0, ls * applied)
    # values only (no gradient buffer), and kind 'none' is a no-op
    pen.zero_()
    N.sparsity(N.make_sparsity(kind, l1, lg, ap_s, ap_t), Wd, None, w0_t, p, Ks, Kt, penalties=pen)
    assert abs(pen[0].item() - ps) <= 1e-5 * max(1.0, ps)
    before = Gd.clone()
    N.sparsity(N.make_sparsity("none"), Wd, Gd, w0_t, p, Ks, Kt, penalties=pen)
    assert torch.equal(before, Gd)
    with pytest.raises(ValueError):
        N.make_sparsity("lasso")


@pytest.mark.parametrize("name", list(cases.SPARSITY_CASES))
def test_sparsity_module_loop_matches_reference(name):
    """The batch body written on the drop-in module: MSE + compute_sparsity_penalty terms, backward
    (train_st_interp.py:617-621,674-693) -- loss, penalties and every gradient against the float64 golden."""
    m, cfg, sp, _ = build_sparsity_model(name)
    g = load(name)
    d = dev()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    m.train()
    loss = torch.nn.MSELoss()(m(X if cfg["p"] else None, coords, t), y)
    pen = m.compute_sparsity_penalty(penalty_type=sp["kind"], lambda_l1=sp["lambda_l1"],
                                     lambda_group=sp["lambda_group"])
    if sp.get("apply_spatial", True):
        loss = loss + pen["spatial_penalty"]
    if sp.get("apply_temporal", True):
        loss = loss + pen["temporal_penalty"]
    loss.backward()
    assert abs(pen["spatial_penalty"].item() - float(g["spatial_penalty64"])) <= 1e-5 * max(1.0, float(g["spatial_penalty64"]))
    assert abs(pen["temporal_penalty"].item() - float(g["temporal_penalty64"])) <= 1e-5 * max(1.0, float(g["temporal_penalty64"]))
    assert abs(pen["total_penalty"].item() - float(g["spatial_penalty64"]) - float(g["temporal_penalty64"])) <= 2e-5 * max(1.0, float(g["spatial_penalty64"]))
    assert abs(loss.item() - float(g["loss64"])) <= 2e-5 * max(1.0, float(g["loss64"]))
    for k, p in m.named_parameters():
        check_vs_digest(p.grad.cpu().numpy(), g, "g", k, cfg["seed"] + 7, tol=2e-5)


@pytest.mark.parametrize("dense,whole", [(False, True), (True, True), (False, False)])
@pytest.mark.parametrize("name", list(cases.SPARSITY_CASES))
def test_sparsity_engine_steps_match_reference(name, dense, whole):
    """OPT['steps'] fused steps with the sparsity gradient added between backward and clipping, through the
    one-call step entry and through the split calls."""
    from stnf.engine import TrainStep
    m, cfg, sp, _ = build_sparsity_model(name)
    g = load(name)
    o = cases.OPT
    d = dev()
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, lr=o["lr"], weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"],
                    grad_clip=o["grad_clip"], ema_decay=o["ema_decay"], max_batch=cfg["B"], force_dense=dense,
                    sparsity_penalty_type=sp["kind"], sparsity_lambda_l1=sp["lambda_l1"],
                    sparsity_lambda_group=sp["lambda_group"],
                    sparsity_apply_to_spatial=sp.get("apply_spatial", True),
                    sparsity_apply_to_temporal=sp.get("apply_temporal", True))
    assert eng._whole_step
    eng._whole_step = whole
    losses = []
    for _ in range(o["steps"]):
        eng.step(X if cfg["p"] else None, coords, t, y)
        losses.append(eng.mean_loss())
    ref = g["opt_losses64"]
    assert np.abs(np.array(losses) - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), (losses, ref)
    for k, p in m.named_parameters():
        check_vs_digest(p.detach().cpu().numpy(), g, "p", k, cfg["seed"] + 7, tol=1e-4)
    eng.swap_in_ema()
    for k, p in m.named_parameters():
        check_vs_digest(p.detach().cpu().numpy(), g, "ema", k, cfg["seed"] + 7, tol=1e-4)
    eng.swap_in_ema()


FOUR_LEVELS = dict(p=2, k_spatial_centers=[64, 144, 256, 400], k_temporal_centers=[10, 15], hidden_dims=[256, 128],
                   layernorm=True, basis="wendland", output_dim=1, B=64, seed=71)


@pytest.mark.parametrize("name", ["tiny9_ln_p3", "default227", "default227_tri", "c2_b257", "c2_b257_noln",
                                  "four_levels_p2"])
def test_layer0_observation_groups_are_bit_identical(name, monkeypatch):
    """Layer-0 window forward with 2 cell-adjacent observations per wave (one fetch of a W0^T row feeds
    both; the default for large batches) against one observation per wave: every observation sums its
    own knots in the same order, so outputs and gradients are bit-identical -- including ragged groups at the
    end of a workgroup's rows, groups whose windows are too far apart to share a candidate box, covariates and
    a fourth level (a second chunk of levels)."""
    cfg = cases.MODEL_CASES[name] if name in cases.MODEL_CASES else FOUR_LEVELS
    d = dev()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    # more rows than the golden batch (an odd count: a ragged last group), a clustered half so that many groups share knots
    rs = np.random.RandomState(5)
    n = 1003
    c2 = torch.from_numpy(np.concatenate([rs.uniform(0, 1, (n // 2, 2)), 0.4 + 0.02 * rs.standard_normal((n - n // 2, 2))])
                          .astype(np.float32)).to(d)
    t2 = torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)).to(d)
    X2 = torch.from_numpy(rs.standard_normal((n, cfg["p"])).astype(np.float32)).to(d) if cfg["p"] else None
    y2 = torch.from_numpy(rs.standard_normal((n, 1)).astype(np.float32)).to(d)
    res = {}
    for grp in ("1", "2"):
        monkeypatch.setenv("STDADK_L1_GROUP", grp)
        m = build_model(cfg, dropout=0.1)
        m.eval()
        with torch.no_grad():
            ye = m(X2, c2, t2).clone()
        m.train()
        torch.manual_seed(11)                 # the module draws its dropout seed from torch's generator
        out = m(X2, c2, t2)
        torch.nn.functional.mse_loss(out, y2).backward()
        res[grp] = (ye, out.detach().clone(), [p.grad.clone() for p in m.parameters()])
    monkeypatch.delenv("STDADK_L1_GROUP", raising=False)
    for grp in ("2",):
        assert torch.equal(res[grp][0], res["1"][0]), grp
        assert torch.equal(res[grp][1], res["1"][1]), grp
        for ga, gb in zip(res[grp][2], res["1"][2]):
            assert torch.equal(ga, gb), grp


@pytest.mark.parametrize("name", ["default227", "default227_tri", "c2_b257", "c2_b257_noln", "four_levels"])
def test_knot_groups_are_bit_identical(name, monkeypatch):
    """Per-knot gather of dW0^T with two neighbouring knots per wave (one fetch of a dZ row feeds both; the
    default for fixed knots) against one knot per wave: every knot sums its own observations in the same
    order, so the gradients are bit-identical -- through the split backward (own kernel, module autograd) and
    through the engine step (merged weight-gradient kernel); odd grid sides leave the last knot of a row alone."""
    from stnf.engine import TrainStep
    cfg = cases.MODEL_CASES[name] if name in cases.MODEL_CASES else dict(FOUR_LEVELS, p=0)
    d = dev()
    rs = np.random.RandomState(6)
    n = 1501
    c2 = torch.from_numpy(np.concatenate([rs.uniform(-0.05, 1.05, (n // 2, 2)), 0.6 + 0.03 * rs.standard_normal((n - n // 2, 2))])
                          .astype(np.float32)).to(d)
    t2 = torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)).to(d)
    y2 = torch.from_numpy(rs.standard_normal((n, 1)).astype(np.float32)).to(d)
    res = {}
    for pairs in ("1", "2"):
        monkeypatch.setenv("STDADK_KNOTS_PER_WAVE", pairs)
        m = build_model(cfg)
        m.train()
        torch.nn.functional.mse_loss(m(None, c2, t2), y2).backward()
        grads = [p.grad.clone() for p in m.parameters()]
        m2 = build_model(cfg)
        eng = TrainStep(m2, lr=1e-3, ema_decay=0.99, max_batch=n, grad_clip=0.0, seed=77)
        for _ in range(2):
            eng.step(None, c2, t2, y2)
        res[pairs] = (grads, eng.flat.clone())
    monkeypatch.delenv("STDADK_KNOTS_PER_WAVE", raising=False)
    for nk in ("2",):
        for ga, gb in zip(res[nk][0], res["1"][0]):
            assert torch.equal(ga, gb), nk
        assert torch.equal(res[nk][1], res["1"][1]), nk


@pytest.mark.parametrize("name", ["tiny9_ln_p3", "default227", "default227_gauss", "default227_noln"])
def test_dense_layer0_inside_tail_launch(name, monkeypatch):
    """Materialising path, D <= 512: features + layer 0 evaluated inside the tail launch (one kernel from the raw
    observations to the activation gradients) against the separate kernels (stdadk_rbf_build_f32, the layer-0
    GEMM, its LayerNorm/ReLU/Dropout kernel): same feature arithmetic, another GEMM summation order."""
    from stnf.engine import TrainStep
    cfg = cases.MODEL_CASES[name]
    d = dev()
    rs = np.random.RandomState(8)
    n = 777                      # ragged against the 16-row tiles
    c2 = torch.from_numpy(rs.uniform(-0.05, 1.05, (n, 2)).astype(np.float32)).to(d)
    t2 = torch.from_numpy(rs.uniform(0, 1, (n, 1)).astype(np.float32)).to(d)
    X2 = torch.from_numpy(rs.standard_normal((n, cfg["p"])).astype(np.float32)).to(d) if cfg["p"] else None
    y2 = torch.from_numpy(rs.standard_normal((n, 1)).astype(np.float32)).to(d)
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("STDADK_NO_DENSE0_TAIL", "1")
        m = build_model(cfg, dropout=0.1)
        m.train()
        # lr = 0: the parameters stay put, so every step compares the two routes' GRADIENTS on identical weights
        # (with a real lr, Adam turns summation-order noise in near-zero gradient entries into lr-sized parameter
        #  differences that depend on the mask draw: 2e-6 .. 8e-5 after three steps)
        eng = TrainStep(m, lr=0.0, weight_decay=0.0, ema_decay=0.99, max_batch=n, force_dense=True, seed=0x5DEECE66D)
        assert not eng.uses_window
        grads = []
        for _ in range(3):
            eng.step(X2, c2, t2, y2)
            grads.append(eng.grad.clone())
        loss = eng.mean_loss()
        # the same comparison without dropout, where the float64 oracle can name the hidden units that sit on a ReLU kink
        m0 = build_model(cfg, dropout=0.0)
        m0.train()
        e0 = TrainStep(m0, lr=0.0, weight_decay=0.0, max_batch=n, force_dense=True)
        e0.step(X2, c2, t2, y2)
        shapes = {k: tuple(p_.shape) for k, p_ in m0.named_parameters()}
        first_w = next(k for k, p_ in m0.named_parameters() if p_ is m0._body[0].weight)
        g0 = {}
        for k, o, cnt in e0.offsets:
            v = e0.grad[o:o + cnt]
            # the engine stores the first weight (and its gradient) transposed, (in, out)
            v = v.view(shapes[k][1], shapes[k][0]).t() if k == first_w else v.view(shapes[k])
            g0[k] = v.cpu().numpy().copy()
        m.eval()
        with torch.no_grad():
            ye = m(X2, c2, t2).clone()        # engine-owned (in,out) storage: the eval forward takes the same path
        res.append((loss, grads, ye, g0))
    monkeypatch.delenv("STDADK_NO_DENSE0_TAIL", raising=False)
    # same masks on both routes (a mismatch would show at 1e-1 in the loss and the gradients); GEMM summation order
    # differs, and with it the side of a ReLU kink a hidden unit within rounding of zero lands on: such a unit moves the
    # gradients by its whole contribution (1e-4 .. 1e-3 at 777 rows) and y by nothing
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * max(1.0, abs(res[1][0]))
    for ga, gb in zip(res[0][1], res[1][1]):
        assert rel_l2(ga.cpu().numpy(), gb.cpu().numpy()) <= 5e-3
    assert rel_l2(res[0][2].cpu().numpy(), res[1][2].cpu().numpy()) <= 2e-6
    # without dropout: the two routes agree to 5e-6 once the units the float64 oracle puts within 1e-6 of a kink may
    # sit on either side in either route
    Xn = X2.cpu().numpy() if X2 is not None else np.zeros((n, 0), np.float32)
    _, _, _, alts = orc.train_step_grads(Xn, c2.cpu().numpy(), t2.cpu().numpy(), y2.cpu().numpy(), cases.make_state(cfg),
                                         cfg, kink_tol=1e-6)
    flipped, adj = orc.fit_kink_sides(res[0][3], {k: v.astype(np.float64) for k, v in res[1][3].items()}, alts, signed=True)
    worst = max(rel_l2(res[0][3][k], adj[k]) for k in adj)
    assert worst <= 5e-6, (worst, flipped, len(alts))


@pytest.mark.parametrize("name,S,T", [("c2_b257", 1003, 7), ("c2_b257_noln", 64, 1), ("default227", 300, 5),
                                      ("default227_tri", 2500, 4), ("c2_b257", 70001, 3)])
def test_predict_grid_equals_row_by_row(name, S, T):
    """Site x time prediction grid: layer 0 as a per-site row + a per-time row (stdadk_spatial_partial_f32,
    stdadk_temporal_partial_f32, stdadk_forward_parts_f32) against the ordinary forward on the expanded T*S
    rows -- the same sums in another order of addition.  The 227-knot models take the materialising path, where
    the per-site half comes from stdadk_rbf_build_f32 + one GEMM with the spatial rows of W0^T."""
    from stnf.engine import Predictor
    cfg = cases.MODEL_CASES[name]
    d = dev()
    m = build_model(cfg)
    if name.startswith("default227"):
        m.force_window_path = False
    m.eval()
    g = torch.Generator().manual_seed(S + T)
    coords = (torch.rand(S, 2, generator=g) * 1.1 - 0.05).to(d)
    tv = (torch.arange(T, dtype=torch.float32) / max(T - 1, 1)).to(d)
    pr = Predictor(m, chunk=32768)
    got = pr.predict_grid(coords, tv, max_rows=100000)
    assert got.shape == (T, S, 1)
    ref = pr.predict(coords.repeat(T, 1), tv.repeat_interleave(S)).view(T, S, 1)
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())


def test_training_trajectory_follows_the_cpu_port():
    """End to end: 24 optimiser steps on fresh batches of a KAUST-shaped synthetic field (C2 model, dropout off,
    lr 2e-2 with clipping, AdamW + EMA), the fused engine on the GPU against the oracle's torch-CPU port of the
    reference's batch body started from the same weights -- the per-step losses follow each other."""
    from oracle import torch_port as tp
    from stnf.models import STInterpMLP
    from stnf.engine import TrainStep
    cfg = dict(p=0, k_spatial_centers=[1024, 4096, 5184], k_temporal_centers=[10, 15, 45], hidden_dims=[256, 256, 128],
               layernorm=True, dropout=0.0, basis="wendland", output_dim=1)
    B, steps = 1024, 24
    g = torch.Generator().manual_seed(77)
    coords = torch.rand(B * steps, 2, generator=g)
    t = torch.randint(0, 100, (B * steps, 1), generator=g).float() / 99.0
    y = (torch.sin(4 * np.pi * coords[:, :1]) * torch.cos(3 * np.pi * coords[:, 1:]) * (1 + 0.5 * torch.sin(2 * np.pi * t))
         + 0.1 * torch.randn(B * steps, 1, generator=g))
    port = tp.PortModel(cfg, seed=0)
    tr = tp.PortTrainer(port, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.99)
    d = dev()
    m = STInterpMLP(p=0, k_spatial_centers=cfg["k_spatial_centers"], k_temporal_centers=cfg["k_temporal_centers"],
                    hidden_dims=cfg["hidden_dims"], dropout=0.0, layernorm=True)
    with torch.no_grad():
        for (k, p), (k2, v) in zip(m.named_parameters(), port.params.items()):
            assert k == k2 and p.shape == v.shape
            p.copy_(v.detach())
    m = m.to(d)
    m.train()
    eng = TrainStep(m, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.99, max_batch=B)
    cd, td, yd = coords.to(d), t.to(d), y.to(d)
    X0 = torch.zeros(B, 0)
    lg, lc = [], []
    for s in range(steps):
        sl = slice(s * B, (s + 1) * B)
        lc.append(tr.step(X0, coords[sl], t[sl], y[sl]))
        eng.step(None, cd[sl], td[sl], yd[sl])
        lg.append(eng.mean_loss())
    lg, lc = np.array(lg), np.array(lc)
    dev_rel = np.abs(lg - lc) / np.maximum(lc, 1e-3)
    print("losses (port, engine):", np.round(lc[[0, 5, 11, 23]], 5), np.round(lg[[0, 5, 11, 23]], 5), "max rel dev", dev_rel.max())
    assert lc[-1] < 0.7 * lc[0]                       # it learns
    assert dev_rel.max() <= 1e-2, dev_rel            # measured 2.6e-3 (rounding differences amplified by 24 Adam steps)
    # predictions of the two trained models on held-out points (parameters themselves are not compared: Adam's
    # m / sqrt(v) turns rounding-level differences of near-zero gradients into lr-sized steps of rarely touched
    # knot rows, which barely move the function)
    ch = torch.rand(2048, 2, generator=g)
    th = torch.randint(0, 100, (2048, 1), generator=g).float() / 99.0
    with torch.no_grad():
        yc = port.forward(torch.zeros(2048, 0), ch, th, train=False)
        m.eval()
        yg = m(None, ch.to(d), th.to(d)).cpu()
    rel = float(((yg - yc) ** 2).mean().sqrt() / (yc ** 2).mean().sqrt())
    print("held-out prediction rel RMS difference", rel)
    assert rel <= 5e-3, rel                         # measured 2.2e-4
