"""CPU-side tests: the C-ABI library loads and exports what include/stdadk.h declares, and the
host logic of the drop-in `stnf` package (knot tables, state_dict layout, config mapping, loader,
EMA, metrics, error behaviour) matches the reference's surface.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from golden import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    from stnf import _native as N
    hdr = open(os.path.join(ROOT, "include", "stdadk.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(stdadk_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert os.path.exists(N.LIB_PATH), "libstdadk.so not built (run __graft_entry__.build())"
    handle = ctypes.CDLL(N.LIB_PATH)
    for sym in declared:
        assert hasattr(handle, sym), f"{sym} declared in stdadk.h but not exported"
    assert declared == set(N.exported_symbols()), declared ^ set(N.exported_symbols())
    hdr_version = int(re.search(r"#define\s+STDADK_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert N.lib().stdadk_abi_version() == N.ABI_VERSION == hdr_version


def test_abi_struct_layout_matches_header():
    from stnf import _native as N
    # stdadk_mlp_desc: int32 n_hidden, in_dim, hidden[8], out_dim, layernorm; float ln_eps, dropout_p
    assert ctypes.sizeof(N.MlpDesc) == 4 * (2 + 8 + 2 + 2)
    assert ctypes.sizeof(N.MlpTensors) == 8 * (9 + 9 + 8 + 8 + 9 + 9)      # + W_bf16, WT_bf16
    assert ctypes.sizeof(N.BF16Region) == 8 + 4 + 4 + 8 + 8 and ctypes.sizeof(N.BF16Shadow) == 8 + 8 * 32
    # stdadk_basis_desc: 3 + 8 int32 (padded to 48), 2 int64, 4 pointers
    assert ctypes.sizeof(N.BasisDesc) == 48 + 16 + 32
    d = N.make_desc(297, [256, 256, 128], 1, True, 0.1)
    assert N.lib().stdadk_mlp_workspace_bytes(ctypes.byref(d), 4096) > 4096 * (256 + 256 + 128) * 2 * 4
    bad = N.make_desc(297, [2000], 1, True, 0.0)
    assert N.lib().stdadk_mlp_workspace_bytes(ctypes.byref(bad), 16) == 0


def test_missing_library_fails_loudly(monkeypatch):
    from stnf import _native as N
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", "/nonexistent/libstdadk.so")
    with pytest.raises(N.NativeLibraryError):
        N.lib()


def test_native_library_refuses_host_tensors():
    """The HIP library and the fused engine have no host fallback: host tensors raise (the module's own plain-torch
    host path, tested in test_cpu_module_path.py, is a separate, explicit implementation)."""
    from stnf import _native as N
    from stnf.engine import TrainStep
    from stnf.models import STInterpMLP
    m = STInterpMLP(k_spatial_centers=[9], k_temporal_centers=[5], hidden_dims=[32, 16], dropout=0.0)
    with pytest.raises(RuntimeError, match="HIP device"):
        TrainStep(m)
    with pytest.raises(RuntimeError, match="HIP device"):
        N.rbf_build(torch.rand(4, 2), None, None, m.spatial_basis.centers, m.spatial_basis.bandwidths, "wendland",
                    None, None, torch.empty(4, 9))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "st-dadk_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("checker", ""), f"{f} mentions the oracle"
                assert "/root/reference" not in src


def test_knot_buffers_bit_exact_with_reference():
    from stnf.models.st_interp import SpatialBasisEmbedding, TemporalBasisEmbedding
    g = np.load(os.path.join(GOLD, "knots.npz"))
    sb = SpatialBasisEmbedding(n_centers=[25, 81, 121])
    assert np.array_equal(sb.centers.numpy(), g["ml_centers"])
    assert np.array_equal(sb.bandwidths.numpy(), g["ml_bw"])
    assert sb.k == 227 and sb.level_sides == [5, 9, 11]
    for side in (3, 5, 9, 11):
        s1 = SpatialBasisEmbedding(n_centers=[side * side])
        assert np.array_equal(s1.centers.numpy(), g[f"s{side}_centers"])
    tb = TemporalBasisEmbedding(n_centers=[10, 15, 45])
    assert np.array_equal(tb.centers.numpy(), g["tml_centers"])
    assert np.array_equal(tb.bandwidths.numpy(), g["tml_bw"])
    assert tb.k_time == 70


def test_constructor_errors_match_reference():
    from stnf.models.st_interp import SpatialBasisEmbedding
    with pytest.raises(ValueError):
        SpatialBasisEmbedding(basis_function="cubic")
    with pytest.raises(ValueError):
        SpatialBasisEmbedding(init_method="nope")
    with pytest.raises(AssertionError):
        SpatialBasisEmbedding(n_centers=[10])
    with pytest.raises(AssertionError):
        SpatialBasisEmbedding(init_method="gmm", train_coords=None)


@pytest.mark.parametrize("name", ["tiny9_ln_p3", "default227", "default227_noln", "c2_b257"])
def test_state_dict_layout(name):
    """Keys/shapes of SURVEY.md §8(b): buffers + mlp.<i>.{weight,bias} in Sequential order."""
    from stnf.models import STInterpMLP
    cfg = cases.MODEL_CASES[name]
    m = STInterpMLP(p=cfg["p"], k_spatial_centers=cfg["k_spatial_centers"],
                    k_temporal_centers=cfg["k_temporal_centers"], hidden_dims=cfg["hidden_dims"],
                    dropout=0.0, layernorm=cfg["layernorm"])
    sd = m.state_dict()
    Ks, Kt = sum(cfg["k_spatial_centers"]), sum(cfg["k_temporal_centers"])
    assert tuple(sd["spatial_basis.centers"].shape) == (Ks, 2)
    assert tuple(sd["spatial_basis._bandwidths"].shape) == (Ks,)
    assert tuple(sd["temporal_basis.centers"].shape) == (Kt,)
    assert tuple(sd["temporal_basis.bandwidths"].shape) == (Kt,)
    expect = {k: shp for k, shp, _ in cases.state_layout(cfg)}
    got = {k: tuple(v.shape) for k, v in sd.items() if k.startswith("mlp.")}
    assert got == expect
    assert [n for n, _ in m.named_parameters()] == list(expect)
    assert m.k_spatial == Ks and m.k_temporal == Kt and m.p == cfg["p"]
    assert m.last_hidden_dim == cfg["hidden_dims"][-1] and m.mlp_trunk is None


def test_state_dict_with_dropout_shifts_indices():
    from stnf.models import STInterpMLP
    m = STInterpMLP(dropout=0.1, layernorm=True)
    keys = [k for k in m.state_dict() if k.startswith("mlp.") and k.endswith("weight")]
    assert keys == ["mlp.0.weight", "mlp.1.weight", "mlp.4.weight", "mlp.5.weight", "mlp.8.weight",
                    "mlp.9.weight", "mlp.12.weight"]
    assert sum(p.numel() for p in m.parameters()) == 176385        # SURVEY.md §8 "ref default"


def test_create_model_config_mapping():
    from stnf.models import create_model
    m = create_model({"regression_type": "mean", "k_spatial_centers": [9], "k_temporal_centers": [5],
                      "hidden_dims": [32, 16], "dropout": 0.0, "layernorm": False, "p_covariates": 2})
    assert m.output_dim == 1 and m.p == 2 and m.input_dim == 2 + 9 + 5
    m5 = create_model({"regression_type": "multi-quantile", "quantile_levels": [0.05, 0.25, 0.5, 0.75, 0.95]})
    assert m5.output_dim == 5 and m5.mlp[-1].out_features == 5
    # learnable knots (reference :94-107): parameters, log-bandwidths, initial-position buffer, penalties
    m6 = create_model({"spatial_learnable": True, "gradient_damping": True, "damping_threshold": 0.0,
                       "damping_strength": 5.0})
    sb = m6.spatial_basis
    names = [k for k, _ in m6.named_parameters()]
    assert names[:2] == ["spatial_basis.centers", "spatial_basis.log_bandwidths"]
    assert "spatial_basis.centers_init" in m6.state_dict() and "spatial_basis._bandwidths" not in m6.state_dict()
    assert torch.equal(sb.bandwidths.detach(), torch.exp(sb.log_bandwidths.detach()))
    assert float(sb.compute_domain_penalty()) == 0.0 and float(sb.compute_movement_penalty()) == 0.0
    with torch.no_grad():
        sb.centers[0] = torch.tensor([-0.1, 1.2])
    assert abs(float(m6.compute_domain_penalty()) - (0.01 + 0.04)) < 1e-6
    assert abs(float(m6.compute_movement_penalty()) - (0.01 + 1.44)) < 1e-6      # knot 0 started at (0,0)
    g = sb._gradient_damping_hook(torch.ones(sb.k, 2))
    assert abs(float(g[0, 0]) - float(np.exp(-5.0 * np.sqrt(1.45)))) < 1e-6 and float(g[1, 0]) == 1.0
    with pytest.raises(NotImplementedError):
        create_model({"spatial_init_method": "kmeans_balanced"}, train_coords=np.random.rand(50, 2))


def test_sparsity_penalty_api():
    from stnf.models import STInterpMLP
    m = STInterpMLP(k_spatial_centers=[9], k_temporal_centers=[5], hidden_dims=[32, 16], dropout=0.0)
    w = m.mlp[0].weight.detach()
    for kind in ("element", "group", "sparse_group", "none"):
        out = m.compute_sparsity_penalty(kind, lambda_l1=0.3, lambda_group=0.7)
        assert set(out) == {"spatial_penalty", "temporal_penalty", "total_penalty"}
        assert float(out["total_penalty"]) >= 0
    out = m.compute_sparsity_penalty("sparse_group", lambda_l1=0.3, lambda_group=0.7)
    sp = w[:, :9]
    expect = 0.7 * sp.norm(2, dim=0).sum() + 0.3 * sp.abs().sum()
    assert abs(float(out["spatial_penalty"]) - float(expect)) < 1e-5
    with pytest.raises(ValueError):
        m.compute_sparsity_penalty("bogus")


def test_kaust_loader(tmp_path):
    from stnf.dataio.kaust_loader import load_kaust_csv_single
    p = tmp_path / "d.csv"
    p.write_text("x,y,t,z\n0.5,0.25,1,1.0\n0.1,0.9,1,2.0\n0.5,0.25,2,3.0\n0.7,0.7,3,5.0\n")
    z, coords, md = load_kaust_csv_single(str(p), normalize=False)
    assert z.shape == (3, 3) and z.dtype == np.float32 and coords.dtype == np.float32
    assert np.allclose(coords, [[0.5, 0.25], [0.1, 0.9], [0.7, 0.7]])
    assert z[0, 0] == 1 and z[0, 1] == 2 and z[1, 0] == 3 and z[2, 2] == 5
    assert np.isnan(z).sum() == 5 and md == {}
    zn, _, md = load_kaust_csv_single(str(p), normalize=True)
    vals = np.array([1, 2, 3, 5], dtype=np.float32)
    assert abs(md["z_mean"] - vals.mean()) < 1e-6 and abs(md["z_std"] - vals.std()) < 1e-6
    assert abs(zn[0, 0] - (1 - vals.mean()) / vals.std()) < 1e-6
    # a purely spatial file (KAUST 1a layout: no t column) is one time slice
    q = tmp_path / "s.csv"
    q.write_text("id_train,x,y,z\n0,0.5,0.25,1.0\n1,0.1,0.9,2.0\n2,0.7,0.7,5.0\n")
    z1, c1, _ = load_kaust_csv_single(str(q), normalize=False)
    assert z1.shape == (1, 3) and np.allclose(z1[0], [1, 2, 5]) and c1.shape == (3, 2)


def test_ema_and_metrics_host_logic():
    from stnf.utils import ModelEMA, compute_metrics, set_seed
    set_seed(3)
    lin = torch.nn.Linear(4, 3)
    ema = ModelEMA(lin, decay=0.9)
    w0 = lin.weight.data.clone()
    lin.weight.data += 1.0
    ema.update(lin)
    assert torch.allclose(ema.shadow["weight"], 0.9 * w0 + 0.1 * (w0 + 1.0), atol=1e-6)
    ema.apply_shadow()
    assert torch.allclose(lin.weight.data, 0.9 * w0 + 0.1 * (w0 + 1.0), atol=1e-6)
    ema.restore()
    assert torch.allclose(lin.weight.data, w0 + 1.0)
    sd = ema.state_dict()
    assert set(sd) == {"decay", "shadow"}
    mt = compute_metrics(np.array([1.0, 2.0, np.nan, 4.0]), np.array([1.5, 2.0, 3.0, 3.0]))
    assert abs(mt["mse"] - (0.25 + 0 + 1) / 3) < 1e-9 and set(mt) == {"rmse", "mae", "r2", "mse"}


def test_device_dataset_matches_reference_sample_order():
    """DeviceDataset.from_mask (vectorised) == the reference's create_dataset_from_mask semantics
    (scripts/train_st_interp.py:413-450): argwhere order, NaN targets skipped, t = t_idx/(T-1)."""
    from stnf.dataio import DeviceDataset
    rs = np.random.RandomState(0)
    T, S = 7, 11
    z = rs.standard_normal((T, S)).astype(np.float32)
    z[2, 3] = np.nan
    coords = rs.uniform(0, 1, (S, 2)).astype(np.float32)
    mask = rs.uniform(size=(T, S)) < 0.4
    mask[2, 3] = True
    ds = DeviceDataset.from_mask(z, coords, mask, device="cpu")
    exp = []
    for t_idx, s_idx in np.argwhere(mask):              # the reference's loop, restated
        if np.isnan(z[t_idx, s_idx]):
            continue
        exp.append((coords[s_idx, 0], coords[s_idx, 1], np.float32(t_idx / (T - 1)), z[t_idx, s_idx]))
    exp = np.array(exp, dtype=np.float32)
    got = torch.cat([ds.coords, ds.t, ds.y], 1).numpy()
    assert got.shape == exp.shape and np.array_equal(got, exp)
    one = DeviceDataset.from_mask(z[:1], coords, mask[:1], device="cpu")
    assert float(one.t.abs().max()) == 0.0               # T == 1 -> t = 0
    parts = ds.epoch_batches(4, shuffle=False)
    assert sum(p.numel() for p in parts) == len(ds) and parts[-1].numel() == len(ds) % 4 or len(ds) % 4 == 0


def test_n3_host_surface():
    """Loss descriptor layout, argument errors and the delta head's state_dict (reference
    st_interp.py:671-686) — host side only."""
    from stnf import _native as N
    from stnf.models import STInterpMLP
    # stdadk_loss_desc: int32 kind, y_cols; float tau[8], nc_weight; int32 nc_power
    assert ctypes.sizeof(N.LossDesc) == 4 * (2 + 8 + 2)
    # stdadk_adam_group: 5 pointers, int64 n, float lr (+pad), pointer, float max_norm (+pad), pointer, int32 (+pad),
    # pointer to the bf16 shadow table
    assert ctypes.sizeof(N.AdamGroup) == 96 and N.AdamGroup.lr_dev.offset == 56 and N.AdamGroup.n_parts.offset == 80
    assert N.AdamGroup.shadow.offset == 88
    # stdadk_optim_desc: 5 pointers, int64, float (+pad), pointer, 4 floats, pointer, float (+pad), pointer, float (+pad),
    # pointer to the bf16 shadow table, pointer to the non-finite guard word (ABI 7)
    assert ctypes.sizeof(N.OptimDesc) == 128 and N.OptimDesc.step_dev.offset == 80 and N.OptimDesc.sumsq_parts.offset == 96
    assert N.OptimDesc.shadow.offset == 112 and N.OptimDesc.nonfinite_step.offset == 120
    assert N.OptimDesc.shadow.offset == 112
    assert ctypes.sizeof(N.SparsityDesc) == 20 and N.SparsityDesc.apply_spatial.offset == 12
    # stdadk_knot_train: pointer, int32, 6 floats (+pad to 8)
    assert ctypes.sizeof(N.KnotTrain) == 40
    ld = N.make_loss("pinball", 3, 1, [0.1, 0.5, 0.9], 0.5, 2)
    assert (ld.kind, ld.y_cols, ld.nc_power) == (1, 1, 2) and abs(ld.tau[2] - 0.9) < 1e-7
    with pytest.raises(ValueError, match="quantile levels"):
        N.make_loss("pinball", 3, 1, [0.1, 0.5])
    with pytest.raises(ValueError, match="Unsupported power"):
        N.make_loss("pinball", 3, 1, [0.1, 0.5, 0.9], 1.0, 3)
    with pytest.raises(ValueError):
        N.make_loss("huber", 1)
    for name in cases.QUANTILE_CASES:
        cfg, _ = cases.quantile_cfg(name)
        if cfg["layernorm"] and cfg["hidden_dims"] == [256, 256, 128] and len(cfg["k_spatial_centers"]) == 3 \
                and cfg["k_spatial_centers"][0] > 100:
            continue        # the C2-sized model is covered on the GPU
        m = STInterpMLP(p=cfg["p"], k_spatial_centers=cfg["k_spatial_centers"],
                        k_temporal_centers=cfg["k_temporal_centers"], hidden_dims=cfg["hidden_dims"],
                        dropout=0.0, layernorm=cfg["layernorm"], output_dim=cfg["output_dim"],
                        use_delta_reparameterization=cfg["delta"])
        keys = [k for k in m.state_dict() if k.startswith(("mlp.", "mlp_trunk.", "delta_params."))]
        lay = cases.state_layout(cfg)
        assert keys == [k for k, _, _ in lay]
        assert all(tuple(m.state_dict()[k].shape) == shp for k, shp, _ in lay)
    from stnf import losses
    assert abs(losses.check_loss_numpy(np.array([1.0, 2.0, 3.0]), np.array([1.5, 2.5, 3.5]), 0.1) - 0.05) < 1e-15


def test_data_adaptive_initialisers_match_reference():
    """gmm / random_site knot tables (host-side one-offs, reference st_interp.py:187-343) against the
    reference's own output on the same points and numpy seed (tests/golden/make_golden.py init)."""
    from stnf.models.st_interp import SpatialBasisEmbedding
    from stnf.models import STInterpMLP
    pts = cases.init_points()
    g = np.load(os.path.join(GOLD, "init_known_answers.npz"))
    for method in ("gmm", "random_site"):
        np.random.seed(7)
        m = SpatialBasisEmbedding(n_centers=[9, 25], init_method=method, train_coords=pts)
        assert m.k == 34 and not m.learnable
        assert np.allclose(m.centers.numpy(), g[f"{method}_centers"], rtol=0, atol=1e-6), method
        assert np.allclose(m.bandwidths.numpy(), g[f"{method}_bw"], rtol=1e-6, atol=1e-7), method
    np.random.seed(7)
    m = SpatialBasisEmbedding(n_centers=[16], init_method="gmm", train_coords=np.concatenate([pts] * 4))
    assert np.allclose(m.centers.numpy(), g["gmm_big_centers"], atol=1e-6)
    assert np.allclose(m.bandwidths.numpy(), g["gmm_big_bw"], rtol=1e-6)
    # the shipped YAML's combination: GMM knots, learnable, multi-quantile head
    np.random.seed(7)
    mm = STInterpMLP(k_spatial_centers=[9, 25], k_temporal_centers=[5], hidden_dims=[32, 16], output_dim=5,
                     spatial_learnable=True, spatial_init_method="gmm", train_coords=pts, gradient_damping=True)
    assert [k for k, _ in mm.named_parameters()][:2] == ["spatial_basis.centers", "spatial_basis.log_bandwidths"]
    assert mm.spatial_basis.init_method == "gmm" and mm.spatial_basis.k == 34
    with pytest.raises(AssertionError):
        SpatialBasisEmbedding(n_centers=[9], init_method="gmm")
    with pytest.raises(ValueError):
        SpatialBasisEmbedding(n_centers=[9], init_method="bogus")


def test_graft_entry_build_imports_and_checks_abi():
    """__graft_entry__.build() (what the driver runs on CPU): compiles, imports the package and agrees
    with the header's ABI version."""
    import __graft_entry__ as g
    from stnf import _native as N
    g.build()
    hdr = open(os.path.join(ROOT, "include", "stdadk.h")).read()
    assert int(re.search(r"#define STDADK_ABI_VERSION (\d+)", hdr).group(1)) == N.ABI_VERSION


def test_spatial_metrics_and_print(capsys):
    """compute_spatial_metrics / print_metrics (reference stnf/utils/metrics.py:67-164) on a hand-checked case."""
    from stnf.utils import compute_spatial_metrics, print_metrics, compute_metrics
    coords = np.array([[0.0, 0.0], [0.3, 0.4], [0.6, 0.8], [0.6, 0.8]])          # |.| = 0, 0.5, 1, 1
    yt = np.zeros((1, 2, 4, 1)); yp = np.zeros((1, 2, 4, 1))
    yp[:, :, 1, :] = 2.0; yp[0, 0, 0, 0] = np.nan
    sm = compute_spatial_metrics(yt, yp, coords, n_bins=2)
    # rings [0,0.5) -> site 0 (one NaN dropped, error 0), [0.5,1.0) -> site 1 (error 2); |c| = 1 sits on the open edge
    assert sm['bin_centers'] == [0.25, 0.75]
    assert sm['rmse_by_distance'] == [0.0, 2.0] and sm['mae_by_distance'] == [0.0, 2.0]
    print_metrics(compute_metrics(np.array([1.0, 2.0, 3.0]), np.array([1.0, 2.0, 4.0])), prefix="Val")
    out = capsys.readouterr().out
    assert out.startswith("Val Metrics:") and "RMSE: 0.577350" in out and "R²:" in out


def test_public_header_is_plain_c():
    """include/stdadk.h is the drop-in boundary: it must parse as C99 and as C++ on its own (no torch / HIP types)."""
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, "include", "stdadk.h")
    for cc, args in (("gcc", ["-std=c99", "-x", "c"]), ("g++", ["-std=c++11", "-x", "c++"])):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not available")
        r = subprocess.run([cc, *args, "-fsyntax-only", "-Wall", "-Werror", hdr], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


# ------------------------------------------------------------------ bench.py: the record's host-side arithmetic
def test_bench_record_helpers(tmp_path, monkeypatch):
    """What bench.py prints is priced by these: SURVEY.md 8(d)'s per-observation figures, the whole-step floors,
    and the provenance gate of `roofline.traffic` (a PMC figure is quoted only for the workload / batch / dtype /
    kernel sources it was measured on)."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    H, Kt, D, B = [256, 256, 128], 70, 10374, 4096
    # 41 508 B per observation for the materialised features of C2
    kind, amount, unit = bench.kernel_work("rbf_build_kernel<4, 0>", B, D, H, 0, 0, Kt)
    assert (kind, unit) == ("hbm", "B") and amount == B * 41508
    # AdamW + EMA: five reads and four writes of 4 B per parameter
    assert bench.kernel_work("adamw_ema_kernel", B, D, H, 1000, 0, Kt) == ("hbm", 36000.0, "B")
    # the fused step kernel: forward and backward of the layers after the first + layer 0 on its non-zero pairs
    nnz = 57 * B
    tail = 2.0 * B * (256 * 256 + 128 * 256 + 128)
    kind, amount, _ = bench.kernel_work("l1_tail_kernel<4, true, 0, false, false>", B, D, H, 0, nnz, Kt)
    assert kind == "mfma" and amount == 2.0 * tail + 2.0 * (nnz + B * Kt) * 256
    assert bench.kernel_work("tail_fwd_bwd_kernel<16, 4, false, false>", B, D, H, 0, nnz, Kt)[1] == 2.0 * tail
    assert bench.kernel_work("some_other_kernel", B, D, H, 0, nnz, Kt) is None
    fl = bench.step_floors(B, H, Kt, 1, 2_756_097, 57.0)
    assert fl["hbm_bytes"] == 40.0 * 2_756_097 + B * (16.0 + 24.0 * 640)
    assert abs(fl["hbm_us"] - fl["hbm_bytes"] / 8e12 * 1e6) < 1e-9 and abs(fl["mfma_us"] - fl["flops"] / 157.3e12 * 1e6) < 1e-9
    # provenance gate, on a scratch copy of the tree layout
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles"); os.makedirs(tmp_path / "st-dadk_amd" / "csrc")
    (tmp_path / "st-dadk_amd" / "csrc" / "k.hip").write_text("// kernel v1\n")
    assert bench.pmc_traffic("l1_tail_kernel", "c2", B, "f32")[0] is None          # no file
    sha = bench.csrc_hash()
    rec = dict(source=dict(workload="c2", batch=B, dtype="f32", commit="abc", csrc_sha16=sha),
               kernels={"l1_tail_kernel": 68.0e6, "rbf_build_kernel@100": 1.0, "rbf_build_kernel@200": 2.0},
               kernel_names={"l1_tail_kernel": "l1_tail_kernel<4, true, 0, false, false>"})
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps(rec))
    val, src = bench.pmc_traffic("l1_tail_kernel(stdadk::", "c2", B, "f32")
    assert val == 68.0e6 and src["commit"] == "abc" and src["kernel"].startswith("l1_tail_kernel<4")
    assert bench.pmc_traffic("rbf_build_kernel", "c2", B, "f32", 200)[0] == 2.0
    assert bench.pmc_traffic("rbf_build_kernel", "c2", B, "f32", 300)[0] is None        # no entry for that grid
    for kw in (dict(workload="c4"), dict(B=8192), dict(dtype="bf16")):
        args = dict(workload="c2", B=B, dtype="f32"); args.update(kw)
        v, why = bench.pmc_traffic("l1_tail_kernel", args["workload"], args["B"], args["dtype"])
        assert v is None and why.startswith("stale")
    (tmp_path / "st-dadk_amd" / "csrc" / "k.hip").write_text("// kernel v2\n")            # any source change
    v, why = bench.pmc_traffic("l1_tail_kernel", "c2", B, "f32")
    assert v is None and "csrc_sha16" in why


def test_bench_self_launch_starts_the_ranks_and_relays_failures(tmp_path, capfd):
    """`python bench.py --gpus N` with WORLD_SIZE unset: bench.self_launch starts N rank processes with the rendezvous
    environment of torch.distributed.run (the parent never touches a GPU), rank 0 owns stdout, a failing rank stops
    the others and its exit code is returned (the driver's N > 1 command died before any GPU call)."""
    import bench
    stub = tmp_path / "rank_stub.py"
    stub.write_text(
        "import os, sys, time\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "assert int(os.environ['MASTER_PORT']) > 0 and os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'\n"
        "print('{\"rank\": %d, \"world\": %d, \"argv\": \"%s\"}' % (r, w, ' '.join(sys.argv[1:])))\n"
        "if '--fail' in sys.argv and r == 1: sys.exit(7)\n"
        "if '--fail' in sys.argv: time.sleep(60)\n")
    assert bench.self_launch(3, ["--gpus", "3", "--steps", "5"], script=str(stub)) == 0
    out, err = capfd.readouterr()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines == ['{"rank": 0, "world": 3, "argv": "--gpus 3 --steps 5"}']          # ONE line on stdout: rank 0's
    assert '"rank": 1' in err and '"rank": 2' in err
    t0 = __import__("time").perf_counter()
    assert bench.self_launch(2, ["--fail"], script=str(stub)) == 7
    assert __import__("time").perf_counter() - t0 < 30                                 # rank 0 was not waited for
