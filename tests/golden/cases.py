"""Golden-vector case definitions shared by the generator (make_golden.py, runs only where
/root/reference exists) and by the parity tests (run anywhere).

Everything here is numpy-only and deterministic: inputs and weights come from
``numpy.random.RandomState`` (MT19937 — stable across numpy versions), never from torch's RNG,
so the GPU box can rebuild the *inputs* of every golden case without the reference and without
shipping multi-megabyte state_dicts.  Only the reference's *outputs* are stored in the .npz files.

Case vocabulary follows SURVEY.md §8(c): G1 knot tables, G2 phi, G3 psi, G4 feature column order,
G5 y/loss/grads, G6 one optimiser step.
"""
import math
import numpy as np

KNOT_SIDES = [3, 5, 9, 11, 32, 64, 72, 100]          # G1
TEMPORAL_NS = [1, 5, 10, 15, 45]                      # G3 / A4

# name -> model config + batch.  `mlp_idx` bookkeeping is derived in state_layout().
MODEL_CASES = {
    # reference test-suite config (direct-cdist regime: B<=25 and K<=25)
    "tiny9": dict(p=0, k_spatial_centers=[9], k_temporal_centers=[5], hidden_dims=[32, 16],
                  layernorm=False, basis="wendland", output_dim=1, B=4, seed=11),
    "tiny9_ln_p3": dict(p=3, k_spatial_centers=[9], k_temporal_centers=[5], hidden_dims=[32, 16],
                        layernorm=True, basis="wendland", output_dim=1, B=7, seed=12),
    # shipped default sizes (K_s=227, K_t=70, D=297, 176 385 params)
    "default227": dict(p=0, k_spatial_centers=[25, 81, 121], k_temporal_centers=[10, 15, 45],
                       hidden_dims=[256, 256, 128], layernorm=True, basis="wendland",
                       output_dim=1, B=193, seed=13),
    "default227_noln": dict(p=0, k_spatial_centers=[25, 81, 121], k_temporal_centers=[10, 15, 45],
                            hidden_dims=[256, 256, 128], layernorm=False, basis="wendland",
                            output_dim=1, B=130, seed=14),
    "default227_gauss": dict(p=2, k_spatial_centers=[25, 81, 121], k_temporal_centers=[10, 15, 45],
                             hidden_dims=[256, 256, 128], layernorm=True, basis="gaussian",
                             output_dim=1, B=65, seed=15),
    "default227_tri": dict(p=0, k_spatial_centers=[25, 81, 121], k_temporal_centers=[10, 15, 45],
                           hidden_dims=[256, 256, 128], layernorm=True, basis="triangular",
                           output_dim=1, B=65, seed=16),
    # BASELINE config C2/C3 model (K_s=10 304, D=10 374, 2 756 097 params), ragged batch
    "c2_b257": dict(p=0, k_spatial_centers=[1024, 4096, 5184], k_temporal_centers=[10, 15, 45],
                    hidden_dims=[256, 256, 128], layernorm=True, basis="wendland",
                    output_dim=1, B=257, seed=17),
    "c2_b257_noln": dict(p=0, k_spatial_centers=[1024, 4096, 5184], k_temporal_centers=[10, 15, 45],
                         hidden_dims=[256, 256, 128], layernorm=False, basis="wendland",
                         output_dim=1, B=257, seed=18),
}

# N3 cases: quantile / multi-quantile objectives and the delta head, on top of the
# model cases above.  `loss`: taus, prediction-level non-crossing (nc_weight, nc_power), or the
# delta-reparameterised head with the parameter-level penalty (delta, nc_lambda).
TAUS5 = [0.05, 0.25, 0.5, 0.75, 0.95]
QUANTILE_CASES = {
    "tiny9_q90": dict(base="tiny9", output_dim=1, seed=31, loss=dict(taus=[0.9])),
    "tiny9_mq5_nc1": dict(base="tiny9", output_dim=5, seed=32,
                          loss=dict(taus=TAUS5, nc_weight=0.5, nc_power=1)),
    "tiny9_mq5_nc2": dict(base="tiny9_ln_p3", output_dim=5, seed=33,
                          loss=dict(taus=TAUS5, nc_weight=2.0, nc_power=2)),
    "tiny9_delta5": dict(base="tiny9", output_dim=5, seed=34,
                         loss=dict(taus=TAUS5, delta=True, nc_lambda=0.1)),
    "default227_q10": dict(base="default227", output_dim=1, seed=35, loss=dict(taus=[0.1])),
    "default227_mq5": dict(base="default227", output_dim=5, seed=36,
                           loss=dict(taus=TAUS5, nc_weight=0.5, nc_power=1)),
    "default227_delta5": dict(base="default227", output_dim=5, seed=37,
                              loss=dict(taus=TAUS5, delta=True, nc_lambda=0.05)),
    "c2_b257_mq3": dict(base="c2_b257", output_dim=3, seed=38,
                        loss=dict(taus=[0.1, 0.5, 0.9], nc_weight=1.0, nc_power=2)),
}


def quantile_cfg(name):
    """Model config of a QUANTILE_CASES entry (base case + overrides) and its loss spec."""
    q = QUANTILE_CASES[name]
    cfg = dict(MODEL_CASES[q["base"]])
    cfg.update(output_dim=q["output_dim"], seed=q["seed"], delta=bool(q["loss"].get("delta")))
    return cfg, q["loss"]


# N2 cases: learnable knots.  The knots start from the uniform grid PLUS a
# deterministic perturbation (so movement, damping and the domain penalty are all active in the
# single-gradient golden); `centers_init` stays the unperturbed grid, as after some training.
LEARN_CASES = {
    "tiny9_learn": dict(base="tiny9", seed=41, knots=dict(gradient_damping=False)),
    "default227_learn": dict(base="default227", seed=42,          # the shipped DA-STDK settings
                             knots=dict(gradient_damping=True, damping_threshold=0.0, damping_strength=5.0,
                                        domain_penalty_weight=0.01, movement_penalty_weight=0.0)),
    "default227_learn_gauss": dict(base="default227_gauss", seed=43,
                                   knots=dict(gradient_damping=True, damping_threshold=0.02,
                                              damping_strength=1.0, domain_penalty_weight=0.5,
                                              movement_penalty_weight=0.1)),
    "default227_learn_tri": dict(base="default227_tri", seed=44, knots=dict(gradient_damping=False,
                                                                            movement_penalty_weight=0.05)),
    "c2_b257_learn": dict(base="c2_b257", seed=45,
                          knots=dict(gradient_damping=True, damping_threshold=0.0, damping_strength=5.0,
                                     domain_penalty_weight=0.01)),
}
BASIS_LR_RATIO = 0.05        # train_st_interp.py:475
BASIS_CLIP_RATIO = 0.1       # train_st_interp.py:703


# Sparsity penalties on the first layer's weights (st_interp.py:724-825, added to the loss by the batch
# body at train_st_interp.py:674-691): element-wise L1, group lasso per basis function, or both.
# `zero_rows`: basis columns of mlp.0.weight set to exactly 0 before the run (the sub-gradient torch takes
# there is 0 for both |w| and the row norm) -- counted from the first spatial column.
SPARSITY_CASES = {
    "tiny9_sp_element": dict(base="tiny9", seed=51, zero_rows=[2],
                             sp=dict(kind="element", lambda_l1=0.01, lambda_group=0.0)),
    "tiny9_sp_group": dict(base="tiny9_ln_p3", seed=52, zero_rows=[0, 11],
                           sp=dict(kind="group", lambda_l1=0.0, lambda_group=0.02, apply_temporal=False)),
    "default227_sp_sg": dict(base="default227", seed=53, zero_rows=[5, 290],
                             sp=dict(kind="sparse_group", lambda_l1=1e-3, lambda_group=1e-2)),
    "c2_b257_sp_sg": dict(base="c2_b257", seed=54, zero_rows=[7, 4000, 10350],
                          sp=dict(kind="sparse_group", lambda_l1=1e-4, lambda_group=1e-3,
                                  apply_spatial=True, apply_temporal=True)),
}


def sparsity_cfg(name):
    q = SPARSITY_CASES[name]
    cfg = dict(MODEL_CASES[q["base"]])
    cfg.update(seed=q["seed"])
    return cfg, q["sp"], q["zero_rows"]


def sparsity_state(cfg, zero_rows):
    """make_state(cfg) with the listed basis columns of the first Linear zeroed."""
    st = make_state(cfg)
    k0 = next(iter(st))
    for r in zero_rows:
        st[k0][:, cfg["p"] + r] = 0.0
    return st


def learn_cfg(name):
    q = LEARN_CASES[name]
    cfg = dict(MODEL_CASES[q["base"]])
    cfg.update(seed=q["seed"], learnable=True)
    return cfg, q["knots"]


def knot_perturbation(cfg):
    """(d_centers (Ks,2), d_log_bw (Ks,)) float32, added to the grid knots / log-bandwidths in fp32."""
    rs = np.random.RandomState(cfg["seed"] + 2000)
    Ks = sum(cfg["k_spatial_centers"])
    side0 = int(math.isqrt(cfg["k_spatial_centers"][-1]))
    step = 1.0 / max(side0 - 1, 1)
    dc = (0.3 * step * rs.standard_normal((Ks, 2))).astype(np.float32)
    dlb = (0.1 * rs.standard_normal(Ks)).astype(np.float32)
    return dc, dlb


def init_points():
    """Clustered training coordinates for the data-adaptive initialisers (with duplicates, as the
    driver passes them: one row per observation)."""
    rs = np.random.RandomState(123)
    blobs = [rs.normal(loc=c, scale=s, size=(n, 2)) for c, s, n in
             [((0.25, 0.3), 0.06, 900), ((0.7, 0.65), 0.10, 1400), ((0.5, 0.1), 0.03, 300)]]
    pts = np.clip(np.concatenate(blobs + [rs.uniform(0, 1, (400, 2))]), 0.0, 1.0).astype(np.float32)
    return np.concatenate([pts, pts[:500]])



# Cases small enough that the fp32 reference arrays are stored next to the float64 truth.
FULL_CASES = ["tiny9", "tiny9_ln_p3"]
# Tensors with more elements than this are stored as digests (samples + row/col sums + norm).
DIGEST_ABOVE = 20000

CALIBRATION = {"wendland": 1.0, "gaussian": 0.223477, "triangular": 0.654714}


def make_inputs(cfg):
    """Deterministic (X, coords, t, y) float32 arrays for a case, with the edge rows first:
    on-knot points, points at exactly r = 1 from a knot, out-of-domain points."""
    rs = np.random.RandomState(cfg["seed"])
    B, p = cfg["B"], cfg["p"]
    coords = rs.uniform(0.0, 1.0, size=(B, 2)).astype(np.float32)
    side0 = int(math.isqrt(cfg["k_spatial_centers"][0]))
    bw0 = np.float32(2.5 * (1.0 / (side0 - 1) if side0 > 1 else 1.0))
    edge = np.array([
        [0.0, 0.0], [1.0, 1.0], [0.5, 0.5],            # on knots of every odd-sided grid
        [min(float(bw0), 1.0), 0.0],                   # r == 1 from knot (0,0) at level 0
        [-0.1, 0.3], [1.2, 1.05],                      # outside [0,1]^2
    ], dtype=np.float32)
    n_edge = min(len(edge), max(B - 1, 0))
    coords[:n_edge] = edge[:n_edge]
    T = 100
    t = (rs.randint(0, T, size=(B, 1)).astype(np.float32) / np.float32(T - 1)).astype(np.float32)
    if B > 2:
        t[0, 0], t[1, 0] = 0.0, 1.0
    x, yy = coords[:, 0].astype(np.float64), coords[:, 1].astype(np.float64)
    y = (np.sin(4 * np.pi * x) * np.cos(3 * np.pi * yy) * (1 + 0.5 * np.sin(2 * np.pi * t[:, 0]))
         + 0.1 * rs.standard_normal(B)).astype(np.float32).reshape(B, 1)
    X = rs.standard_normal((B, p)).astype(np.float32)
    return X, coords, t, y


def state_layout(cfg):
    """[(key, shape, kind)] for the trainable tensors in nn.Sequential index order
    (st_interp.py:656-690: Linear [, LayerNorm], ReLU [, Dropout]; dropout=0 in all cases)."""
    D = cfg["p"] + sum(cfg["k_spatial_centers"]) + sum(cfg["k_temporal_centers"])
    out = []
    idx, prev = 0, D
    for h in cfg["hidden_dims"]:
        out.append((f"mlp.{idx}.weight", (h, prev), "lin_w"))
        out.append((f"mlp.{idx}.bias", (h,), "lin_b"))
        idx += 1
        if cfg["layernorm"]:
            out.append((f"mlp.{idx}.weight", (h,), "ln_g"))
            out.append((f"mlp.{idx}.bias", (h,), "ln_b"))
            idx += 1
        idx += 1  # ReLU
        prev = h
    if cfg.get("delta"):
        # st_interp.py:671-686: shared trunk + one (d+1,) delta vector per quantile
        out = [(k.replace("mlp.", "mlp_trunk."), shp, kind) for k, shp, kind in out]
        for k in range(cfg["output_dim"]):
            out.append((f"delta_params.{k}", (prev + 1,), "delta"))
        return out
    out.append((f"mlp.{idx}.weight", (cfg["output_dim"], prev), "lin_w"))
    out.append((f"mlp.{idx}.bias", (cfg["output_dim"],), "lin_b"))
    return out


def make_state(cfg):
    """Deterministic float32 parameters {key: ndarray}; Linear ~ U(+-1/sqrt(fan_in)) as the torch
    default would give in distribution, LayerNorm gamma/beta perturbed away from (1, 0) so that
    their gradients are exercised."""
    rs = np.random.RandomState(cfg["seed"] + 1000)
    st = {}
    fan_in = None
    for key, shape, kind in state_layout(cfg):
        if kind == "lin_w":
            fan_in = shape[1]
            b = 1.0 / math.sqrt(fan_in)
            st[key] = rs.uniform(-b, b, size=shape).astype(np.float32)
        elif kind == "lin_b":
            b = 1.0 / math.sqrt(fan_in)
            st[key] = rs.uniform(-b, b, size=shape).astype(np.float32)
        elif kind == "delta":
            # mixed signs so that both branches of J(delta_k) and of the check loss are exercised
            v = 0.05 * rs.standard_normal(shape)
            v[0] = 0.15 * rs.standard_normal()
            st[key] = v.astype(np.float32)
        elif kind == "ln_g":
            st[key] = (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)
        else:
            st[key] = (0.05 * rs.standard_normal(shape)).astype(np.float32)
    return st


# Hyper-parameters of the G6 optimiser-step golden (train_st_interp.py:506-510,696-712; ema.py:52-66)
OPT = dict(lr=2e-2, weight_decay=5e-4, betas=(0.9, 0.999), eps=1e-8, grad_clip=10.0,
           ema_decay=0.99, steps=3)


def digest_positions(shape, n, seed):
    """n deterministic flat positions inside an array of `shape` (for big-gradient digests)."""
    rs = np.random.RandomState(seed)
    size = int(np.prod(shape))
    return np.sort(rs.randint(0, size, size=n)).astype(np.int64)
