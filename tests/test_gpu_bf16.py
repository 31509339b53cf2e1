"""BASELINE config C3: "bf16 MLP with MFMA" (STDADK_FLAG_BF16 / TrainStep(dtype="bf16") / model.compute_dtype).

The Linear layers AFTER the first take bf16 operands on the matrix cores; phi/psi, layer 0, the accumulators,
LayerNorm, the loss, master weights, gradients and optimiser state stay fp32 (SURVEY.md 7 "bf16 config").
Two kinds of check:

  * against an EMULATION of exactly that arithmetic in float64 (`emulate`, below: the same products with the
    operands rounded to bf16 where the kernels round them, everything else exact).  What is left is fp32
    accumulation order PLUS rounding flips: the kernels round fp32 activations, the emulation exact ones, and a
    value within fp32 error of a bf16 rounding boundary (about 5e-5 of all operands) lands one bf16 ulp (0.4 %)
    apart -- a few entries per batch; a flipped activation of size ~4 moves by 2^-5 and shifts a prediction by
    ~2e-3, and a row whose pre-activations moved by 1e-3 flips the ReLU mask of the units that close to zero.
    Achieved on MI355X: gradients 1e-7 .. 4e-5 rel-L2 in eight of the ten golden cases and at the full batches of
    4 096 / 20 000 rows, 3e-4 .. 3.8e-3 where a flip happened (C2 window path at B = 257, 9 000 rows); y 1e-8 ..
    2.3e-3.  Tolerances: y max-abs 5e-3, loss 5e-4 relative, gradients 1e-2 rel-L2 -- a wrong k mapping / operand
    layout / missing product shows up at O(1);
  * against the float64 GOLDENS of the real reference -- the error of the bf16 configuration itself.  bf16 keeps
    8 significant bits (unit roundoff 2^-9 = 2.0e-3 per operand); over K = 128..256 products with independent
    rounding errors a pre-activation is off by ~2^-9 sqrt(2/K) |a||w| ~ 3e-4 of its scale, and three layers and the
    backward chain (ReLU masks of units within that distance of zero flip) compound that.  Stated tolerances, with
    the values achieved on MI355X over the ten golden cases x paths (printed by the test, recorded in DESIGN.md):
    y max-abs 1e-2 max(1, max|y|) (achieved <= 3.7e-3), loss 1e-2 relative (<= 4.6e-3), gradients 1e-1 rel-L2 per
    tensor (<= 5.8e-2: mlp.1.weight of the 227-knot model at B = 257; 2.9e-2 on the C2 model).
"""
import os

import numpy as np
import pytest
import torch

from golden import cases
from oracle import stdadk_oracle as orc

import test_gpu_parity as T

pytestmark = pytest.mark.gpu

EMU_Y, EMU_LOSS, EMU_GRAD = 5e-3, 5e-4, 1e-2
Y_TOL, LOSS_TOL = 1e-2, 1e-2

# The bf16 configuration's error against the float64 reference, per case and per tensor, as MEASURED on the MI355X with
# the committed kernels (tests/golden/bf16_achieved.json, written by tools/bf16_error_table.py = this module's
# measurements dumped).  A test allows 2 x the achieved figure of its own tensor, not a blanket
# bound: the worst entry is 5.8e-2 (mlp.1.weight of the 227-knot model at 257 rows), most are 3e-3 .. 3e-2.
ACHIEVED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_achieved.json")
_MEASURED = {}        # filled by the tests of this run (what tools/bf16_error_table.py dumps)


def _achieved():
    import json
    if not os.path.exists(ACHIEVED_FILE):
        return None
    return json.load(open(ACHIEVED_FILE))


def check_against_achieved(case, errs, slack=2.0):
    """errs: {tensor or 'y' / 'loss': error vs the float64 reference}.  Recorded for the table; asserted against
    `slack` x the committed achieved figure (a tensor without a figure fails: regenerate the table)."""
    _MEASURED[case] = {k: float(v) for k, v in errs.items()}
    table = _achieved()
    if os.environ.get("STDADK_BF16_TABLE_WRITE"):
        return
    assert table is not None, "tests/golden/bf16_achieved.json missing: run tools/bf16_error_table.py on the MI355X box"
    assert case in table, f"no achieved bf16 errors recorded for {case}: run tools/bf16_error_table.py"
    for k, e in errs.items():
        if k in ("y", "loss"):
            continue        # recorded for the table; bounded by Y_TOL / LOSS_TOL (a signed sum of roundings: its
                            # achieved value can be anywhere below the bound, 2 x it means nothing)
        assert k in table[case], (case, k)
        assert e <= slack * table[case][k] + 1e-12, (case, k, e, table[case][k])


def dev():
    return T.dev()


def rb(x):
    """Round to bf16 (nearest even) and come back, in the dtype of x."""
    return x.float().bfloat16().to(x.dtype)


class _BfLinear(torch.autograd.Function):
    """z = rb(a) rb(W)^T + b with the backward the kernels run: dA = rb(dZ) rb(W);  dW = op(dZ)^T op(a) with
    op = rb when the weight-gradient products take bf16 operands too (`dw_bf`), else exact; db = colsum(dZ)."""

    @staticmethod
    def forward(ctx, a, W, b, dw_bf):
        ctx.save_for_backward(a, W)
        ctx.dw_bf = dw_bf
        return rb(a) @ rb(W).t() + b

    @staticmethod
    def backward(ctx, dz):
        a, W = ctx.saved_tensors
        dA = rb(dz) @ rb(W)
        dW = (rb(dz).t() @ rb(a)) if ctx.dw_bf else (dz.t() @ a)
        return dA, dW, dz.sum(0), None


def emulate(feat, state, cfg, y, dw_bf, taus=None):
    """float64 forward / MSE / backward of the MLP on given features with the bf16 operand rounding of the HIP
    path: layer 0 and the output layer exact, hidden layers l >= 1 through _BfLinear.  Returns y_pred, loss and
    the gradients keyed like the state dict."""
    ps = {k: torch.from_numpy(np.asarray(v, np.float64)).clone().requires_grad_(True) for k, v in state.items()}
    keys = list(ps.keys())
    a = torch.from_numpy(np.asarray(feat, np.float64))
    L = len(cfg["hidden_dims"])
    i = 0
    for l in range(L):
        W, b = ps[keys[i]], ps[keys[i + 1]]
        i += 2
        z = a @ W.t() + b if l == 0 else _BfLinear.apply(a, W, b, dw_bf)
        if cfg["layernorm"]:
            g, be = ps[keys[i]], ps[keys[i + 1]]
            i += 2
            z = torch.nn.functional.layer_norm(z, (z.shape[1],), g, be, 1e-5)
        a = torch.relu(z)
    yp = a @ ps[keys[i]].t() + ps[keys[i + 1]]
    loss = ((yp - torch.from_numpy(np.asarray(y, np.float64))) ** 2).mean()
    loss.backward()
    return yp.detach().numpy(), float(loss), {k: p.grad.numpy() for k, p in ps.items()}


def features64(cfg, m, X, coords, t):
    cen = m.spatial_basis.centers.cpu().numpy(); bw = m.spatial_basis.bandwidths.cpu().numpy()
    tc = m.temporal_basis.centers.cpu().numpy(); tb = m.temporal_basis.bandwidths.cpu().numpy()
    return orc.features(X, orc.spatial_basis(coords, cen, bw, cfg["basis"]), orc.temporal_basis(t, tc, tb), cfg["p"])


DW_BF = True     # the grouped weight-gradient products of the layers after the first take bf16 operands too


# ------------------------------------------------------------------ operand copies
def test_bf16_copies_round_to_nearest_even_and_follow_the_optimiser():
    """stdadk_bf16_shadow_refresh == torch's fp32 -> bf16 cast bit for bit (plain and transposed copies, NaN and
    infinities kept); the optimiser entry points rewrite the copies from the stepped values in the same launch."""
    from stnf import _native as N
    d = dev()
    rs = np.random.RandomState(0)
    flat = torch.from_numpy(rs.standard_normal(5000).astype(np.float32)).to(d)
    flat[100:104] = torch.tensor([float("nan"), float("inf"), -float("inf"), 0.0], device=d)
    flat[104] = 1.00390625        # exactly between two bf16 values: ties to even
    regs = [(96, 16, 32), (1024, 48, 16), (2000, 64, 36)]
    copies = [(torch.zeros(r, c, device=d, dtype=torch.bfloat16), torch.zeros(c, r, device=d, dtype=torch.bfloat16))
              for _, r, c in regs]
    sh = N.make_bf16_shadow([(o, r, c, cp[0], cp[1]) for (o, r, c), cp in zip(regs, copies)])
    N.bf16_shadow_refresh(flat, sh)
    for (o, r, c), (wb, wt) in zip(regs, copies):
        want = flat[o:o + r * c].view(r, c).bfloat16()
        assert torch.equal(wb.view(torch.int16), want.view(torch.int16))
        assert torch.equal(wt.view(torch.int16), want.t().contiguous().view(torch.int16))
    # AdamW + EMA with the table: copies == bf16(parameters after the step)
    flat[100:104] = 0.5
    g = torch.from_numpy(rs.standard_normal(5000).astype(np.float32)).to(d)
    m_, v_, e_ = torch.zeros_like(flat), torch.zeros_like(flat), flat.clone()
    N.adamw_ema(flat, g, m_, v_, e_, 1e-2, (0.9, 0.999), 1e-8, 1e-3, 1, ema_decay=0.9, shadow=sh)
    for (o, r, c), (wb, wt) in zip(regs, copies):
        want = flat[o:o + r * c].view(r, c).bfloat16()
        assert torch.equal(wb.view(torch.int16), want.view(torch.int16))
        assert torch.equal(wt.view(torch.int16), want.t().contiguous().view(torch.int16))
    with pytest.raises(RuntimeError):
        N.bf16_shadow_refresh(flat, N.make_bf16_shadow([(2, 4, 4, copies[0][0], copies[0][1])]))     # offset not 4-aligned
    with pytest.raises(RuntimeError):
        N.bf16_shadow_refresh(flat, N.make_bf16_shadow([(4990, 4, 4, copies[0][0], copies[0][1])]) if False else
                              N.make_bf16_shadow([(0, 4, 6, copies[0][0], copies[0][1])]))            # cols % 4


# ------------------------------------------------------------------ forward / backward
@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("name", ["tiny9", "default227", "default227_noln", "c2_b257", "c2_b257_noln"])
def test_bf16_forward_backward_matches_emulation_and_goldens(name, dense):
    cfg = cases.MODEL_CASES[name]
    g = T.load(name)
    d = dev()
    X, coords, t, y = cases.make_inputs(cfg)
    m = T.build_model(cfg)
    m.force_dense_path = dense
    m.compute_dtype = "bf16"
    m.train()
    args = [torch.from_numpy(a).to(d) for a in (X, coords, t)]
    yp = m(*args)
    loss = torch.nn.MSELoss()(yp, torch.from_numpy(y).to(d))
    loss.backward()
    got_y = yp.detach().cpu().numpy()
    state = cases.make_state(cfg)
    ye, le, ge = emulate(features64(cfg, m, X, coords, t), state, cfg, y, DW_BF)
    emu_y = np.abs(got_y - ye).max() / max(1.0, np.abs(ye).max())
    emu_g = {k: T.rel_l2(p.grad.cpu().numpy(), ge[k]) for k, p in m.named_parameters()}
    worst_e = max(emu_g.values())
    # the configuration's own error against the float64 goldens of the reference
    y64 = g["y64"]
    ey = np.abs(got_y - y64).max() / max(1.0, np.abs(y64).max())
    el = abs(loss.item() - float(g["loss64"])) / float(g["loss64"])
    eg = {}
    for k, p in m.named_parameters():
        gg = p.grad.cpu().numpy().astype(np.float64)
        if f"g64/{k}" in g:
            eg[k] = np.linalg.norm((gg - g[f"g64/{k}"]).ravel()) / float(g[f"gnorm64/{k}"])
        else:
            pos = cases.digest_positions(gg.shape, 2048, cfg["seed"] + 7)
            ref = g[f"g64s/{k}"]
            eg[k] = np.linalg.norm(gg.ravel()[pos] - ref) / max(np.linalg.norm(ref), 1e-30)
    print(f"bf16 vs float64 golden [{name}, {'dense' if dense else 'window'}]: y {ey:.2e}  loss {el:.2e}  "
          f"grads max {max(eg.values()):.2e} ({max(eg, key=eg.get)})   | vs emulation: y {emu_y:.1e} grads {worst_e:.1e}")
    assert emu_y <= EMU_Y and abs(loss.item() - le) <= EMU_LOSS * le
    for k, e in emu_g.items():
        assert e <= EMU_GRAD, (k, e)
    assert ey <= Y_TOL and el <= LOSS_TOL
    check_against_achieved(f"golden/{name}/{'dense' if dense else 'window'}", dict(eg, y=ey, loss=el))
    # eval mode: same forward without the saved tensors
    m.eval()
    with torch.no_grad():
        assert np.abs(m(*args).cpu().numpy() - got_y).max() <= 1e-6 * max(1.0, np.abs(got_y).max())
    # and the fp32 mode of the same module is untouched
    m.compute_dtype = "f32"
    with torch.no_grad():
        y32 = m(*args).cpu().numpy()
    assert np.abs(y32 - y64).max() <= 1e-5 * max(1.0, np.abs(y64).max())


@pytest.mark.parametrize("B", [4096, 9000, 20000])
def test_bf16_full_batches_match_emulation(B):
    """The one-launch step kernel (B <= 4096), 32-row and 64-row tile kernels of the C2 model: bf16 train
    forward + backward through the engine's split path against the emulation (features from the parity-checked
    fp32 feature builder)."""
    from stnf.engine import TrainStep
    cfg = dict(cases.MODEL_CASES["c2_b257"], B=B, seed=300 + B % 97)
    d = dev()
    X, coords, t, y = cases.make_inputs(cfg)
    m = T.build_model(cfg)
    m.train()
    eng = TrainStep(m, max_batch=B, dtype="bf16", world_size=2)       # split path: gradients stay in eng.grad
    c, tt, yy = (torch.from_numpy(a).to(d) for a in (coords, t, y))
    eng._enqueue_grads(None, c, tt.view(-1), yy, B, B)
    loss = eng.loss_sum.item() / B
    feat = m.build_features(None, c, tt)[:, :m.input_dim].double().cpu().numpy()
    state = cases.make_state(cfg)
    _, le, ge = emulate(feat, state, cfg, y, DW_BF)
    assert abs(loss - le) <= EMU_LOSS * le
    by = {n: (o, k) for n, o, k in eng.offsets}
    worst = 0.0
    for k, p in m.named_parameters():
        o, n = by[k]
        got = eng.grad[o:o + n].cpu().numpy().astype(np.float64)
        ref = ge[k]
        if k == "mlp.0.weight":
            ref = ref.T                        # the engine stores dW0 transposed
        e = T.rel_l2(got, ref.ravel())
        worst = max(worst, e)
        assert e <= EMU_GRAD, (k, e)
    print(f"bf16 full batch B={B}: gradients vs emulation, worst rel-L2 {worst:.1e}")
    # ... and against the float64 oracle of the reference's arithmetic (no operand rounding): the configuration's own
    # error at full batch size, per tensor, bounded by 2 x what was measured for it
    yo, lo, go = orc.train_step_grads(X, coords, t, y, state, cfg)
    errs = {"loss": abs(loss - lo) / lo}
    for k, p in m.named_parameters():
        o, n = by[k]
        got = eng.grad[o:o + n].cpu().numpy().astype(np.float64)
        ref = go[k].T if k == "mlp.0.weight" else go[k]
        errs[k] = T.rel_l2(got, ref.ravel())
    print(f"bf16 full batch B={B} vs float64 oracle: loss {errs['loss']:.2e}, gradients max "
          f"{max(v for k, v in errs.items() if k != 'loss'):.2e}")
    assert errs["loss"] <= LOSS_TOL
    check_against_achieved(f"full_batch/{B}", errs)


# ------------------------------------------------------------------ engine
@pytest.mark.parametrize("name", ["default227", "c2_b257"])
def test_bf16_engine_steps(name):
    """TrainStep(dtype='bf16'), three fused steps: the operand copies stay equal to bf16(master weights) through the
    whole-step optimiser path and through the split path (virtual ranks), the two paths agree with each other, and
    the first step's loss (before any update) is within the bf16 tolerance of the reference's.  (The goldens'
    lr = 2e-2 makes the losses jump 0.43 -> 4.1 -> 3.7 within three steps: a trajectory that amplifies ANY
    perturbation, so the parameters after three steps are compared between the two bf16 paths, not with the fp32
    golden; how bf16 training tracks fp32 training is test_bf16_training_tracks_fp32.)"""
    from stnf.engine import TrainStep
    import test_gpu_round2 as R2
    cfg = cases.MODEL_CASES[name]
    g = T.load(name)
    o = cases.OPT
    d = dev()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    res = []
    for split in (False, True):
        m = T.build_model(cfg)
        m.train()
        eng = TrainStep(m, lr=1e-3, weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"],
                        grad_clip=o["grad_clip"], ema_decay=o["ema_decay"], max_batch=cfg["B"], dtype="bf16",
                        world_size=2 if split else None)
        assert eng._whole_step == (not split)
        if split:
            losses = R2._virtual_steps(eng, None, coords, t, y, 100, o["steps"])
        else:
            losses = []
            for _ in range(o["steps"]):
                eng.step(None, coords, t, y)
                losses.append(eng.mean_loss())
        lins = m._linears()
        for l in range(1, len(cfg["hidden_dims"])):
            wb, wt = m._bf16_engine[l]
            want = lins[l].weight.detach().bfloat16()
            assert torch.equal(wb.view(torch.int16), want.view(torch.int16)), l
            assert torch.equal(wt.view(torch.int16), want.t().contiguous().view(torch.int16)), l
        res.append((losses, eng.flat.clone()))
        assert abs(losses[0] - float(g["opt_losses64"][0])) <= LOSS_TOL * float(g["opt_losses64"][0])
    e = T.rel_l2(res[0][1].cpu().numpy(), res[1][1].cpu().numpy())
    print(f"bf16 engine [{name}]: whole-step vs split path after {o['steps']} steps, parameters rel-L2 {e:.1e}; losses {res[0][0]}")
    assert e <= 2e-3 and np.abs(np.array(res[0][0]) - np.array(res[1][0])).max() <= 2e-3 * max(res[0][0])
    # swap_in_ema re-rounds the copies
    eng.swap_in_ema()
    for l in range(1, len(cfg["hidden_dims"])):
        want = m._linears()[l].weight.detach().bfloat16()
        assert torch.equal(m._bf16_engine[l][0].view(torch.int16), want.view(torch.int16))
    eng.swap_in_ema()


def test_bf16_training_tracks_fp32():
    """What the bf16 configuration is for: the C2 model trained on a KAUST-shaped synthetic field for 150 steps
    (fresh batches of 4096, the reference's AdamW settings at lr 2e-3, dropout off), once with fp32 and once with
    bf16 operands from the same initial weights.  Per-step training losses stay within 5 % of each other and the
    held-out RMSE of the two trained models within 2 %."""
    from stnf.engine import TrainStep
    import bench
    cfg = cases.MODEL_CASES["c2_b257"]
    d = dev()
    coords, t, y = bench.synth(160_000, 11, d)
    hc, ht, hy = bench.synth(20_000, 12, d)
    out = {}
    for dtype in ("f32", "bf16"):
        m = T.build_model(cfg)
        m.train()
        eng = TrainStep(m, lr=2e-3, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.99, max_batch=4096, dtype=dtype, seed=1)
        losses = []
        for i in range(150):
            idx = torch.arange(i * 1000, i * 1000 + 4096, device=d) % coords.shape[0]
            eng.step_indexed(coords, t, y, idx)
            losses.append(eng.mean_loss())
        m.eval()
        with torch.no_grad():
            rmse = float(((m(None, hc, ht) - hy) ** 2).mean().sqrt())
        out[dtype] = (np.array(losses), rmse)
    l32, l16 = out["f32"][0], out["bf16"][0]
    rel = np.abs(l16 - l32) / l32
    print(f"bf16 vs fp32 training: max per-step loss deviation {rel.max():.2e} (last {rel[-1]:.2e}); held-out RMSE "
          f"fp32 {out['f32'][1]:.4f} bf16 {out['bf16'][1]:.4f}; loss {l32[0]:.3f} -> {l32[-1]:.4f}")
    assert l32[-1] < 0.2 * l32[0]                       # it did train
    assert rel.max() <= 5e-2
    assert abs(out["bf16"][1] - out["f32"][1]) <= 2e-2 * out["f32"][1]


def test_bf16_learnable_knots_and_quantile_head_run_and_track_fp32():
    """The other objectives on bf16 operands: learnable knots (two AdamW groups: the operand copies ride in the MLP
    group's launch) and the delta head with 5 quantiles -- three steps at lr 1e-3: the losses follow the fp32
    engine's within the bf16 tolerance, and the copies stay current."""
    from stnf.engine import TrainStep
    d = dev()
    o = cases.OPT
    for kind in ("learn", "delta"):
        traj = []
        for dtype in ("f32", "bf16"):
            if kind == "learn":
                m, cfg, kn, _ = T.build_learn_model("c2_b257_learn")
                kw = dict(basis_lr_ratio=cases.BASIS_LR_RATIO, basis_clip_ratio=cases.BASIS_CLIP_RATIO,
                          domain_penalty_weight=kn.get("domain_penalty_weight", 0.0))
            else:
                m, cfg, lc = T.build_quantile_model("default227_delta5")
                kw = dict(loss="pinball", quantile_levels=lc["taus"], non_crossing_lambda=lc.get("nc_lambda", 0.0))
            m.train()
            X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
            eng = TrainStep(m, lr=1e-3, weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"],
                            grad_clip=o["grad_clip"], ema_decay=o["ema_decay"], max_batch=cfg["B"], dtype=dtype, **kw)
            losses = []
            for _ in range(o["steps"]):
                eng.step(None, coords, t, y)
                losses.append(eng.mean_loss())
            traj.append(np.array(losses))
            if dtype == "bf16":
                for l in range(1, len(cfg["hidden_dims"])):
                    want = m._linears()[l].weight.detach().bfloat16()
                    assert torch.equal(m._bf16_engine[l][0].view(torch.int16), want.view(torch.int16)), (kind, l)
        e = np.abs(traj[1] - traj[0]).max() / np.abs(traj[0]).max()
        print(f"bf16 vs fp32 engine losses over {o['steps']} steps [{kind}]: max deviation {e:.2e}  {traj[0]} {traj[1]}")
        assert e <= LOSS_TOL


def test_bf16_predictor_and_grid():
    """Forward-only callers on bf16 operands: Predictor.predict / predict_grid agree with the module's eval
    forward (same kernels) and stay within the bf16 tolerance of the fp32 predictions."""
    from stnf.engine import Predictor
    cfg = cases.MODEL_CASES["c2_b257"]
    d = dev()
    m = T.build_model(cfg)
    m.eval()
    g = torch.Generator().manual_seed(5)
    S, Tn = 3001, 4
    coords = torch.rand(S, 2, generator=g).to(d)
    tv = torch.linspace(0, 1, Tn).to(d)
    pr32 = Predictor(m, chunk=4096)
    y32 = pr32.predict_grid(coords, tv).clone()
    m.compute_dtype = "bf16"
    pr = Predictor(m, chunk=4096)
    yg = pr.predict_grid(coords, tv)
    yr = pr.predict(coords.repeat(Tn, 1), tv.repeat_interleave(S)).view(Tn, S, 1)
    # (another order of addition in layer 0 -> fp32-level differences in its activations -> bf16 rounding flips)
    assert (yg - yr).abs().max().item() <= EMU_Y * max(1.0, yr.abs().max().item())
    e = (yg - y32).abs().max().item() / max(1.0, y32.abs().max().item())
    print(f"bf16 vs fp32 prediction grid: max-abs {e:.2e}")
    assert 0 < e <= Y_TOL


def test_bf16_needs_the_fused_tail():
    """Hidden widths the fused tail kernels do not cover: the flag is refused loudly (no silent fp32 run)."""
    from stnf.models import STInterpMLP
    d = dev()
    m = STInterpMLP(k_spatial_centers=[9], k_temporal_centers=[5], hidden_dims=[40, 24], dropout=0.0).to(d)
    m.compute_dtype = "bf16"
    with pytest.raises(RuntimeError, match="BF16"):
        m(None, torch.rand(8, 2, device=d), torch.rand(8, 1, device=d))
