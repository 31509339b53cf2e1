"""GPU tests added in round 3 (all through the C ABI):

  * the SHARDED optimiser (reduce-scatter -> local clip partials -> AdamW/EMA on the slice -> all-gather) with two
    virtual ranks against the reference's union-batch goldens (G6), plain MSE and learnable knots (two parameter
    groups whose boundary does not coincide with the slice boundary), and with two REAL ranks over gloo;
  * replicas made identical at construction (broadcast from rank 0) when the ranks were seeded differently;
  * the non-finite guard: first bad step recorded on the device, `run_epoch(check_every=...)` stops there;
  * Predictor + ModelEMA.apply_shadow / restore (delta head, W0^T copy, bf16 operand copies) and a bf16 engine after
    load_state_dict;
  * a step on a batch other than the announced one (the side stream's binning may not race with it);
  * the DPP wave sum is covered by every parity test of the other files (it is the default build); the rotated
    K-chunk order of the tail kernels (a switch, off by default): on == off to rounding.
"""
import os

import numpy as np
import pytest
import torch

from golden import cases

import test_gpu_parity as T
import test_gpu_round2 as R2

pytestmark = pytest.mark.gpu
TOL = 1e-5


def dev():
    return T.dev()


# ------------------------------------------------------------------ sharded optimiser, two virtual ranks
def _virtual_sharded_steps(eng, X, coords, t, y, cut, steps):
    """As test_gpu_round2._virtual_steps, with the optimiser of each virtual rank confined to its slice: gradients
    summed (= what both ranks' slices hold after the reduce-scatter), each rank's clip partials summed (= the small
    all-reduce), each rank's AdamW on its slice of the shared flat buffer (= the all-gather)."""
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
        parts = torch.zeros_like(eng._sumsq_all)
        for r in range(2):
            eng.set_virtual_rank(r)
            eng._shard_sumsq()
            parts += eng._sumsq_all
        eng.step_dev -= 1                   # one engine played two ranks: the device step counter advanced twice
        eng._sumsq_all.copy_(parts)
        for r in range(2):
            eng.set_virtual_rank(r)
            eng._shard_adamw()
        eng._stepped(B)
        losses.append(eng.mean_loss())
    return losses


@pytest.mark.parametrize("name", ["default227", "c2_b257"])
def test_sharded_optimizer_virtual_ranks_match_union_batch_golden(name):
    """G6 goldens (three clip + AdamW + EMA steps of the reference on the union batch) through the sharded
    optimiser: global-norm clip from the summed per-slice partials, moments / EMA per slice."""
    from stnf.engine import TrainStep
    cfg = cases.MODEL_CASES[name]
    g = T.load(name)
    o = cases.OPT
    d = dev()
    m = T.build_model(cfg)
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, lr=o["lr"], weight_decay=o["weight_decay"], betas=o["betas"], eps=o["eps"],
                    grad_clip=o["grad_clip"], ema_decay=o["ema_decay"], max_batch=cfg["B"], world_size=2,
                    shard_optimizer=True)
    assert eng.shard and eng.flat.numel() == 2 * eng.chunk and eng.chunk % 32 == 0 and eng.m.numel() == eng.chunk
    losses = _virtual_sharded_steps(eng, X if cfg["p"] else None, coords, t, y, 100, o["steps"])
    ref = g["opt_losses64"]
    assert np.abs(np.array(losses) - ref).max() <= 5 * TOL * max(1.0, np.abs(ref).max())
    R2._check_params(m, eng, g, cfg, 2e-5)
    assert eng.first_nonfinite_step() is None


@pytest.mark.parametrize("name", ["default227_learn", "c2_b257_learn"])
def test_sharded_optimizer_virtual_ranks_learnable_knots(name):
    """Two parameter groups (knots | MLP) with their own clip norms and learning rates: the knot group lies inside
    rank 0's slice, so rank 0 steps two groups and rank 1 one."""
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
                    movement_penalty_weight=kn.get("movement_penalty_weight", 0.0), world_size=2,
                    shard_optimizer=True)
    assert 0 < eng.knot_end < eng.chunk
    losses = _virtual_sharded_steps(eng, X if cfg["p"] else None, coords, t, y, 131, o["steps"])
    ref = g["opt_losses64"]
    assert np.abs(np.array(losses) - ref).max() <= 5 * TOL * max(1.0, np.abs(ref).max()), (losses, ref)
    R2._check_params(m, eng, g, cfg, 1e-4)


def test_sharded_optimizer_bf16_copies_follow_the_gathered_weights():
    """dtype='bf16' + sharded optimiser: the operand copies are re-rounded from the gathered master weights."""
    from stnf.engine import TrainStep
    cfg = cases.MODEL_CASES["c2_b257"]
    d = dev()
    m = T.build_model(cfg)
    m.train()
    X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, max_batch=cfg["B"], world_size=2, shard_optimizer=True, dtype="bf16", ema_decay=0.9)
    _virtual_sharded_steps(eng, None, coords, t, y, 100, 2)
    eng.refresh_bf16()                        # what _enqueue_sharded_optimizer does after its all-gather
    for off, h, hp, wb, wt in eng._shadow_regions:
        w = eng.flat[off:off + h * hp].view(h, hp)
        assert torch.equal(wb, w.to(torch.bfloat16)) and torch.equal(wt, w.t().contiguous().to(torch.bfloat16))


# ------------------------------------------------------------------ two real ranks on the one GPU (gloo)
def _dp3_worker(rank, world, port, sizes, B, learn, shard, q):
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
        d = torch.device("cuda:0")
        torch.cuda.set_device(d)
        coords, t, y = R2._dp_data(sum(sizes), d)
        lo = sum(sizes[:rank])
        ds = DeviceDataset(coords[lo:lo + sizes[rank]].contiguous(), t[lo:lo + sizes[rank]].contiguous(),
                           y[lo:lo + sizes[rank]].contiguous())
        m, kw = R2._dp_model(learn)
        if rank == 1:
            # a replica that was initialised differently (another seed): construction must overwrite it with rank 0's
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(0.01 * torch.randn_like(p))
            torch.manual_seed(777)
        eng = TrainStep(m, lr=1e-3, eps=R2._DP_EPS, ema_decay=0.9, max_batch=B, shard_optimizer=shard, **kw)
        assert eng.distributed and eng.world == world and eng.rank == rank and eng.shard == shard
        seeds = [None, None]
        dist.all_gather_object(seeds, eng.base_seed)
        assert seeds[0] == seeds[1]
        losses = [eng.run_epoch(ds, B, shuffle=False) for _ in range(2)]
        eng.swap_in_ema()
        ema = eng.flat.cpu().numpy().copy()
        eng.swap_in_ema()
        if rank == 0:
            q.put((losses, eng.flat.cpu().numpy(), ema, eng.step_count))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("learn,shard", [(False, True), (True, True), (False, False)])
def test_two_rank_epochs_sharded_optimizer_and_synchronised_init(learn, shard):
    """Two processes on this GPU over gloo, ragged shards, rank 1 constructed from DIFFERENT weights and another
    torch seed: the broadcast at construction makes the replicas identical, and after two epochs through the sharded
    optimiser (reduce-scatter emulated by all-reduce + slice where gloo lacks it) the parameters AND the gathered
    EMA equal a single-process run on the union batches."""
    import socket
    import torch.multiprocessing as mp
    from stnf import distributed as D
    from stnf.engine import TrainStep
    sizes, B = (2048, 2047), 1024
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp3_worker, args=(r, 2, port, sizes, B, learn, shard, q)) for r in range(2)]
    for p in procs:
        p.start()
    losses, flat, ema, steps = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    d = dev()
    coords, t, y = R2._dp_data(sum(sizes), d)
    table = D.epoch_schedule(list(sizes), B)
    assert steps == 2 * len(table)
    m, kw = R2._dp_model(learn)
    # dropout streams: the single process cannot reproduce two ranks' masks -- the model cases have dropout 0
    eng = TrainStep(m, lr=1e-3, eps=R2._DP_EPS, ema_decay=0.9, max_batch=2 * B, **kw)
    for _ in range(2):
        off = [0, sizes[0]]
        for row in table:
            idx = torch.cat([torch.arange(off[r], off[r] + row[r], device=d) for r in range(2)])
            off = [off[r] + row[r] for r in range(2)]
            eng.step(None, coords[idx], t[idx], y[idx])
        eng.mean_loss()
    n = eng.flat.numel()
    assert T.rel_l2(flat[:n], eng.flat.cpu().numpy()) <= 1e-4
    assert np.all(flat[n:] == 0.0)                       # the padding of the sharded buffer stays zero
    assert T.rel_l2(ema[:n], eng.ema.cpu().numpy()) <= 1e-4


# ------------------------------------------------------------------ non-finite guard
def _guard_setup(B=512, n=4096):
    from stnf.engine import TrainStep
    from stnf.dataio.device_dataset import DeviceDataset
    d = dev()
    m = T.build_model(cases.MODEL_CASES["default227"])
    m.train()
    coords, t, y = R2._dp_data(n, d)
    return m, DeviceDataset(coords, t, y.clone()), d


@pytest.mark.parametrize("mode", ["whole_step", "split", "learnable"])
def test_nonfinite_guard_records_the_first_bad_step(mode):
    """A NaN target in the batch of step 3 (1-based): the device word holds 3 however many steps follow, the epoch's
    loss is NaN as the reference's would be, and check_every stops the epoch after that batch
    (scripts/train_st_interp.py:724-733)."""
    from stnf.engine import TrainStep
    m, ds, d = _guard_setup()
    kw = {}
    if mode == "learnable":
        m, cfg, kn, _ = T.build_learn_model("default227_learn")
        m.train()
        kw = dict(domain_penalty_weight=0.01)
    eng = TrainStep(m, max_batch=512, world_size=2 if mode == "split" else None, **kw)
    assert eng._whole_step == (mode == "whole_step")
    ds.y[2 * 512 + 17] = float("nan")                    # shuffle=False: row of the third batch
    loss = eng.run_epoch(ds, 512, shuffle=False)
    assert np.isnan(loss) and eng.step_count == 8
    assert eng.first_nonfinite_step() == 3
    # a fresh engine with polling: stops after batch index 3 (first poll at a multiple of 2 behind the bad step)
    m2, ds2, _ = _guard_setup()
    eng2 = TrainStep(m2, max_batch=512)
    ds2.y[2 * 512 + 17] = float("inf")
    loss2 = eng2.run_epoch(ds2, 512, shuffle=False, check_every=2)
    assert not np.isfinite(loss2) and eng2.stopped_at == 3 and eng2.step_count == 4
    assert eng2.first_nonfinite_step() == 3
    # finite data: nothing recorded
    m3, ds3, _ = _guard_setup()
    eng3 = TrainStep(m3, max_batch=512)
    eng3.run_epoch(ds3, 512, shuffle=False, check_every=3)
    assert eng3.first_nonfinite_step() is None and eng3.stopped_at is None


# ------------------------------------------------------------------ Predictor / bf16 copies vs ModelEMA and load_state_dict
def test_predictor_follows_model_ema_swaps_with_the_delta_head():
    """ModelEMA.apply_shadow / restore used to write through `.data` (no version bump), so a Predictor kept
    the delta head's derived output layer / its W0^T copy of the OTHER weight set."""
    from stnf.engine import Predictor
    from stnf.utils import ModelEMA
    m, cfg, lc = T.build_quantile_model("default227_delta5")
    d = dev()
    m.eval()
    _, coords, t, _ = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    ema = ModelEMA(m, decay=0.5)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))           # training weights != shadow
    pr = Predictor(m)
    y_train = pr.predict(coords, t).clone()
    ema.apply_shadow()
    y_ema = pr.predict(coords, t).clone()
    assert torch.equal(y_ema, Predictor(m).predict(coords, t))
    assert not torch.allclose(y_ema, y_train)
    ema.restore()
    assert torch.equal(pr.predict(coords, t), y_train)
    assert torch.equal(y_train, Predictor(m).predict(coords, t))


def test_bf16_engine_copies_follow_load_state_dict_and_model_ema():
    """a TrainStep(dtype='bf16') model after load_state_dict / ModelEMA swaps must not run the hidden layers
    on the bf16 copies of the OLD weights: forward(), Predictor and the next engine step re-round them."""
    import copy
    from stnf.engine import TrainStep, Predictor
    from stnf.utils import ModelEMA
    cfg = cases.MODEL_CASES["c2_b257"]
    d = dev()
    m = T.build_model(cfg)
    m.train()
    _, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
    eng = TrainStep(m, max_batch=cfg["B"], dtype="bf16")
    eng.step(None, coords, t, y)
    sd = {k: v.clone() + 0.02 * torch.randn_like(v) if v.dtype == torch.float32 and "basis" not in k else v.clone()
          for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m.eval()
    # a fresh bf16 model holding those weights (its copies are made per call from the current weights)
    ref = T.build_model(cfg)
    ref.load_state_dict(copy.deepcopy(sd))
    ref.compute_dtype = "bf16"
    ref.eval()
    want = Predictor(ref).predict(coords, t)
    assert torch.equal(Predictor(m).predict(coords, t), want)
    with torch.no_grad():
        assert torch.equal(m(None, coords, t), ref(None, coords, t))
    # the engine's own next step starts from re-rounded copies too
    m.train()
    eng.step(None, coords, t, y)
    for off, h, hp, wb, wt in eng._shadow_regions:
        assert torch.equal(wb, eng.flat[off:off + h * hp].view(h, hp).to(torch.bfloat16))
    # ModelEMA swap on the bf16 engine's model
    m.eval()
    ema = ModelEMA(m, decay=0.5)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    pr = Predictor(m)
    y_train = pr.predict(coords, t).clone()
    ema.apply_shadow()
    ref.load_state_dict(copy.deepcopy(m.state_dict()))
    assert torch.equal(pr.predict(coords, t), Predictor(ref).predict(coords, t))
    ema.restore()
    assert torch.equal(pr.predict(coords, t), y_train)


# ------------------------------------------------------------------ pipelined preparation: announce A, step B
@pytest.mark.parametrize("torch_events", [False, True])
def test_stepping_another_batch_than_the_announced_one(torch_events, monkeypatch):
    """the step on a batch that was NOT announced takes the workspace the side stream is still binning the
    announced batch into; it must wait for that binning.  Also: non-contiguous index tensors match their announcement."""
    from stnf import engine as E
    monkeypatch.setattr(E, "_FORCE_TORCH_EVENTS", torch_events)
    d = dev()
    cfg = cases.MODEL_CASES["c2_b257"]
    coords, t, y = R2._dp_data(60000, d)
    B = 4096
    g = torch.Generator(device="cpu").manual_seed(3)
    perm = torch.randperm(60000, generator=g).to(d)
    batches = [perm[i * B:(i + 1) * B] for i in range(6)]

    def run(announce):
        m = T.build_model(cfg).train()
        eng = E.TrainStep(m, max_batch=B, seed=11)
        for i in range(5):
            eng.step_indexed(coords, t, y, batches[i], next_idx=announce(i))
        torch.cuda.synchronize()
        return eng.flat.clone()
    want = run(lambda i: None)
    # always announce a batch that is NOT stepped next
    got = run(lambda i: batches[(i + 3) % 6])
    assert torch.equal(got, want)
    # strided index views: announced and stepped through the same (non-contiguous) tensors
    wide = torch.stack([perm, perm], 1)                   # column 0 has stride 2
    views = [wide[i * B:(i + 1) * B, 0] for i in range(6)]
    m = T.build_model(cfg).train()
    eng = E.TrainStep(m, max_batch=B, seed=11)
    for i in range(5):
        eng.step_indexed(coords, t, y, views[i], next_idx=views[i + 1])
        if i > 0:
            assert eng._pipe is not None
    torch.cuda.synchronize()
    assert torch.equal(eng.flat, want)


# ------------------------------------------------------------------ rotated K-chunk order
def test_rotated_chunk_order_changes_rounding_only(monkeypatch):
    """STDADK_KROT=1 (workgroups start their walk over the K chunks of the shared weights at different chunks; a
    measured-and-left-off switch) against the default order: the same sums in another order."""
    import subprocess, sys, json
    if os.environ.get("STDADK_NO_FUSED_TAIL"):
        pytest.skip("the switch lives in the fused tail kernels")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, json, torch
sys.path[:0] = [%r, %r, %r]
from golden import cases
import test_gpu_parity as T
from stnf.engine import TrainStep
cfg = dict(cases.MODEL_CASES["c2_b257"], B=4096, seed=5)
d = T.dev()
m = T.build_model(cfg).train()
X, coords, t, y = (torch.from_numpy(a).to(d) for a in cases.make_inputs(cfg))
eng = TrainStep(m, max_batch=4096, world_size=2)
eng._enqueue_grads(None, coords, t.view(-1), y, 4096, 4096)
torch.cuda.synchronize()
torch.save(eng.grad.cpu(), sys.argv[1])
""" % (root, os.path.join(root, "st-dadk_amd"), os.path.join(root, "tests"))
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for k in ("1", "0"):
            f = os.path.join(td, f"g{k}.pt")
            env = dict(os.environ, STDADK_KROT=k)
            subprocess.run([sys.executable, "-c", code, f], check=True, env=env, timeout=600)
            outs.append(torch.load(f, weights_only=True).double())
    a, b = outs
    assert not torch.equal(a, b)                         # the order did change
    assert float((a - b).norm() / b.norm()) <= 2e-6


# ------------------------------------------------------------------ reductions folded into the merged dW launch
def _fin_run(cfg, monkeypatch, mode, dtype="f32", steps=3, dropout=0.0):
    """Gradients of one batch (data-parallel entry: forward + backward only) and the parameters after `steps` whole
    steps, with STDADK_DW_FIN=mode: 2 = the reductions inside dw_all_kernel, 1 = dw_all_kernel leaves the squared
    norms of its knot rows and reduce_jobs_kernel does the sums, 0 = reduce_jobs_kernel also re-reads the knot rows."""
    from stnf.engine import TrainStep
    monkeypatch.setenv("STDADK_DW_FIN", str(mode))
    d = dev()
    B = cfg["B"]
    X, coords, t, y = (torch.from_numpy(a).to(d) if a is not None else None for a in cases.make_inputs(cfg))
    if cfg["p"] == 0:
        X = None
    # eps in Adam's linear regime (as test_gpu_round2._DP_EPS): the two routes sum the clip norm over different
    # partials, and with eps 1e-8 Adam turns rounding-level differences of near-zero gradient entries into O(lr) steps
    kw = dict(lr=1e-3, grad_clip=0.5, max_batch=B, dtype=dtype, eps=R2._DP_EPS)
    m = T.build_model(cfg, dropout=dropout).train()
    g_eng = TrainStep(m, world_size=2, sync_init=False, **kw)
    g_eng._enqueue_grads(X, coords, t.view(-1), y, B, B)
    torch.cuda.synchronize()
    grad = g_eng.grad.clone()
    m2 = T.build_model(cfg, dropout=dropout).train()
    eng = TrainStep(m2, **kw)
    losses = []
    for _ in range(steps):
        eng.step(X, coords, t.view(-1), y)
        losses.append(eng.mean_loss())
    torch.cuda.synchronize()
    return grad, eng.flat.clone(), losses, int(eng.step_dev.item())


@pytest.mark.parametrize("name,B,dtype", [("c2_b257", 4096, "f32"), ("c2_b257", 300, "f32"), ("default227", 227, "f32"),
                                           ("c2_b257", 20000, "f32"), ("c2_b257", 4096, "bf16"),
                                           ("c2_b257_noln", 1000, "f32"), ("default227_tri", 4096, "f32")])
def test_reductions_inside_the_weight_gradient_launch(name, B, dtype, monkeypatch):
    """dw_all_kernel with the finishing work (last-arriving K slice sums its tile's slabs in slice order, tall reduce
    jobs as extra workgroups, squared-norm slots; the default from 24 576 rows) and dw_all_kernel leaving only the
    squared norms of its knot rows (the default below that) against the round-2 route (reduce_jobs_kernel behind it,
    which re-reads the knot rows): the gradients are the same sums in the same order (bit-identical while a product
    has <= 64 K slices, which is the wide reduce job's serial order); the clip norm is summed over other partials, so
    whole steps agree to rounding."""
    if any(os.environ.get(k) for k in ("STDADK_NO_DW_ALL", "STDADK_NO_FUSED_TAIL")):
        pytest.skip("the merged weight-gradient launch is switched off")
    cfg = dict(cases.MODEL_CASES[name], B=B, seed=11)
    g0, p0, l0, s0 = _fin_run(cfg, monkeypatch, 0, dtype=dtype)
    for mode in (1, 2):
        g1, p1, l1, s1 = _fin_run(cfg, monkeypatch, mode, dtype=dtype)
        assert s1 == s0 == 3
        assert torch.isfinite(g1).all() and float(g1.norm()) > 0
        if B <= 4096:
            assert torch.equal(g1, g0)
        else:
            assert float((g1 - g0).double().norm() / g0.double().norm()) <= 1e-6
        # (bf16 operands: a last-bit difference of a master weight can move its bf16 copy by 2^-9)
        assert float((p1 - p0).double().norm() / p0.double().norm()) <= (1e-6 if dtype == "f32" else 5e-5)
        np.testing.assert_allclose(l1, l0, rtol=2e-6 if dtype == "f32" else 1e-4)


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_large_batch_steps_are_reproducible_run_to_run(dropout):
    """65 536 rows take the weight-gradient launch with the reductions inside (arrival counters, last K slice sums):
    which workgroup arrives last differs from run to run, the sums may not -- two runs of three whole steps from the
    same state end in bit-identical parameters (the loss sum is accumulated with float atomics: compared to rounding)."""
    from stnf.engine import TrainStep
    if any(os.environ.get(k) for k in ("STDADK_NO_DW_ALL", "STDADK_NO_FUSED_TAIL")):
        pytest.skip("the merged weight-gradient launch is switched off")
    cfg = dict(cases.MODEL_CASES["c2_b257"], B=65536, seed=3)
    d = dev()
    X, coords, t, y = (torch.from_numpy(a).to(d) if a is not None else None for a in cases.make_inputs(cfg))
    res = []
    for _ in range(2):
        m = T.build_model(cfg, dropout=dropout).train()
        eng = TrainStep(m, lr=1e-3, grad_clip=0.5, max_batch=cfg["B"], seed=7)
        for _ in range(3):
            eng.step(None, coords, t.view(-1), y)
        torch.cuda.synchronize()
        res.append((eng.flat.clone(), eng.mean_loss()))
        del eng, m
    assert torch.isfinite(res[0][0]).all()
    assert torch.equal(res[0][0], res[1][0])
    assert abs(res[0][1] - res[1][1]) <= 1e-6 * abs(res[0][1])


@pytest.mark.parametrize("name,B,dtype", [("c2_b257", 4096, "f32"), ("c2_b257", 1500, "f32"), ("default227", 700, "f32"),
                                           ("c2_b257", 4096, "bf16"), ("c2_b257", 6000, "f32")])
def test_next_batch_binned_inside_the_optimiser_launch(name, B, dtype):
    """One-call steps on small batches bin the NEXT batch with extra workgroups of their optimiser launch
    (stdadk_train_step_next_f32, adamw_bin_kernel) instead of on a side stream: the binning is the same integer
    work (bit-identical bins) and the optimiser's arithmetic does not depend on the launch shape, so a run of steps
    ends in bit-identical parameters either way; the loss sums (float atomics) agree to rounding."""
    from stnf.engine import TrainStep
    cfg = dict(cases.MODEL_CASES[name], B=4 * B + 37, seed=21)
    d = dev()
    X, coords, t, y = (torch.from_numpy(a).to(d) if a is not None else None for a in cases.make_inputs(cfg))
    g = torch.Generator(device="cpu").manual_seed(5)
    perm = torch.randperm(coords.shape[0], generator=g).to(d)
    batches = [perm[i * B:(i + 1) * B] for i in range(4)] + [perm[:B // 2]]        # the last one ragged
    res = []
    for inline in (True, False):
        m = T.build_model(cfg, dropout=0.1).train()
        eng = TrainStep(m, lr=1e-3, grad_clip=0.5, max_batch=B, dtype=dtype, seed=11, inline_prep=inline)
        assert eng._whole_step
        for i, idx in enumerate(batches):
            eng.step_indexed(coords, t.view(-1), y, idx, next_idx=batches[i + 1] if i + 1 < len(batches) else None)
            if i == 1:
                # (6 000 rows take a 128 x 128 cell grid: the library declines, the engine prepares on the side stream)
                # (STNF_NO_INLINE_PREP=1 in the environment switches it off for every engine)
                assert eng._prepared is not None and eng._prepared[3] == (eng.inline_prep and B <= 4096)
        torch.cuda.synchronize()
        res.append((eng.flat.clone(), eng.mean_loss(), int(eng.step_dev.item())))
    assert res[0][2] == res[1][2] == len(batches)
    assert torch.isfinite(res[0][0]).all()
    assert torch.equal(res[0][0], res[1][0])
    assert abs(res[0][1] - res[1][1]) <= 1e-6 * abs(res[0][1])
