"""world_size-2 gloo tests (CPU) of the data-parallel host logic: shard ranges, the count-weighted
gradient reduction for ragged shards, and the post-reduce clip norm."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "st-dadk_amd"))
    from stnf import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)                      # same data on every rank, each takes its shard
        X = torch.randn(n_rows, 7, dtype=torch.float64)
        y = torch.randn(n_rows, 1, dtype=torch.float64)
        w = torch.randn(7, 1, dtype=torch.float64)
        lo, hi = D.shard_range(n_rows, rank, world)
        G = D.global_rows(hi - lo)
        assert G == n_rows
        scale = D.grad_scale(G, 1)
        # local gradient of scale * sum((Xw - y)^2) — what stdadk_train_fwd_bwd_f32 produces per rank
        r = X[lo:hi] @ w - y[lo:hi]
        g_local = (2.0 * scale) * (X[lo:hi].T @ r)
        flat = g_local.reshape(-1).clone()
        D.allreduce_gradients(flat)
        # reference: gradient of the global-batch MEAN squared error
        g_ref = (2.0 / n_rows) * (X.T @ (X @ w - y))
        assert torch.allclose(flat.view_as(g_ref), g_ref, rtol=1e-12, atol=1e-14)
        coef = D.clip_coefficient(flat, 0.05)
        ref = min(1.0, 0.05 / (float(g_ref.norm()) + 1e-6))
        assert abs(coef - ref) < 1e-12
        q.put((rank, lo, hi, float(flat.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [64, 101])          # 101 => ragged shards (51 / 50)
def test_count_weighted_allreduce_gloo(n_rows):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_rows
    assert abs(res[0][3] - res[1][3]) < 1e-12          # every rank holds the same reduced gradient


def test_shard_range_covers_everything():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "st-dadk_amd"))
    from stnf.distributed import shard_range
    for n in (0, 1, 7, 8, 100_000, 1_000_003):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _grid_worker(rank, world, port, S, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "st-dadk_amd"))
    from stnf import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1)
        coords = torch.rand(S, 2)
        tv = torch.linspace(0, 1, 5)
        # stand-in for Predictor.predict_grid: any row-independent function of (site, time), two outputs
        fn = lambda c, t: torch.stack([torch.sin(3 * c[:, 0])[None, :] * (1 + t[:, None]),
                                       (c[:, 1] ** 2)[None, :] - t[:, None]], -1)
        out = D.sharded_predict_grid(fn, coords, tv)
        assert out.shape == (5, S, 2) and torch.equal(out, fn(coords, tv))
        q.put((rank, float(out.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("S", [64, 101, 1])            # 101, 1 => ragged / empty shards
def test_sharded_prediction_grid_gloo(S):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grid_worker, args=(r, world, port, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert abs(res[0][1] - res[1][1]) < 1e-9


# ------------------------------------------------------------------ epoch schedule (ragged shards)
def test_epoch_schedule_properties():
    """Every rank gets the same number of steps, a non-empty batch in each, all its rows exactly once, never
    more than the batch size; equal shards keep the DataLoader shape (full batches + a ragged last one)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "st-dadk_amd"))
    from stnf.distributed import epoch_schedule, shard_range
    for n, world, B in [(4095, 2, 1024), (2049, 2, 1024), (4097, 2, 1024), (100_000, 8, 4096), (1_000_003, 8, 16384),
                        (7, 2, 1024), (8193, 4, 1024), (4096 * 8, 8, 4096), (4096 * 8 + 1, 8, 4096), (6000, 2, 1)]:
        sizes = [hi - lo for lo, hi in (shard_range(n, r, world) for r in range(world))]
        table = epoch_schedule(sizes, B)
        assert len(table) == -(-max(sizes) // B)
        for r in range(world):
            col = [row[r] for row in table]
            assert sum(col) == sizes[r] and min(col) >= 1 and max(col) <= B, (n, world, B, col)
        if len(set(sizes)) == 1:
            full, rem = divmod(sizes[0], B)
            assert [row[0] for row in table] == [B] * full + ([rem] if rem else [])
    # the cases of ADVICE r1: (2048, 2047) and (1025, 1024) rows at B = 1024
    assert epoch_schedule([2048, 2047], 1024) == [[1024, 1024], [1024, 1023]]
    assert epoch_schedule([1025, 1024], 1024) == [[1024, 1023], [1, 1]]
    # very unequal caller-made shards: the short one spreads its rows over the long one's steps
    t = epoch_schedule([5000, 1000], 1024)
    assert len(t) == 5 and sum(r[1] for r in t) == 1000 and min(r[1] for r in t) == 200
    with pytest.raises(RuntimeError):
        epoch_schedule([5000, 3], 1024)
    assert epoch_schedule([0, 0], 16) == [] and epoch_schedule([], 16) == []


def _sched_worker(rank, world, port, sizes, B, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "st-dadk_amd"))
    from stnf import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = D.gather_shard_sizes(sizes[rank])
        assert got == list(sizes)
        table = D.epoch_schedule(got, B)
        # the loop run_epoch runs: one gradient all-reduce per step and NO other collective; a rank-dependent
        # number of steps or an extra collective on one rank would hang here (the parent's join times out)
        w = torch.zeros(3, dtype=torch.float64)
        seen = 0
        for row in table:
            mine, rows = row[rank], sum(row)
            g = torch.full((3,), float(mine) / rows, dtype=torch.float64)      # this rank's share of a mean
            D.allreduce_gradients(g)
            assert torch.allclose(g, torch.ones(3, dtype=torch.float64), atol=1e-12)   # shares add up to the mean
            w += g
            seen += mine
        assert seen == sizes[rank]
        q.put((rank, len(table), float(w.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes,B", [((2048, 2047), 1024), ((1025, 1024), 1024), ((2049, 2048), 1024), ((300, 37), 64)])
def test_epoch_schedule_keeps_collectives_matched_gloo(sizes, B):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sched_worker, args=(r, world, port, sizes, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res[0][1] == res[1][1] and abs(res[0][2] - res[1][2]) < 1e-12


# ------------------------------------------------------------------ sharded optimiser: the collective pattern
def _shard_worker(rank, world, port, P, ke, q):
    """The host-side pattern of TrainStep(shard_optimizer=True) in float64 torch on the CPU: reduce-scatter of the
    flat gradient, per-slice / per-group sums of squares, ONE all-reduce of the partial vector, clipped AdamW + EMA on
    the slice, all-gather of the parameters -- against the replicated pattern (all-reduce, everything everywhere)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "st-dadk_amd"))
    from stnf import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chunk = -(-P // (32 * world)) * 32                 # the engine's padding: slices start on 128-byte lines
        n = chunk * world
        lo, hi = rank * chunk, (rank + 1) * chunk
        g0 = torch.Generator().manual_seed(5)
        p_all = torch.zeros(n, dtype=torch.float64)
        p_all[:P] = torch.randn(P, generator=g0, dtype=torch.float64)
        gr = torch.Generator().manual_seed(100 + rank)      # every rank's own share of the gradient
        g_loc = torch.zeros(n, dtype=torch.float64)
        g_loc[:P] = torch.randn(P, generator=gr, dtype=torch.float64)
        lr, lr_k, clip, clip_k, b1, b2, eps, wd, dec = 2e-2, 1e-3, 0.7, 0.07, 0.9, 0.999, 1e-8, 5e-4, 0.9

        def adam(p, g, m, v, e, coef, lr_):
            g = g * coef
            p = p * (1 - lr_ * wd)
            m = b1 * m + (1 - b1) * g
            v = b2 * v + (1 - b2) * g * g
            p = p - lr_ / (1 - b1) * m / (v.sqrt() / (1 - b2) ** 0.5 + eps)
            return p, m, v, dec * e + (1 - dec) * p

        # replicated reference
        g_full = g_loc.clone()
        D.allreduce_gradients(g_full)
        ck = min(1.0, clip_k / (float(g_full[:ke].norm()) + 1e-6)) if ke else 1.0
        cm = min(1.0, clip / (float(g_full[ke:].norm()) + 1e-6))
        ref = p_all.clone()
        ref_e = p_all.clone()
        z = torch.zeros(n, dtype=torch.float64)
        ref[:ke], _, _, ref_e[:ke] = adam(p_all[:ke], g_full[:ke], z[:ke], z[:ke], p_all[:ke], ck, lr_k)
        ref[ke:], _, _, ref_e[ke:] = adam(p_all[ke:], g_full[ke:], z[ke:], z[ke:], p_all[ke:], cm, lr)
        # sharded
        g_sh = g_loc.clone()
        mine = g_sh[lo:hi]
        D.reduce_scatter_gradients(g_sh, mine)
        assert torch.allclose(mine, g_full[lo:hi], rtol=1e-13, atol=1e-15)
        a1, b1_ = max(0, lo), min(ke, hi)                   # this rank's part of the knot group
        a0, b0 = max(ke, lo), hi                            # ... of the MLP group
        parts = torch.tensor([float((g_sh[a0:b0] ** 2).sum()) if b0 > a0 else 0.0,
                              float((g_sh[a1:b1_] ** 2).sum()) if b1_ > a1 else 0.0], dtype=torch.float64)
        D.allreduce_gradients(parts)
        cm2 = min(1.0, clip / (float(parts[0].sqrt()) + 1e-6))
        ck2 = min(1.0, clip_k / (float(parts[1].sqrt()) + 1e-6)) if ke else 1.0
        assert abs(cm2 - cm) < 1e-12 and abs(ck2 - ck) < 1e-12
        flat = p_all.clone()
        ema_sh = p_all[lo:hi].clone()
        zs = torch.zeros(chunk, dtype=torch.float64)
        if b1_ > a1:
            flat[a1:b1_], _, _, ema_sh[a1 - lo:b1_ - lo] = adam(p_all[a1:b1_], g_sh[a1:b1_], zs[a1 - lo:b1_ - lo],
                                                               zs[a1 - lo:b1_ - lo], p_all[a1:b1_], ck2, lr_k)
        if b0 > a0:
            flat[a0:b0], _, _, ema_sh[a0 - lo:b0 - lo] = adam(p_all[a0:b0], g_sh[a0:b0], zs[a0 - lo:b0 - lo],
                                                              zs[a0 - lo:b0 - lo], p_all[a0:b0], cm2, lr)
        D.allgather_parameters(flat, flat[lo:hi].clone())
        assert torch.allclose(flat, ref, rtol=1e-12, atol=1e-14)
        ema_all = torch.empty(n, dtype=torch.float64)
        D.allgather_parameters(ema_all, ema_sh)
        assert torch.allclose(ema_all, ref_e, rtol=1e-12, atol=1e-14)
        assert float(flat[P:].abs().sum()) == 0.0           # the padding stays zero
        q.put((rank, float(flat.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("P,ke", [(1000, 0), (1000, 90), (777, 600), (4096, 2048)])
def test_sharded_optimizer_pattern_equals_replicated_gloo(P, ke):
    """ke = size of the knot group at the head of the flat buffer: inside rank 0's slice (90), across the slice
    boundary (600 of 777 -> slices of 416), exactly on it (2048)."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, P, ke, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert abs(res[0][1] - res[1][1]) < 1e-9              # both ranks hold the same gathered parameters
