"""Observation-sharded data parallelism: host-side logic (one process per GPU, RCCL over xGMI).

The reference has no distributed code; this is the one collective the build adds.
Every rank holds the full model (<= 51 MB) and its own shard of the observations.  Per step:
  1. each rank runs forward/backward on its local rows with grad_scale = 1/(GLOBAL rows * Q), so
     its gradient is its share of the global-batch mean (nn.MSELoss over the union of the shards);
  2. ONE all-reduce (SUM) of the flat fp32 gradient buffer;
  3. the clip norm is taken from the reduced gradient, AdamW/EMA run replicated.
A ragged batch needs no special case in the arithmetic: grad_scale carries the global row count
(SURVEY.md §7 "DDP equivalence").  What ragged SHARDS need is a common step count: `epoch_schedule` fixes it
(and every step's global row count) on the host from one all-gather of the shard sizes, so no rank ever
waits in a collective its partners do not enter.  These helpers are backend-agnostic (`nccl` = RCCL on the GPUs,
`gloo` in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous [lo, hi) slice of n rows owned by `rank`: sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_rows(local_rows, group=None, device=None):
    """Sum of the ranks' local row counts (needed when shards are ragged)."""
    if not (dist.is_available() and dist.is_initialized()):
        return int(local_rows)
    t = torch.tensor([int(local_rows)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def gather_shard_sizes(local_rows, group=None, device=None):
    """Row count of every rank's shard, identical on all ranks: ONE all-gather per data set (not per step)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [int(local_rows)]
    world = dist.get_world_size(group)
    mine = torch.tensor([int(local_rows)], dtype=torch.int64, device=device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return [int(p.item()) for p in parts]


def _rank_batches(n, batch_size, steps):
    """Batch sizes of one rank's epoch in exactly `steps` non-empty batches.  Full batches with a ragged last one
    (a DataLoader without drop_last, scripts/train_st_interp.py:2295-2301) whenever that already takes `steps`
    batches; a rank that would finish one step early hands one row of its last full batch to an extra batch; a
    rank with far fewer rows than the longest shard splits its rows evenly."""
    if n < steps:
        raise RuntimeError(f"a shard of {n} rows cannot take part in {steps} steps; give every rank at least as "
                           f"many rows as the longest shard has batches")
    full, rem = divmod(n, batch_size)
    own = full + (1 if rem else 0)
    if own == steps:
        return [batch_size] * full + ([rem] if rem else [])
    if own == steps - 1 and rem == 0:
        return [batch_size] * (full - 1) + [batch_size - 1, 1]
    base, extra = divmod(n, steps)
    return [base + 1] * extra + [base] * (steps - extra)


def epoch_schedule(shard_sizes, batch_size):
    """Collective-safe epoch plan for observation-sharded training, computed on the host from the gathered
    shard sizes (every rank computes the same table): returns sizes[step][rank].  Every rank runs the same
    number of steps (ceil(longest shard / batch)) with a non-empty batch in each, so the per-step gradient
    all-reduce always meets its partners, and a step's global row count is sum(sizes[step]) with no
    per-step collective."""
    if batch_size < 1:
        raise ValueError("batch_size must be >= 1")
    if not shard_sizes or max(shard_sizes) == 0:
        return []
    steps = -(-max(shard_sizes) // batch_size)
    per_rank = [_rank_batches(n, batch_size, steps) for n in shard_sizes]
    return [[per_rank[r][i] for r in range(len(shard_sizes))] for i in range(steps)]


def grad_scale(global_row_count, out_dim):
    """d(mean over all B*Q elements of the GLOBAL batch)/d(sum of squared errors)."""
    return 1.0 / (float(global_row_count) * float(out_dim))


def allreduce_gradients(flat_grad, group=None):
    """In-place SUM all-reduce of the flat gradient; returns the async work handle's result."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


_RS_NATIVE = {}        # backend name -> "in_place" | "scratch" (separate output) | "all_reduce" (no reduce-scatter: gloo)


def reduce_scatter_gradients(flat_grad, out_slice, group=None):
    """SUM reduce-scatter of the flat gradient (length world * chunk): `out_slice` = this rank's chunk
    flat_grad[rank * chunk:(rank + 1) * chunk] (a view: in place, as RCCL allows) receives the sum over ranks of
    that chunk.  On the xGMI full mesh each GPU exchanges 1/world of the buffer with every peer directly -- all 7
    links at once -- where a ring all-reduce passes 2 (world-1)/world of it over one link.
    Backends without the collective (gloo in the CPU / one-GPU rehearsals) fall back to all-reduce: the slice then
    holds the same sums."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return out_slice
    backend = dist.get_backend(group)
    mode = _RS_NATIVE.get(backend, "in_place")
    if mode == "in_place":
        # RCCL / NCCL reduce-scatter in place: recvbuff = sendbuff + rank * recvcount (what Megatron's distributed
        # optimiser does with its gradient buffer views)
        try:
            dist.reduce_scatter_tensor(out_slice, flat_grad, op=dist.ReduceOp.SUM, group=group)
            _RS_NATIVE[backend] = "in_place"
            return out_slice
        except (RuntimeError, NotImplementedError, ValueError):
            # refused before anything was enqueued (argument check of this torch build, or a backend without the
            # collective): every rank runs the same build, so all of them take the next branch together
            mode = _RS_NATIVE[backend] = "scratch" if backend == "nccl" else "all_reduce"
    if mode == "scratch":
        tmp = torch.empty_like(out_slice)
        dist.reduce_scatter_tensor(tmp, flat_grad, op=dist.ReduceOp.SUM, group=group)
        out_slice.copy_(tmp)
        return out_slice
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return out_slice


def allgather_parameters(flat, my_slice, group=None):
    """All-gather of the ranks' stepped parameter chunks into the flat buffer (`my_slice` may be this rank's own
    chunk of `flat`: in place, sendbuff = recvbuff + rank * sendcount)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return flat
    try:
        dist.all_gather_into_tensor(flat, my_slice, group=group)
    except (RuntimeError, ValueError):
        if my_slice.untyped_storage().data_ptr() != flat.untyped_storage().data_ptr():
            raise
        dist.all_gather_into_tensor(flat, my_slice.clone(), group=group)      # a build that refuses aliased buffers
    return flat


def clip_coefficient(flat_grad, max_norm):
    """min(1, max_norm/(||g||+1e-6)) of the REDUCED gradient (torch.nn.utils.clip_grad_norm_)."""
    if not max_norm or max_norm <= 0:
        return 1.0
    total = float(torch.linalg.vector_norm(flat_grad.double()))
    return min(1.0, max_norm / (total + 1e-6))


def sharded_predict_grid(predict_grid, coords, t_values, group=None):
    """Dense prediction grid over several GPUs (BASELINE config C5): the rows are independent, so every rank
    evaluates its contiguous shard of the SITES at all times with `predict_grid(coords_shard, t_values)`
    -> (T, S_r, Q) (Predictor.predict_grid) and the shards are all-gathered into (T, S, Q) on every rank.
    Ragged shards (S not a multiple of the world size) are padded to the longest for the collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return predict_grid(coords, t_values)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    S = coords.shape[0]
    lo, hi = shard_range(S, rank, world)
    local = predict_grid(coords[lo:hi], t_values)                   # (T, hi - lo, Q)
    T, Q = local.shape[0], local.shape[2]
    longest = shard_range(S, 0, world)[1]                           # rank 0 owns one of the longest shards
    pad = torch.zeros(T, longest, Q, dtype=local.dtype, device=local.device)
    pad[:, :hi - lo] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.empty(T, S, Q, dtype=local.dtype, device=local.device)
    for r in range(world):
        a, b = shard_range(S, r, world)
        out[:, a:b] = parts[r][:, :b - a]
    return out
