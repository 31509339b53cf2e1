"""Observation-sharded data parallelism: host-side logic (one process per GPU, RCCL over xGMI).

The reference has no distributed code; this is the one collective the build adds.
Every rank holds the full model (<= 51 MB) and its own shard of the observations.  Per step:
  1. each rank runs forward/backward on its local rows with grad_scale = 1/(GLOBAL rows * Q), so
     its gradient is its share of the global-batch mean (nn.MSELoss over the union of the shards);
  2. ONE all-reduce (SUM) of the flat fp32 gradient buffer;
  3. the clip norm is taken from the reduced gradient, AdamW/EMA run replicated.
A ragged last batch needs no special case: grad_scale carries the global row count
(SURVEY.md §7 "DDP equivalence").  These helpers are backend-agnostic (`nccl` = RCCL on the GPUs,
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


def grad_scale(global_row_count, out_dim):
    """d(mean over all B*Q elements of the GLOBAL batch)/d(sum of squared errors)."""
    return 1.0 / (float(global_row_count) * float(out_dim))


def allreduce_gradients(flat_grad, group=None):
    """In-place SUM all-reduce of the flat gradient; returns the async work handle's result."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


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
