"""Fused train / inference steps over libstdadk — the fast path behind the drop-in model class.

The reference's batch body (scripts/train_st_interp.py:608-721) is: H2D copies, zero_grad, forward,
MSELoss, backward, clip_grad_norm_, AdamW.step, EMA.update, two loss.item() syncs.  `TrainStep`
runs the same arithmetic as one chain of HIP kernels on one stream with

  * parameters, gradients, Adam moments and the EMA shadow in FLAT fp32 buffers (the nn.Module's
    parameters become views of the flat buffer, so state_dict()/checkpoints are unchanged);
  * the loss accumulated on the device (no per-step .item());
  * optional hipGraph capture of the whole step (static input buffers), replayed per batch;
  * observation-sharded data parallelism: one RCCL all-reduce of the flat gradient per step,
    count-weighted so a ragged last batch still equals the global-batch mean
    (SURVEY.md §7 "DDP equivalence"), with the clip norm taken after the reduction.
"""
import math
import weakref

import os

import torch
import torch.distributed as dist

from . import _native as N
from . import distributed as D


# Batches up to this size are binned by ONE workgroup (window.hip: SMALL_B); random row reads from a
# single CU are slow, so there a separate many-workgroup gather launch first is faster (MI355X, C2,
# B = 4096: 0.185 ms/step gathered first vs 0.197 ms reading in place); above it the multi-kernel binning
# spreads the reads over the chip and reading in place saves the gather launch.
_INDEXED_MIN_B = 8192


def flatten_parameters(model, transpose_first=True, pad_multiple=4):
    """Move every trainable parameter of `model` into one flat fp32 buffer (the parameters become
    views, so state_dict()/checkpoints keep their names and shapes).  Each parameter is padded to a
    multiple of 4 floats so every view is 16-byte aligned.  With `transpose_first` the first
    Linear's weight is STORED (in,out) row-major — the layout the window kernels gather rows from —
    and `model.mlp[0].weight` becomes its .t() view (same values, shape (out,in)).
    The buffer's length is rounded up to `pad_multiple` floats (zeros: equal shards of a sharded optimiser).
    Returns (flat, [(name, offset, numel)])."""
    params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    dev = params[0][1].device
    offs, total = [], 0
    for n, p in params:
        offs.append((n, total, p.numel()))
        total += (p.numel() + 3) // 4 * 4
    total = (total + pad_multiple - 1) // pad_multiple * pad_multiple
    flat = torch.zeros(total, device=dev, dtype=torch.float32)
    first_w = model._body[0].weight
    for (n, p), (_, o, k) in zip(params, offs):
        if transpose_first and p is first_w:
            out_f, in_f = p.shape
            flat[o:o + k].view(in_f, out_f).copy_(p.data.t())
            p.data = flat[o:o + k].view(in_f, out_f).t()
        else:
            flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = flat[o:o + k].view(p.shape)
    return flat, offs


def _rank_seed(base, rank):
    """Dropout seed of one rank: the base seed with the rank mixed in (an odd 64-bit multiplier, so different
    ranks never share a (seed, step, layer, element) stream)."""
    return (int(base) + 0xD1B54A32D192ED03 * int(rank)) & 0xFFFFFFFFFFFFFFFF


def _wait(stream, ev):
    if isinstance(ev, _LightEvent):
        import ctypes
        if ev.recorded:
            rc = _LightEvent._hip.hipStreamWaitEvent(ctypes.c_void_p(stream.cuda_stream), ev.h, 0)
            if rc:
                raise RuntimeError(f"hipStreamWaitEvent failed: {rc}")
    else:
        stream.wait_event(ev)


class _LightEvent:
    """HIP event without timing and without the system-scope fence (hipEventDisableSystemFence): the
    streams it orders are on one device."""
    _hip = None

    def __init__(self):
        import ctypes
        if _LightEvent._hip is None:
            _LightEvent._hip = ctypes.CDLL("libamdhip64.so")
        self.h = ctypes.c_void_p()
        rc = _LightEvent._hip.hipEventCreateWithFlags(ctypes.byref(self.h), 0x2 | 0x20000000)
        if rc:
            raise RuntimeError(f"hipEventCreateWithFlags failed: {rc}")
        self.recorded = False

    def record(self, stream):
        import ctypes
        rc = _LightEvent._hip.hipEventRecord(self.h, ctypes.c_void_p(stream.cuda_stream))
        if rc:
            raise RuntimeError(f"hipEventRecord failed: {rc}")
        self.recorded = True


_FORCE_TORCH_EVENTS = False     # tests: take the torch.cuda.Event branch of _event_factory


def _event_factory():
    """_LightEvent when the HIP runtime can be reached through ctypes, else torch.cuda.Event (same ordering
    semantics, a system-scope fence per record)."""
    if _FORCE_TORCH_EVENTS:
        return torch.cuda.Event
    try:
        _LightEvent()
        return _LightEvent
    except (OSError, RuntimeError, AttributeError):
        return torch.cuda.Event


class TrainStep:
    """One fused optimisation step of STInterpMLP (fixed or learnable knots).

    Objective (scripts/train_st_interp.py:617-658): loss="mse" (regression_type 'mean'), or
    loss="pinball" with `quantile_levels` — one level = 'quantile', several = 'multi-quantile' (mean
    over levels of the per-level check loss on (B,1) targets) plus either the prediction-level
    non-crossing penalty (non_crossing_weight, non_crossing_power) or, with the delta head, the
    parameter-level P_nc(delta) (non_crossing_lambda).

    Learnable knots (`spatial_learnable=True` models; scripts/train_st_interp.py:470-499,660-672,698-705):
    the knot tensors form their own AdamW group with lr x `basis_lr_ratio` (`set_basis_lr` for the
    progressive unfreezing of :582-602) and clip norm `grad_clip` x `basis_clip_ratio`;
    `domain_penalty_weight` / `movement_penalty_weight` add the penalties of st_interp.py:493-546 to the
    objective, and the model's gradient-damping settings apply to the centres' gradient.

    Sparsity penalties on the first layer (`sparsity_penalty_type` 'element' | 'group' | 'sparse_group' with the
    config keys of scripts/train_st_interp.py:674-691): value into the loss accumulator, gradient into dW0.

    Launch mode: an eager chain of kernels by default (`use_graph=True` replays it from a hipGraph);
    `step_indexed(..., next_idx=...)` / `run_epoch` overlap the next batch's preparation with the step.

    Data parallel (one process per GPU, observations sharded, `torch.distributed` initialised): the replicas are
    made identical at construction (`sync_init`: rank 0's parameters, buffers and dropout base seed are broadcast --
    data-dependent knot initialisers see different shards on every rank), then per step either
      * `shard_optimizer=False`: ONE all-reduce of the flat gradient, clip norm + AdamW + EMA replicated; or
      * `shard_optimizer=True`: reduce-scatter of the flat gradient (every rank receives the SUM of its 1/world
        slice -- on the xGMI full mesh all 7 links of a GPU carry 1/8 of the buffer at once instead of a ring passing
        7/8 of it over one link), sum of squares of the local slice, one all-reduce of the 2 x 256 clip-norm
        partials, AdamW + EMA on the slice only (Adam moments and the EMA shadow exist only for it: 1/world of the
        optimiser's HBM traffic and memory), all-gather of the stepped parameters.  Same arithmetic as the
        replicated mode (scripts/train_st_interp.py:696-712: global-norm clip, step, EMA) up to the order of the
        clip norm's partial sums.

    Non-finite guard (`nonfinite_guard`): the optimiser launch of every step checks the objective accumulator on the
    device; `first_nonfinite_step()` / `run_epoch(check_every=k)` report or stop at the first batch whose objective
    was NaN/inf, as scripts/train_st_interp.py:724-733 does, without a host sync per step."""

    def __init__(self, model, lr=2e-2, weight_decay=5e-4, betas=(0.9, 0.999), eps=1e-8,
                 grad_clip=10.0, ema_decay=None, max_batch=4096, use_graph=False,
                 process_group=None, distributed=None, force_dense=False, two_streams=False,
                 loss="mse", quantile_levels=None, non_crossing_weight=0.0, non_crossing_power=1,
                 non_crossing_lambda=0.0, basis_lr_ratio=0.05, basis_clip_ratio=0.1,
                 domain_penalty_weight=0.0, movement_penalty_weight=0.0, sparsity_penalty_type="none",
                 sparsity_lambda_l1=0.001, sparsity_lambda_group=0.01, sparsity_apply_to_spatial=True,
                 sparsity_apply_to_temporal=True, seed=None, world_size=None, dtype="f32", shard_optimizer=False,
                 sync_init=True, nonfinite_guard=True, inline_prep=True):
        self.model = model
        if dtype not in ("f32", "bf16"):
            raise ValueError(f"unknown dtype '{dtype}'; use 'f32' or 'bf16'")
        self.dtype = dtype
        if loss not in ("mse", "pinball"):
            raise ValueError(f"unknown loss '{loss}'; use 'mse' or 'pinball'")
        self.loss_kind = loss
        self.quantile_levels = [float(q) for q in quantile_levels] if quantile_levels is not None else None
        if loss == "pinball" and (self.quantile_levels is None
                                  or len(self.quantile_levels) != model.output_dim):
            raise ValueError(f"pinball loss needs output_dim={model.output_dim} quantile levels")
        self.nc_weight = 0.0 if model._has_delta else float(non_crossing_weight)
        self.nc_power = int(non_crossing_power)
        self.nc_lambda = float(non_crossing_lambda) if model._has_delta else 0.0
        self._loss_descs = {}
        self.indexed_min_batch = _INDEXED_MIN_B
        self.dev = next(model.parameters()).device
        if self.dev.type != "cuda" and not N.dry_run():
            raise RuntimeError("TrainStep needs the model on a HIP device; there is no CPU path")
        # distributed
        if distributed is None:
            distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
        self.distributed = bool(distributed)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self.rank = dist.get_rank(process_group) if self.distributed else 0
        if world_size is not None:
            # tests: this engine plays one of `world_size` ranks whose gradients the caller sums itself
            # (split path, 1/world shares of the parameter-level penalties); see set_virtual_rank
            self.world = int(world_size)
        self.shard = bool(shard_optimizer) and self.world > 1
        if self.distributed and sync_init:
            # identical replicas: parameters AND buffers (knot tables of the gmm / random_site initialisers depend on
            # the rank's own train_coords) from rank 0, before anything is derived from them
            with torch.no_grad():
                for tns in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(tns.data, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                   group=process_group)
        # every rank's slice of the flat buffers starts on a 128-byte line: chunk = a multiple of 32 floats
        self.flat, self.offsets = flatten_parameters(model, transpose_first=True,
                                                     pad_multiple=32 * self.world if self.shard else 4)
        self.grad = torch.zeros_like(self.flat)
        self._shard_cache = None        # cached slices / descriptors of the sharded optimiser (_shard_views)
        self._adam_groups = None        # cached descriptors of the two-group optimiser launch (_enqueue_optimizer)
        self._knot_train = None         # ... and of the knot penalties
        self.chunk = self.flat.numel() // self.world if self.shard else self.flat.numel()
        self.lo = self.rank * self.chunk if self.shard else 0
        self.hi = self.lo + self.chunk
        # sharded optimiser: Adam moments and the EMA shadow exist for this rank's slice [lo, hi) only
        self.m = torch.zeros(self.chunk, device=self.dev)
        self.v = torch.zeros(self.chunk, device=self.dev)
        self.ema = self.flat[self.lo:self.hi].clone() if ema_decay is not None else None
        self._ema_backup = None
        self._vstate = {}          # tests (set_virtual_rank with the sharded optimiser): rank -> (m, v, ema) slices
        self.ema_decay = 0.0 if ema_decay is None else float(ema_decay)
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), betas, float(eps)
        self.grad_clip = float(grad_clip or 0.0)
        self.step_count = 0
        self.max_batch = int(max_batch)
        # BASELINE config C3 (dtype="bf16"): the Linear layers after the first take bf16 operands on the matrix
        # cores.  Master weights, gradients, AdamW state and the EMA stay fp32 in the flat buffers; the bf16
        # operand copies (each weight and its transpose) live in one buffer that the optimiser kernel rewrites
        # from the values it has just stepped.
        self._shadow_buf = None
        self._shadow_regions = []
        self._shadow_tables = {}
        model.compute_dtype = dtype
        model._bf16_engine = None
        if dtype == "bf16":
            self._install_bf16_copies()
        self.state = model._step_state(self.dev, force_dense=force_dense, training=True)
        assert self.state.w0_transposed
        self.uses_window = N.step_uses_window(self.state.basis, self.state.desc, self.state.flags)
        # gradient views in _param_list() order; dW0 is stored transposed like W0
        views, dviews = [], []
        by_name = {n: (o, k) for n, o, k in self.offsets}
        first_w = model._body[0].weight
        for n, p in model.named_parameters():
            if p.requires_grad:
                o, k = by_name[n]
                if p is first_w:
                    views.append(self.grad[o:o + k].view(p.shape[1], p.shape[0]))
                elif n.startswith("delta_params."):
                    dviews.append(self.grad[o:o + k])
                elif n.startswith("spatial_basis."):
                    pass            # knot gradients come from stdadk_knot_backward_f32 (below)
                else:
                    views.append(self.grad[o:o + k].view(p.shape))
        self.grad_views = views
        # sparsity penalties on the first layer (config keys of train_st_interp.py:674-691): their gradient
        # is added to dW0^T by stdadk_sparsity_f32 between backward and clipping
        self._sparsity = None
        if sparsity_penalty_type != "none":
            self._sparsity = N.make_sparsity(sparsity_penalty_type, sparsity_lambda_l1, sparsity_lambda_group,
                                             sparsity_apply_to_spatial, sparsity_apply_to_temporal)
            o, k = by_name[next(n for n, p in model.named_parameters() if p is first_w)]
            self._w0t = self.flat[o:o + k].view(first_w.shape[1], first_w.shape[0])
            self._g_w0t = self.grad[o:o + k].view(first_w.shape[1], first_w.shape[0])
        # learnable knots (DA-STDK): their own AdamW group (lr x basis_lr_ratio) and clip norm
        # (grad_clip x basis_clip_ratio), scripts/train_st_interp.py:470-483,698-705; the knot tensors
        # are registered first, so they occupy the head [0, knot_end) of the flat buffers
        self.learnable = bool(model.spatial_basis.learnable)
        self.knot_end = 0
        self.knot_train = None
        if self.learnable:
            sb = model.spatial_basis
            (nc, oc, kc), (nl, ol, kl) = self.offsets[0], self.offsets[1]
            assert nc == "spatial_basis.centers" and nl == "spatial_basis.log_bandwidths" and oc == 0
            self.g_centers = self.grad[oc:oc + kc].view(sb.k, 2)
            self.g_log_bw = self.grad[ol:ol + kl]
            self.knot_end = self.offsets[2][1]
            self.basis_lr = self.lr * float(basis_lr_ratio)
            self.basis_clip = self.grad_clip * float(basis_clip_ratio)
            self.basis_lr_dev = torch.full((1,), self.basis_lr, device=self.dev)
            self.sumsq_basis = torch.zeros(N.SUMSQ_PARTS, device=self.dev)
            self.domain_w, self.movement_w = float(domain_penalty_weight), float(movement_penalty_weight)
        self.d_head = self.d_delta = None
        if model._has_delta:
            # the library differentiates w.r.t. the derived output layer; dWo/dbo are scratch and
            # stdadk_delta_head_backward_f32 folds them into the delta rows of the flat gradient
            self.d_head = [torch.empty_like(self.state.head[0]), torch.empty_like(self.state.head[1])]
            self.d_delta = model._delta_matrix(dviews)
            assert self.d_delta.data_ptr() == dviews[0].data_ptr() and self.state.delta.stride(0) == self.d_delta.stride(0)
        self.grads_t = model._pack(views + (self.d_head or []))
        B = self.max_batch
        self.ws = torch.empty(N.step_workspace_bytes(self.state.basis, self.state.desc, B, self.state.flags) // 4,
                              device=self.dev)
        self.y_pred = torch.empty(B, model.output_dim, device=self.dev)
        self.loss_sum = torch.zeros(1, device=self.dev)       # running sum of squared errors
        self.sumsq = torch.zeros(N.SUMSQ_PARTS, device=self.dev)
        self.lr_dev = torch.full((1,), self.lr, device=self.dev)
        self.step_dev = torch.zeros(1, device=self.dev, dtype=torch.int32)
        # dropout stream: the base seed comes from torch's generator (so `set_seed` / torch.manual_seed decide
        # it, as they decide nn.Dropout's masks in the reference) unless given; the rank is mixed in below so
        # that the shards of a data-parallel batch draw independent masks
        self.base_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)
        # optional: independent kernels of a step fork onto this stream (fork/join inside the library).
        # Measured on MI355X at B = 4096: no gain eager, 12 % SLOWER under hipGraph replay (cross-stream
        # edges cost more than the overlap of ~20 us kernels buys), hence off by default.
        self.aux_stream = torch.cuda.Stream(device=self.dev) if two_streams else None
        self.rows_seen = 0
        if self.distributed and sync_init:
            bs = torch.tensor([self.base_seed], dtype=torch.int64, device=self.dev)
            dist.broadcast(bs, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                           group=process_group)
            self.base_seed = int(bs.item())
        self.seed = _rank_seed(self.base_seed, self.rank)
        # graph
        self.use_graph = bool(use_graph)
        self._graph = None
        self._g_in = None
        self._g_B = None
        self._warm = False
        self._ix = None
        self._ix_graph = None
        # the one-call step applies with a single parameter group on one GPU (data-parallel training needs
        # the all-reduce between backward and optimiser; learnable knots / the delta head have extra kernels there)
        self._whole_step = (not self.distributed and self.world == 1 and not self.learnable
                            and not model._has_delta and self.aux_stream is None)
        self._optim = None
        self._sumsq512 = torch.zeros(N.GRADSQ_PARTS, device=self.dev)
        self._pipe = None          # two workspaces + side stream of the pipelined batch preparation
        self._prepared = None      # ((idx data_ptr, numel, stride), workspace index, idx tensor, inline) of the announced batch
        # small batches of the one-call step: the next batch is binned inside this step's optimiser launch
        # (False or STNF_NO_INLINE_PREP=1: on the side stream, as for every other step kind)
        self.inline_prep = bool(inline_prep) and os.environ.get("STNF_NO_INLINE_PREP", "") != "1"
        self.time_allreduce = False   # bench: bracket the step's collectives with timing events
        self.allreduce_events = []
        # clip-norm partials of both parameter groups in ONE buffer (the sharded optimiser all-reduces it once)
        self._sumsq_all = torch.zeros(2 * N.SUMSQ_PARTS, device=self.dev)
        if self.shard:
            self.sumsq = self._sumsq_all[:N.SUMSQ_PARTS]
            if self.learnable:
                self.sumsq_basis = self._sumsq_all[N.SUMSQ_PARTS:]
        self._rs_native = None     # does the backend have reduce_scatter_tensor (gloo: emulated by all-reduce + slice)
        # non-finite guard: first (1-based) step whose objective left the accumulator NaN/inf, 0 = none so far
        self.nonfinite = torch.zeros(1, device=self.dev, dtype=torch.int32) if nonfinite_guard else None
        self.stopped_at = None     # run_epoch(check_every=...): index of the batch it stopped after, or None

    # ------------------------------------------------------------------------------------
    def _install_bf16_copies(self):
        m = self.model
        lins = m._linears()
        by_name = {n: (o, k) for n, o, k in self.offsets}
        names = {id(p): n for n, p in m.named_parameters()}
        total = sum(2 * lins[l].weight.numel() for l in range(1, len(m.hidden_dims)))
        self._shadow_buf = torch.empty(max(total, 8), device=self.dev, dtype=torch.bfloat16)
        pairs, regions, pos = [None] * len(lins), [], 0
        for l in range(1, len(m.hidden_dims)):
            w = lins[l].weight
            h, hp = w.shape
            wb = self._shadow_buf[pos:pos + h * hp].view(h, hp)
            wt = self._shadow_buf[pos + h * hp:pos + 2 * h * hp].view(hp, h)
            pos += 2 * h * hp
            pairs[l] = (wb, wt)
            regions.append((by_name[names[id(w)]][0], h, hp, wb, wt))
        self._shadow_regions = regions
        m._bf16_engine = pairs
        m._bf16_refresh = weakref.WeakMethod(self.refresh_bf16)
        self.refresh_bf16()

    def _shadow(self, base=0):
        """stdadk_bf16_shadow table with offsets relative to flat[base:] (None in fp32 mode or without layers
        after the first)."""
        if not self._shadow_regions:
            return None
        key = int(base)
        if key not in self._shadow_tables:
            self._shadow_tables[key] = N.make_bf16_shadow([(o - key, h, hp, wb, wt)
                                                           for o, h, hp, wb, wt in self._shadow_regions])
        return self._shadow_tables[key]

    def refresh_bf16(self):
        """Re-round the bf16 operand copies from the fp32 master weights: needed after the parameters were
        changed by anything but this engine's optimiser (load_state_dict, in-place edits); swap_in_ema calls it."""
        sh = self._shadow(0)
        if sh is not None:
            N.bf16_shadow_refresh(self.flat, sh)
            self.model._bf16_key = self.model._bf16_master_key()

    def _check_bf16_current(self):
        """The operand copies follow this engine's own optimiser; anything else that rewrote the master weights
        (load_state_dict, ModelEMA.apply_shadow / restore, an in-place edit under no_grad) moves the key."""
        if self._shadow_regions and self.model._bf16_key != self.model._bf16_master_key():
            self.refresh_bf16()

    def set_lr(self, lr):
        self.lr = float(lr)
        self.lr_dev.fill_(self.lr)

    def set_basis_lr(self, lr):
        """Learning rate of the knot group: 0 while frozen, ramped after `basis_unfreeze_epoch`
        (scripts/train_st_interp.py:582-602)."""
        if not self.learnable:
            raise RuntimeError("the model's knots are not learnable")
        self.basis_lr = float(lr)
        self.basis_lr_dev.fill_(self.basis_lr)

    def _enqueue(self, X, coords, t, y, B, global_rows, idx=None, ws=None, prebinned=False, nxt=None):
        """All kernels of one step on the current stream (capturable: no sync, no allocation).
        With `idx` (window path) X/coords/t/y are the RESIDENT arrays and the batch is their rows idx;
        `prebinned`: the batch already sits binned in workspace `ws` (pipelined preparation).
        `nxt` = (next_idx, next_workspace), one-call step only: the optimiser launch also bins the next batch
        (stdadk_train_step_next_f32); returns True when it did."""
        st = self.state
        ws = self.ws if ws is None else ws
        flags = st.flags | (N.FLAG_PREBINNED if prebinned else 0)
        Q = self.model.output_dim
        if self._whole_step:
            # single GPU, one parameter group: the whole step is ONE library call, which also takes the
            # gradient's squared norm out of the launches that produce it (no separate pass over it)
            if self._optim is None:
                self._optim = N.make_optim(self.flat, self.grad, self.m, self.v, self.ema, self.lr, self.lr_dev,
                                           self.betas, self.eps, self.wd, self.step_dev, self.grad_clip,
                                           self._sumsq512 if self.grad_clip > 0 else None, self.ema_decay,
                                           shadow=self._shadow(0), nonfinite_step=self.nonfinite)
            if nxt is not None:
                return N.train_step_next(st.basis, st.desc, st.params, self.grads_t, coords, t, X, y, idx,
                                         D.grad_scale(global_rows, Q), self.loss_sum, ws, flags, self._optim, nxt[0],
                                         nxt[1], seed=self.seed, loss_desc=self._loss_desc(y.shape[1]),
                                         sparsity_desc=self._sparsity)
            N.train_step(st.basis, st.desc, st.params, self.grads_t, coords, t, X, y,
                         idx if not prebinned else None, B, D.grad_scale(global_rows, Q), self.loss_sum, ws, flags,
                         self._optim, seed=self.seed, loss_desc=self._loss_desc(y.shape[1]),
                         sparsity_desc=self._sparsity)
            return False
        self._enqueue_grads(X, coords, t, y, B, global_rows, idx=idx, ws=ws, prebinned=prebinned)
        if self.shard:
            if not self.distributed:
                # world_size=... without a process group is the tests' virtual-rank mode: there the caller plays the
                # collectives between _enqueue_grads / _shard_sumsq / _shard_adamw itself
                raise RuntimeError("shard_optimizer=True needs an initialised torch.distributed process group")
            self._enqueue_sharded_optimizer()
            return
        if self.distributed:
            self._timed(lambda: D.allreduce_gradients(self.grad, self.pg))
        self._enqueue_optimizer()

    def _timed(self, fn):
        """Run a collective; under `time_allreduce` bracket it with timing events on the step's stream."""
        if not self.time_allreduce:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.allreduce_events.append((e0, e1))
        return out

    # ---- sharded optimiser (shard_optimizer=True) ---------------------------------------------------------
    def _local(self, a, b):
        """This rank's part of the flat range [a, b): (start, stop) in flat coordinates, stop <= start if empty."""
        return max(a, self.lo), min(b, self.hi)

    def _shard_views(self):
        """Slices and optimiser descriptors of this rank's part of the flat buffers -- rebuilt only when a buffer, a
        boundary or a rate changes (a dozen slices and up to 14 address checks per step otherwise, on a path whose
        host side also has three collectives to enqueue per step)."""
        ke, lo = self.knot_end, self.lo
        ema = self.ema
        key = (self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
               ema.data_ptr() if ema is not None else 0, lo, self.hi, ke, self.lr, self.grad_clip, id(self.lr_dev)) + \
              ((self.basis_lr, self.basis_clip, id(self.basis_lr_dev)) if ke else ())
        c = self._shard_cache
        if c is not None and c["key"] == key:
            return c
        clip = self.grad_clip > 0
        a0, b0 = self._local(ke, self.flat.numel())
        a1, b1 = self._local(0, ke) if ke else (0, 0)

        def bufs(a, b):
            return (self.flat[a:b], self.grad[a:b], self.m[a - lo:b - lo], self.v[a - lo:b - lo],
                    ema[a - lo:b - lo] if ema is not None else None)
        c = {"key": key, "g_mlp": self.grad[a0:max(b0, a0)], "g_knot": self.grad[a1:max(b1, a1)] if ke else None,
             "groups": [], "single": None}
        if b0 > a0:
            c["groups"].append(N.make_adam_group(*bufs(a0, b0), self.lr, self.lr_dev, self.grad_clip if clip else 0.0,
                                                 self.sumsq if clip else None))
        if ke and b1 > a1:
            c["groups"].append(N.make_adam_group(*bufs(a1, b1), self.basis_lr, self.basis_lr_dev,
                                                 self.basis_clip if clip else 0.0, self.sumsq_basis if clip else None))
        if len(c["groups"]) == 1:
            mlp = b0 > a0
            c["single"] = (bufs(a0, b0) if mlp else bufs(a1, b1), c["groups"][0].lr, c["groups"][0].max_norm,
                           self.sumsq if mlp else self.sumsq_basis, self.lr_dev if mlp else self.basis_lr_dev)
        self._shard_cache = c
        return c

    def _shard_sumsq(self):
        """Sum of squares of the REDUCED gradient on this rank's slice, per parameter group, into the 2 x 256
        partials of `_sumsq_all` (a group this rank holds nothing of contributes zeros); advances the device step
        counter once.  The SUM of the ranks' partial vectors gives the global clip norms."""
        if self.grad_clip <= 0:
            N.step_advance(self.step_dev)
            return
        c = self._shard_views()
        if self.knot_end:
            N.sumsq2(c["g_mlp"], self.sumsq, c["g_knot"], self.sumsq_basis, step_inc=self.step_dev)
        else:
            N.sumsq(c["g_mlp"], self.sumsq, step_inc=self.step_dev)

    def _shard_adamw(self):
        """AdamW + EMA on this rank's slice (moments / EMA shadow indexed from `lo`), each group with its own learning
        rate and its GLOBAL clip norm (the summed partials in `_sumsq_all`)."""
        c = self._shard_views()
        watch = self.loss_sum if self.nonfinite is not None else None
        groups = c["groups"]
        if len(groups) == 2:
            N.adamw_ema2(groups[0], groups[1], self.betas, self.eps, self.wd, self.step_count + 1,
                         ema_decay=self.ema_decay, step_dev=self.step_dev, loss_watch=watch,
                         nonfinite_step=self.nonfinite)
        elif groups:
            (p_, g_, m_, v_, e_), lr, max_norm, parts, lr_dev = c["single"]
            N.adamw_ema(p_, g_, m_, v_, e_, lr, self.betas, self.eps, self.wd, self.step_count + 1, max_norm=max_norm,
                        sumsq_parts=parts, ema_decay=self.ema_decay, lr_dev=lr_dev, step_dev=self.step_dev,
                        loss_watch=watch, nonfinite_step=self.nonfinite)

    def _enqueue_sharded_optimizer(self):
        """reduce-scatter(gradient) -> local sum of squares -> all-reduce(2 x 256 partials) -> AdamW/EMA on the slice
        -> all-gather(parameters) [-> bf16 operand copies re-rounded from the gathered master weights]."""
        mine = self.grad[self.lo:self.hi]
        self._timed(lambda: D.reduce_scatter_gradients(self.grad, mine, self.pg))
        self._shard_sumsq()
        if self.grad_clip > 0:
            n = 2 * N.SUMSQ_PARTS if self.knot_end else N.SUMSQ_PARTS
            self._timed(lambda: D.allreduce_gradients(self._sumsq_all[:n], self.pg))
        self._shard_adamw()
        self._timed(lambda: D.allgather_parameters(self.flat, self.flat[self.lo:self.hi], self.pg))
        if self._shadow_regions:
            self.refresh_bf16()

    def _enqueue_grads(self, X, coords, t, y, B, global_rows, idx=None, ws=None, prebinned=False):
        """Split path, first half: this rank's share of the global-batch gradient into `self.grad` (every
        tensor overwritten) and of the objective into `self.loss_sum`.  d(mean over the GLOBAL batch)/dparams:
        each rank scales by 1/global_rows and adds 1/world of the parameter-level penalty gradients, so the
        SUM over ranks is the gradient of the single-process objective on the union batch."""
        st = self.state
        ws = self.ws if ws is None else ws
        flags = st.flags | (N.FLAG_PREBINNED if prebinned else 0)
        Q = self.model.output_dim
        if B == 0:
            # a rank without rows in this step (possible only with caller-made batches; run_epoch's schedule
            # never produces one): no data term; the penalty shares below need a batch's backward state
            if self.learnable or self._sparsity is not None or (st.head is not None and self.nc_lambda != 0.0):
                raise RuntimeError("an empty local batch cannot carry this rank's share of the parameter-level "
                                   "penalties; use run_epoch's schedule (stnf.distributed.epoch_schedule)")
            self.grad.zero_()
            return
        if st.head is not None:
            N.delta_head(st.delta, st.head[0], st.head[1])          # output layer of this step's delta
        if idx is not None and not prebinned:
            N.train_fwd_bwd_indexed(st.basis, st.desc, st.params, self.grads_t, coords, t, X, y, idx,
                                    D.grad_scale(global_rows, Q), self.loss_sum, None, ws, flags,
                                    seed=self.seed, step_dev=self.step_dev, aux_stream=self.aux_stream,
                                    loss_desc=self._loss_desc(y.shape[1]))
        else:
            N.train_fwd_bwd(st.basis, st.desc, st.params, self.grads_t, coords, t, X, y, B,
                            D.grad_scale(global_rows, Q), self.loss_sum, None, ws, flags,
                            seed=self.seed, step_dev=self.step_dev, aux_stream=self.aux_stream,
                            loss_desc=self._loss_desc(y.shape[1]))
        if st.head is not None:
            # every rank adds 1/world of the parameter-level penalty gradient (the all-reduce SUMs);
            # the loss accumulator is in units of rows*Q like the data term
            N.delta_head_backward(st.delta, self.d_head[0], self.d_head[1], self.nc_lambda / self.world,
                                  self.nc_lambda * B * Q, self.d_delta,
                                  self.loss_sum if self.nc_lambda != 0.0 else None)
        if self.learnable:
            sb = self.model.spatial_basis
            kkey = (sb.centers_init.data_ptr(), sb.gradient_damping, sb.damping_threshold, sb.damping_strength,
                    self.domain_w, self.movement_w, self.world, B * Q)
            if self._knot_train is None or self._knot_train[0] != kkey:
                self._knot_train = (kkey, N.make_knot_train(sb.centers_init, sb.gradient_damping, sb.damping_threshold,
                                                            sb.damping_strength, self.domain_w, self.movement_w,
                                                            penalty_grad_scale=1.0 / self.world,
                                                            penalty_loss_scale=float(B * Q)))
            kt = self._knot_train[1]
            N.knot_backward(st.basis, st.desc, st.params, coords, B, ws, st.flags, kt,
                            self.g_centers, self.g_log_bw, self.loss_sum)
        if self._sparsity is not None:
            m = self.model
            N.sparsity(self._sparsity, self._w0t, self._g_w0t, True, m.p, m.k_spatial, m.k_temporal,
                       grad_scale=1.0 / self.world, loss_scale=float(B * Q), loss_sum=self.loss_sum)

    def _enqueue_optimizer(self):
        """Split path, second half: clip norm(s) of the (reduced) gradient, AdamW + EMA; advances the device
        step counter once."""
        ke = self.knot_end
        watch = self.loss_sum if self.nonfinite is not None else None
        if ke and self.grad_clip > 0:
            # learnable knots: both groups' clip norms in one launch, both AdamW/EMA updates in one launch.
            # The two group descriptors (14 buffer addresses, 10 slices) only change when a buffer or a rate does:
            # rebuilt per step they were a quarter of this path's host time, and this path is host-bound.
            ema = self.ema
            key = (self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                   ema.data_ptr() if ema is not None else 0, self.lr, self.basis_lr, self.grad_clip, self.basis_clip,
                   ke, id(self.lr_dev), id(self.basis_lr_dev), len(self._shadow_regions))
            if self._adam_groups is None or self._adam_groups[0] != key:
                g_mlp = N.make_adam_group(self.flat[ke:], self.grad[ke:], self.m[ke:], self.v[ke:],
                                          ema[ke:] if ema is not None else None, self.lr, self.lr_dev, self.grad_clip,
                                          self.sumsq, shadow=self._shadow(ke))
                g_knot = N.make_adam_group(self.flat[:ke], self.grad[:ke], self.m[:ke], self.v[:ke],
                                           ema[:ke] if ema is not None else None, self.basis_lr, self.basis_lr_dev,
                                           self.basis_clip, self.sumsq_basis)
                self._adam_groups = (key, g_mlp, g_knot, self.grad[ke:], self.grad[:ke])
            _, g_mlp, g_knot, gv_mlp, gv_knot = self._adam_groups
            N.sumsq2(gv_mlp, self.sumsq, gv_knot, self.sumsq_basis, step_inc=self.step_dev)
            N.adamw_ema2(g_mlp, g_knot, self.betas, self.eps, self.wd, self.step_count + 1,
                         ema_decay=self.ema_decay, step_dev=self.step_dev, loss_watch=watch,
                         nonfinite_step=self.nonfinite)
            return
        if self.grad_clip > 0:
            N.sumsq(self.grad[ke:], self.sumsq, step_inc=self.step_dev)
            if ke:
                N.sumsq(self.grad[:ke], self.sumsq_basis)
        else:
            N.step_advance(self.step_dev)
        ema = self.ema
        N.adamw_ema(self.flat[ke:], self.grad[ke:], self.m[ke:], self.v[ke:], ema[ke:] if ema is not None else None,
                    self.lr, self.betas, self.eps, self.wd, self.step_count + 1, max_norm=self.grad_clip,
                    sumsq_parts=self.sumsq, ema_decay=self.ema_decay, lr_dev=self.lr_dev, step_dev=self.step_dev,
                    shadow=self._shadow(ke), loss_watch=watch, nonfinite_step=self.nonfinite)
        if ke:
            N.adamw_ema(self.flat[:ke], self.grad[:ke], self.m[:ke], self.v[:ke], ema[:ke] if ema is not None else None,
                        self.basis_lr, self.betas, self.eps, self.wd, self.step_count + 1, max_norm=self.basis_clip,
                        sumsq_parts=self.sumsq_basis, ema_decay=self.ema_decay, lr_dev=self.basis_lr_dev,
                        step_dev=self.step_dev)

    def set_virtual_rank(self, rank):
        """Tests of the data-parallel arithmetic on one GPU: play rank `rank` of `world_size` (dropout stream
        of that rank; the caller sums the ranks' `grad` buffers between _enqueue_grads and _enqueue_optimizer)."""
        if self.shard:
            # the sharded optimiser state of every virtual rank is kept: playing rank r swaps its (m, v, ema) in
            self._vstate[self.rank] = (self.m, self.v, self.ema)
            lo = int(rank) * self.chunk
            if int(rank) not in self._vstate:
                self._vstate[int(rank)] = (torch.zeros_like(self.m), torch.zeros_like(self.v),
                                           self.flat[lo:lo + self.chunk].clone() if self.ema is not None else None)
            self.m, self.v, self.ema = self._vstate[int(rank)]
            self.lo, self.hi = lo, lo + self.chunk
        self.rank = int(rank)
        self.seed = _rank_seed(self.base_seed, self.rank)

    def _loss_desc(self, y_cols):
        """ABI loss descriptor for targets with `y_cols` columns (None = the plain MSE fast path)."""
        Q = self.model.output_dim
        if y_cols not in (1, Q):
            raise RuntimeError(f"targets have {y_cols} columns; expected 1 or output_dim={Q}")
        if self.loss_kind == "mse" and y_cols == Q:
            return None
        if y_cols not in self._loss_descs:
            self._loss_descs[y_cols] = N.make_loss(self.loss_kind, Q, y_cols, self.quantile_levels,
                                                   self.nc_weight, self.nc_power)
        return self._loss_descs[y_cols]

    def step(self, X, coords, t, y, global_rows=None):
        """One optimisation step on device tensors coords (B,2), t (B,1)/(B,), y (B,Q), X (B,p)|None.
        `global_rows` = rows of the global batch over all ranks (defaults to B * world)."""
        B = coords.shape[0]
        if B > self.max_batch:
            raise RuntimeError(f"batch {B} > max_batch {self.max_batch}")
        self._check_bf16_current()
        if global_rows is None:
            global_rows = B * self.world
        coords = coords.contiguous().float()
        t = t.contiguous().float().view(-1)
        y = y.contiguous().float()
        if self.model.p > 0:
            X = X.contiguous().float()
        else:
            X = None
        if self.use_graph and not self.distributed:
            self._step_graph(X, coords, t, y, B, global_rows)
        else:
            self._enqueue(X, coords, t, y, B, global_rows)
        self._stepped(B)

    def _stepped(self, B):
        self.step_count += 1
        self.rows_seen += B
        # the kernels update the parameters through raw pointers (no autograd version bump): Predictors that
        # cache derived tensors (the delta head's output layer) watch this counter
        self.model._engine_version = getattr(self.model, "_engine_version", 0) + 1
        if self._shadow_regions:
            self.model._bf16_key = self.model._bf16_master_key()      # the optimiser kernel has just re-rounded them

    def step_indexed(self, coords_all, t_all, y_all, idx, X_all=None, global_rows=None, next_idx=None):
        """One optimisation step on rows `idx` (int64 device tensor) of device-RESIDENT observation
        arrays coords_all (N,2), t_all (N,) or (N,1), y_all (N,Q), X_all (N,p)|None.  The batch is
        gathered by one library kernel into static buffers (the reference builds it from a Python
        list of dicts, torch.stack and four H2D copies per step: train_st_interp.py:413-460,609-612);
        with use_graph the gather is part of the captured graph.

        `next_idx` (window path, eager launch chain): the rows of the FOLLOWING step.  Their gather and
        binning are enqueued on a second stream into a second workspace while this step runs, and the
        following call (same tensor as its `idx`) starts at the layer-0 kernel — batch preparation
        leaves the critical path.  `next_idx` (and the resident arrays) only need to be ENQUEUED on the
        current stream, not complete: the side stream is ordered behind the current stream's work at the
        time of this call.  The index tensors must not be modified in between."""
        B = idx.numel()
        if B > self.max_batch:
            raise RuntimeError(f"batch {B} > max_batch {self.max_batch}")
        self._check_bf16_current()
        if global_rows is None:
            global_rows = B * self.world
        p, yc = self.model.p, y_all.shape[1]
        if self._ix is None or self._ix[0].numel() != B or self._ix[3].shape[1] != yc:
            self._ix = (torch.empty(B, dtype=torch.int64, device=self.dev),
                        torch.empty(B, 2, device=self.dev), torch.empty(B, device=self.dev),
                        torch.empty(B, yc, device=self.dev),
                        torch.empty(B, p, device=self.dev) if p > 0 else None)
            self._ix_graph = None
        ib, cb, tb, yb, xb = self._ix
        t_all = t_all.view(-1)
        Xa = X_all if p > 0 else None

        graphed = self.use_graph and not self.distributed
        # a captured graph reads the indices from a static buffer; the eager chain takes them as they are
        src = ib if graphed else (idx if idx.is_contiguous() else idx.contiguous())
        if self.uses_window and not graphed and (next_idx is not None or self._prepared is not None):
            # announcements are keyed on the CALLER's tensors (a .contiguous() copy has a new address every call)
            self._step_pipelined(coords_all, t_all, y_all, Xa, src, next_idx, B, global_rows,
                                 key=(idx.data_ptr(), idx.numel(), idx.stride(0) if idx.dim() else 1))
            self._stepped(B)
            return

        def enqueue():
            if self.uses_window and B > self.indexed_min_batch:
                # the binning kernels read rows idx of the resident arrays in place
                self._enqueue(Xa, coords_all, t_all, y_all, B, global_rows, idx=src)
            else:
                N.gather_batch(coords_all, t_all, y_all, Xa, src, cb, tb, yb, xb)
                self._enqueue(xb, cb, tb, yb, B, global_rows)

        if graphed:
            ib.copy_(idx)
        if graphed:
            if not self._warm:
                self._warm = True
                enqueue()
            else:
                key = (B, global_rows, coords_all.data_ptr(), t_all.data_ptr(), y_all.data_ptr())
                if self._ix_graph is None or self._ix_graph[0] != key:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        enqueue()
                    self._ix_graph = (key, g)
                self._ix_graph[1].replay()
        else:
            enqueue()
        self._stepped(B)

    def run_epoch(self, dataset, batch_size, generator=None, shuffle=True, check_every=0):
        """One pass over a `stnf.dataio.device_dataset.DeviceDataset` in shuffled mini-batches (the
        epoch loop of scripts/train_st_interp.py:608-724 with the set resident in HBM): every step
        announces the next batch so that its preparation overlaps the running step.  Returns the mean
        batch objective of the epoch (ONE host sync).

        Non-finite objectives: the reference leaves the epoch at the first batch whose loss is NaN
        (scripts/train_st_interp.py:724-733, after that batch's optimiser step).  The device-side guard records that
        step without a host sync; `check_every=k` reads it every k steps (one 4-byte read = one sync each) and stops
        the epoch there (`stopped_at` = index of the last batch stepped), `check_every=0` runs the epoch through and
        leaves the step in `first_nonfinite_step()`.

        Data-parallel: `dataset` is this rank's shard.  The shard sizes are all-gathered once per call (8 bytes per
        rank: every rank enters it, whatever it has cached) and `stnf.distributed.epoch_schedule` gives every rank the
        same number of steps, a non-empty batch in each and each step's global row count -- no per-step collective
        besides the gradient exchange.  A `check_every` poll is a MAX all-reduce of the guard word, so that every rank
        leaves the epoch after the same step."""
        if self.distributed:
            sizes = D.gather_shard_sizes(len(dataset), self.pg, device=self.dev)
            table = D.epoch_schedule(sizes, int(batch_size))
            batches = dataset.epoch_batches([row[self.rank] for row in table], generator=generator, shuffle=shuffle)
            rows = [sum(row) for row in table]
        else:
            batches = dataset.epoch_batches(batch_size, generator=generator, shuffle=shuffle)
            rows = [None] * len(batches)
        self.stopped_at = None
        for i, idx in enumerate(batches):
            nxt = batches[i + 1] if i + 1 < len(batches) else None
            self.step_indexed(dataset.coords, dataset.t, dataset.y, idx, X_all=dataset.X, global_rows=rows[i],
                              next_idx=nxt)
            if check_every and self.nonfinite is not None and (i + 1) % check_every == 0 and nxt is not None:
                bad = self.nonfinite
                if self.distributed:
                    bad = bad.clone()
                    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.pg)
                if int(bad.item()) > 0:
                    self.stopped_at = i
                    break
        return self.mean_loss()

    def _step_pipelined(self, coords_all, t_all, y_all, Xa, idx, next_idx, B, global_rows, key=None):
        """Step on a batch that was (or is now) binned into one of two workspaces, and batch
        preparation of `next_idx` on the side stream into the other one."""
        main = torch.cuda.current_stream(self.dev)
        if self._pipe is None:
            # events without the system-scope fence (both streams are on this device): 3 us per step
            # cheaper than torch.cuda.Event on MI355X; fall back if the HIP runtime cannot be reached
            mk = _event_factory()
            self._pipe = dict(ws=[self.ws, torch.empty_like(self.ws)], stream=torch.cuda.Stream(device=self.dev),
                              announce=mk(), binned=mk(), last=1)
        pp = self._pipe
        st = self.state
        prep = self._prepared
        self._prepared = None
        if key is None:
            key = (idx.data_ptr(), idx.numel(), idx.stride(0) if idx.dim() else 1)
        if prep is not None and prep[0] == key:
            wsi, prebinned = prep[1], True
            if not prep[3]:
                _wait(main, pp["binned"])       # (prepared inside the previous step's optimiser launch: same stream)
        else:
            wsi, prebinned = 1 - pp["last"], False        # not announced: bin inside the step, in place
            if prep is not None and not prep[3]:
                # another batch was announced: the side stream may still be binning it into exactly this workspace
                _wait(main, pp["binned"])
        # one-call step on a small batch: the NEXT batch is binned by extra workgroups of this step's optimiser launch
        # (no side stream, no cross-stream packets in the main queue: DESIGN.md section 8, "the bubble between steps")
        inline = (next_idx is not None and self._whole_step and self.inline_prep and not self.distributed
                  and next_idx.numel() <= 8192 and next_idx.dtype == torch.int64)
        if inline:
            nxt = next_idx if next_idx.is_contiguous() else next_idx.contiguous()
            wsj = 1 - wsi
            me = idx if idx.is_contiguous() else idx.contiguous()
            done = self._enqueue(Xa, coords_all, t_all, y_all, B, global_rows, idx=me, ws=pp["ws"][wsi],
                                 prebinned=prebinned, nxt=(nxt, pp["ws"][wsj]))
            if done:
                self._prepared = ((next_idx.data_ptr(), next_idx.numel(), next_idx.stride(0) if next_idx.dim() else 1),
                                  wsj, nxt, True)
                pp["last"] = wsi
                return
            # (the library did not take it -- e.g. more than 64 x 64 cells: the step itself is done, prepare as usual)
            pp["announce"].record(main)
            _wait(pp["stream"], pp["announce"])
            with torch.cuda.stream(pp["stream"]):
                N.bin_batch(st.basis, st.desc, coords_all, t_all, Xa, y_all, nxt, pp["ws"][wsj], st.flags)
                pp["binned"].record(pp["stream"])
            self._prepared = ((next_idx.data_ptr(), next_idx.numel(), next_idx.stride(0) if next_idx.dim() else 1),
                              wsj, nxt, False)
            pp["last"] = wsi
            return
        if next_idx is not None:
            nxt = next_idx if next_idx.is_contiguous() else next_idx.contiguous()
            # the side stream starts after everything enqueued on the main stream so far: the step that last
            # used that workspace (the previous one), AND whatever produced `next_idx` and the resident arrays
            # (a device randperm at the start of an epoch is a multi-kernel sort on the main stream)
            pp["announce"].record(main)
        # this step's launches go to the main stream BEFORE the side stream's: when the GPU is idle (first step
        # after a host synchronise) it starts on the step at once instead of after the host has enqueued the
        # five launches of the next batch's preparation
        if prebinned:
            self._enqueue(None, None, None, y_all, B, global_rows, ws=pp["ws"][wsi], prebinned=True)
        else:
            self._enqueue(Xa, coords_all, t_all, y_all, B, global_rows, idx=idx, ws=pp["ws"][wsi])
        if next_idx is not None:
            wsj = 1 - wsi
            _wait(pp["stream"], pp["announce"])
            with torch.cuda.stream(pp["stream"]):
                N.bin_batch(st.basis, st.desc, coords_all, t_all, Xa, y_all, nxt, pp["ws"][wsj], st.flags)
                pp["binned"].record(pp["stream"])
            self._prepared = ((next_idx.data_ptr(), next_idx.numel(), next_idx.stride(0) if next_idx.dim() else 1),
                              wsj, nxt, False)
        pp["last"] = wsi

    def _step_graph(self, X, coords, t, y, B, global_rows):
        if not self._warm:
            # the first step runs eagerly: HIP loads code objects and applies the kernels' LDS
            # attributes on first launch, neither of which may happen inside a stream capture
            self._warm = True
            self._enqueue(X, coords, t, y, B, global_rows)
            return
        if self._graph is None or self._g_B != (B, global_rows, y.shape[1]):
            p = self.model.p
            self._g_in = (torch.empty(B, p, device=self.dev) if p > 0 else None,
                          torch.empty(B, 2, device=self.dev), torch.empty(B, device=self.dev),
                          torch.empty(B, y.shape[1], device=self.dev))
            self._g_B = (B, global_rows, y.shape[1])
            # AdamW and the dropout generator read lr / step from device scalars, so a replay
            # advances them; capture itself executes nothing
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._enqueue(self._g_in[0], self._g_in[1], self._g_in[2], self._g_in[3], B, global_rows)
            self._graph = g
        gi = self._g_in
        if gi[0] is not None:
            gi[0].copy_(X)
        gi[1].copy_(coords); gi[2].copy_(t); gi[3].copy_(y)
        self._graph.replay()

    def mean_loss(self, reset=True):
        """Mean batch objective (MSE, or check loss + penalties) over the rows seen since the last
        reset (ONE host sync)."""
        val = self.loss_sum.item() / max(self.rows_seen * self.model.output_dim, 1)
        if reset:
            self.loss_sum.zero_()
            self.rows_seen = 0
        return val

    def swap_in_ema(self):
        """Exchange the live parameters with the EMA shadow (validation under EMA weights,
        scripts/train_st_interp.py:739,790); call again to swap back.  With the sharded optimiser the shadow exists in
        slices: the first call all-gathers it into the parameter buffer (the live values are kept aside), the
        second puts the live values back -- a collective, every rank must call it."""
        if self.ema is None:
            raise RuntimeError("EMA is disabled")
        if self.shard:
            if self._ema_backup is None:
                self._ema_backup = self.flat.clone()
                if self.distributed:
                    D.allgather_parameters(self.flat, self.ema, self.pg)
                else:                               # virtual ranks (tests): the slices this engine has played
                    self._vstate[self.rank] = (self.m, self.v, self.ema)
                    for r, (_, _, e) in self._vstate.items():
                        self.flat[r * self.chunk:(r + 1) * self.chunk].copy_(e)
            else:
                self.flat.copy_(self._ema_backup)
                self._ema_backup = None
        else:
            tmp = self.flat.clone()
            self.flat.copy_(self.ema)
            self.ema.copy_(tmp)
        self.model._engine_version = getattr(self.model, "_engine_version", 0) + 1
        self.refresh_bf16()

    def first_nonfinite_step(self):
        """1-based number of the first optimisation step whose batch objective was NaN/inf, or None (ONE host sync;
        scripts/train_st_interp.py:724-733 prints and leaves the epoch there)."""
        if self.nonfinite is None:
            raise RuntimeError("the engine was built with nonfinite_guard=False")
        v = int(self.nonfinite.item())
        return v if v > 0 else None


class Predictor:
    """Dense-grid inference (reference scripts/train_st_interp.py:1091-1107,1232-1248,1378-1409;
    evaluate_model :884-961): batched forward into a preallocated output, chunked so the workspace
    stays bounded.  Eager launches by default (MI355X, C2: 175 M obs/s at 262 144-row chunks, 163 M at
    65 536; replaying each full chunk from a hipGraph, `use_graph=True`, measured 153 M at 65 536 — the
    static-buffer copies and per-node overhead cost more than the launches)."""

    def __init__(self, model, chunk=262144, use_graph=False, force_dense=False):
        self.model = model
        self.dev = next(model.parameters()).device
        self.chunk = int(chunk)
        self.force_dense = force_dense
        self.state = model._step_state(self.dev, force_dense=force_dense, training=False)
        self._state_key = self._param_key()
        self.ws = torch.empty(N.step_workspace_bytes(self.state.basis, self.state.desc, self.chunk,
                                                     self.state.flags) // 4, device=self.dev)
        self.use_graph = use_graph
        self._graph = None
        self._warm = False
        self._in = (torch.empty(self.chunk, 2, device=self.dev), torch.empty(self.chunk, device=self.dev))
        self._out = torch.empty(self.chunk, model.output_dim, device=self.dev)

    def _param_key(self):
        """Identity + version of everything the cached state was derived from: parameter storage (a TrainStep
        built later re-points it), autograd versions (optimizer.step(), load_state_dict, EMA swaps through
        copy_) and the engine's step counter (its kernels write through raw pointers)."""
        m = self.model
        return (getattr(m, "_engine_version", 0),) + tuple((p.data_ptr(), p._version) for p in m.parameters()) \
            + tuple((b.data_ptr(), b._version) for b in m.buffers())

    def _refresh(self):
        """Rebuild the descriptors when the model changed since they were made (stale transposed copy of W0,
        stale output layer of the delta head, re-pointed parameter storage); a captured graph is dropped."""
        key = self._param_key()
        if key != self._state_key:
            self.state = self.model._step_state(self.dev, force_dense=self.force_dense, training=False)
            self._state_key = key
            self._graph = None

    def _enqueue(self, coords, t, out, B):
        st = self.state
        N.forward(st.basis, st.desc, st.params, coords, t, None, B, out, self.ws, st.flags, training=False)

    @torch.no_grad()
    def predict_grid(self, coords, t_values, max_rows=None):
        """The same S sites at every one of T times (what the reference's dense-grid callers loop over, one
        model call per time slice): coords (S,2), t_values (T,) -> (T, S, Q).  Layer 0's pre-activation is a
        per-site row plus a per-time row, so the basis evaluation and the gather of first-layer weights
        happen once per site; the rest of the network runs on the T*S rows.  Needs the window path (fixed
        grid knots, compact-support basis) and p = 0; otherwise falls back to predict() on the expanded rows."""
        self._refresh()
        st = self.state
        S, T = coords.shape[0], t_values.numel()
        coords = coords.contiguous().float()
        t_values = t_values.contiguous().float().view(-1)
        Q = self.model.output_dim
        m = self.model
        parts = m.p == 0 and bool(st.flags & N.FLAG_W0_T) and len(m.hidden_dims) >= 1 and S > 0 and T > 0
        window = parts and not m.spatial_basis.learnable and N.step_uses_window(st.basis, st.desc, st.flags)
        if not parts:
            cc = coords.repeat(T, 1)
            tt = t_values.repeat_interleave(S)
            return self.predict(cc, tt).view(T, S, Q)
        h0 = m.hidden_dims[0]
        sp = torch.empty(S, h0, device=self.dev)
        w0t = st.keep if st.keep is not None else m._body[0].weight.detach().t()    # (D, h0), contiguous
        # sites per call: the workspace's chunk on the window path; about 1 GiB of features on the other
        step = self.chunk if window else max(1024, min(self.chunk, (1 << 28) // max(m.input_dim, 1)))
        for s0 in range(0, S, step):
            n = min(step, S - s0)
            if window:
                N.spatial_partial(st.basis, st.desc, st.params, coords[s0:s0 + n], sp[s0:s0 + n], self.ws, st.flags)
            else:
                # materialising path (scattered or few knots, Gaussian bases): phi of the chunk's sites, then one
                # GEMM with the spatial rows of W0^T (p = 0: they are the first k_spatial rows)
                feats = m.build_features(None, coords[s0:s0 + n], torch.zeros(n, device=self.dev))
                N.gemm(feats, False, w0t[:m.k_spatial], True, n, h0, m.k_spatial, out=sp[s0:s0 + n])
        tp = torch.empty(T, h0, device=self.dev)
        N.temporal_partial(st.basis, st.desc, st.params, t_values, tp, st.flags)
        out = torch.empty(T * S, Q, device=self.dev)
        per = max(1, (max_rows or (1 << 30)) // max(S, 1))        # time slices per call (int32 row indices)
        for t0 in range(0, T, per):
            n = min(per, T - t0)
            N.forward_parts(st.desc, st.params, sp, tp[t0:t0 + n], out[t0 * S:(t0 + n) * S])
        return out.view(T, S, Q)

    @torch.no_grad()
    def predict(self, coords, t):
        """coords (N,2), t (N,) or (N,1) on the device -> (N,Q)."""
        if self.model.p != 0:
            raise RuntimeError("Predictor: covariates (p>0) are not wired into the dense-grid path")
        self._refresh()
        n = coords.shape[0]
        coords = coords.contiguous().float()
        t = t.contiguous().float().view(-1)
        out = torch.empty(n, self.model.output_dim, device=self.dev)
        for s in range(0, n, self.chunk):
            B = min(self.chunk, n - s)
            if self.use_graph and B == self.chunk and not self._warm:
                self._warm = True           # first full chunk eagerly (see TrainStep._step_graph)
                self._enqueue(coords[s:s + B], t[s:s + B], out[s:s + B], B)
            elif self.use_graph and B == self.chunk:
                if self._graph is None:
                    torch.cuda.synchronize()
                    self._graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._graph):
                        self._enqueue(self._in[0], self._in[1], self._out, B)
                self._in[0].copy_(coords[s:s + B]); self._in[1].copy_(t[s:s + B])
                self._graph.replay()
                out[s:s + B].copy_(self._out)
            else:
                self._enqueue(coords[s:s + B], t[s:s + B], out[s:s + B], B)
        return out
