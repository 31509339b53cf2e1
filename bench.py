#!/usr/bin/env python3
"""bench.py — train-step throughput of the ST-DADK interpolation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload c2|default|c4]

One "step" = one optimisation step on one mini-batch of synthetic KAUST-shaped observations resident
in HBM: feature build (multi-resolution Wendland + Gaussian bases) -> MLP forward -> MSE -> backward
-> (gradient exchange) -> clip + AdamW + EMA.  N > 1: one rank per GPU over RCCL, observation-sharded (weak
scaling: per-GPU batch fixed), launched by torch.distributed.run -- or by this script itself: `python bench.py
--gpus N` with WORLD_SIZE unset starts the N ranks as child processes BEFORE anything touches a GPU and relays
rank 0's line.  Gradient exchange per step (--dp-mode): `shard` = reduce-scatter + sharded AdamW/EMA + all-gather,
`allreduce` = one all-reduce + replicated optimiser (`auto` = whichever is faster in a 100-step calibration, both times
in `dp_mode_calibration`); default `shard`, the other mode is reported beside the headline.
Rank 0 prints ONE JSON line:
  value      whole-job observations/s: the MEDIAN of >= 10 timed windows of K steps each (every window bracketed by
             barrier + synchronize, max over ranks); min / max of the windows beside it
  roofline   the dominant kernel of the timed step: average launch duration measured live with HIP
             events on the launch stream (stdadk_profile_*), algorithmic bytes/flops per launch
  rbf_build  the standalone materialising feature builder (the "RBF-build GB/s" of BASELINE.json)
  cpu_baseline  the oracle's torch-CPU port of the reference's batch body on this box's host cores
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "st-dadk_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: synthetic 100k obs, 3-resolution Wendland basis (~10k knots), fp32
    "c2": dict(k_spatial_centers=[1024, 4096, 5184], k_temporal_centers=[10, 15, 45],
               hidden_dims=[256, 256, 128], n_obs=100_000,
               name="C2 synthetic 100k obs, 3-res Wendland 32^2+64^2+72^2=10304 knots + 70 temporal"),
    "default": dict(k_spatial_centers=[25, 81, 121], k_temporal_centers=[10, 15, 45],
                    hidden_dims=[256, 256, 128], n_obs=100_000, name="reference default 227 knots"),
    # BASELINE.json configs[2]: KAUST 2b full = 10 000 sites x 100 times = 1 M rows (the 2b files are missing
    # blobs in the reference, SURVEY.md 8: a 2b-SHAPED synthetic field), 3-resolution basis; printed as an fp32
    # line plus the "bf16 MLP with MFMA" variant (never as `value`)
    "c3": dict(k_spatial_centers=[1024, 4096, 5184], k_temporal_centers=[10, 15, 45],
               hidden_dims=[256, 256, 128], n_obs=1_000_000, sites=10_000, times=100,
               name="C3 KAUST-2b-shaped 10000 sites x 100 times = 1M rows, 3-res Wendland 10304 knots + 70 temporal"),
    "c4": dict(k_spatial_centers=[1024, 4096, 16384, 28224], k_temporal_centers=[10, 15, 45],
               hidden_dims=[256, 256, 128], n_obs=1_000_000, name="C4 synthetic 1M obs, 4-res 49728 knots"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_32x32x2_f32)


def synth(n, seed, device):
    """SURVEY.md §8(d) synthetic field: coords ~ U[0,1)^2, t = i/(T-1), T = 100,
    y = sin(4 pi x) cos(3 pi y) (1 + 0.5 sin(2 pi t)) + 0.1 N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    coords = torch.rand(n, 2, generator=g)
    t = torch.randint(0, 100, (n, 1), generator=g).float() / 99.0
    y = (torch.sin(4 * np.pi * coords[:, :1]) * torch.cos(3 * np.pi * coords[:, 1:2])
         * (1 + 0.5 * torch.sin(2 * np.pi * t)) + 0.1 * torch.randn(n, 1, generator=g))
    return coords.to(device), t.to(device), y.to(device)


def synth_sites(n_sites, n_times, seed, device):
    """KAUST 2b shape: the same `n_sites` scattered sites observed at every one of `n_times` times, rows in the
    reference's sample order (time-major, train_st_interp.py:413-450), t = t_idx/(T-1)."""
    g = torch.Generator().manual_seed(seed)
    sites = torch.rand(n_sites, 2, generator=g)
    coords = sites.repeat(n_times, 1)
    t = (torch.arange(n_times).float() / max(n_times - 1, 1)).repeat_interleave(n_sites).view(-1, 1)
    y = (torch.sin(4 * np.pi * coords[:, :1]) * torch.cos(3 * np.pi * coords[:, 1:2])
         * (1 + 0.5 * torch.sin(2 * np.pi * t)) + 0.1 * torch.randn(n_sites * n_times, 1, generator=g))
    return coords.to(device), t.to(device), y.to(device)


def csrc_hash():
    """sha256 (16 hex digits) over the kernel sources the library was built from: ties a committed PMC traffic
    figure to the exact kernels it was measured on (any source change makes it stale)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "st-dadk_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h")) +
                    glob.glob(os.path.join(base, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, workload, B, dtype, grid_threads=None):
    """(bytes per launch, source) from the committed rocprofv3 PMC summary, or (None, reason): the figure is only
    quoted for the kernel, workload, batch, dtype AND kernel sources it was measured on.  `grid_threads`: for a
    kernel the profiled run launches at several sizes, the entry of that grid size."""
    f = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(f):
        return None, "no profiles/pmc_traffic.json"
    try:
        tj = json.load(open(f))
    except Exception as e:                                   # noqa: BLE001
        return None, f"unreadable profiles/pmc_traffic.json: {e}"
    src = tj.get("source", {})
    want = dict(workload=workload, batch=B, dtype=dtype, csrc_sha16=csrc_hash())
    for k, v in want.items():
        if src.get(k) != v:
            return None, (f"stale: profiles/pmc_traffic.json was measured at {k}={src.get(k)!r}, this run has {v!r}; "
                          f"re-take the FETCH_SIZE / WRITE_SIZE passes (profiles/README.md)")
    short = kernel.replace("(stdadk::", "").replace("(", "").replace(")", "").split("<")[0]
    if grid_threads is not None and f"{short}@{grid_threads}" in tj.get("kernels", {}):
        short = f"{short}@{grid_threads}"
    elif grid_threads is not None and any(k.startswith(short + "@") for k in tj.get("kernels", {})):
        return None, f"{short}: no PMC entry for a grid of {grid_threads} threads"
    val = tj.get("kernels", {}).get(short)
    if val is None:
        return None, f"kernel {short} not in profiles/pmc_traffic.json"
    return val, {"file": "profiles/pmc_traffic.json", "commit": src.get("commit"), "csrc_sha16": src.get("csrc_sha16"),
                 "kernel": tj.get("kernel_names", {}).get(short, short),
                 "how": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; 2*FETCH + WRITE (gfx950 read correction)"}


def step_floors(B, H, Kt, Q, P_flat, nnz_per_obs):
    """Floors of ONE whole train step (window path): bytes that cross HBM at least once and flops of the
    mathematical products, against the chip's peaks.
      HBM : AdamW/EMA 36 B per parameter + the gradient written once (4 B) + per row the observation (16 B),
            what the forward keeps for the backward (xhat, act of every hidden layer: written and read once) and
            dZ of every hidden layer (written by the backward chain, read by the weight-gradient products)
      MFMA: forward of the layers after the first, their dA (no dA below layer 1) and dW products, the temporal
            rows of dW0, the output layer, and layer 0 on its non-zero (observation, knot) pairs, forward and dW0."""
    sumH = float(sum(H))
    hbm = 40.0 * P_flat + B * (16.0 + 4.0 * 6.0 * sumH)
    tail = sum(2.0 * H[i] * H[i - 1] for i in range(1, len(H))) + 2.0 * H[-1] * Q
    dA = sum(2.0 * H[i] * H[i - 1] for i in range(2, len(H))) + 2.0 * H[-1] * Q + 2.0 * H[1] * H[0]
    dW = sum(2.0 * H[i] * H[i - 1] for i in range(1, len(H))) + 2.0 * H[-1] * Q
    l0 = 2.0 * 2.0 * (nnz_per_obs + Kt) * H[0]
    fl = B * (tail + dA + dW + l0)
    return {"hbm_bytes": hbm, "hbm_us": hbm / (HBM_PEAK_GBS * 1e9) * 1e6, "flops": fl,
            "mfma_us": fl / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e6}


def time_events(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3      # seconds per call


def host_cores():
    """CPU threads this process may really use: cgroup quota if set, else affinity, capped at the
    box's per-GPU CPU share (16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(wl, batch, dropout, budget_s=12.0, cores=None):
    """The oracle's torch-CPU port of the reference batch body, timed on this box's host cores on a
    bounded sample (a few steps of the same batch size) with `cores` torch threads (default: all this process
    may use)."""
    from oracle import torch_port as tp
    cores = host_cores() if cores is None else max(1, min(int(cores), host_cores()))
    torch.set_num_threads(cores)
    cfg = dict(p=0, k_spatial_centers=wl["k_spatial_centers"], k_temporal_centers=wl["k_temporal_centers"],
               hidden_dims=wl["hidden_dims"], layernorm=True, dropout=dropout, basis="wendland", output_dim=1)
    model = tp.PortModel(cfg, seed=0)
    tr = tp.PortTrainer(model, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999)
    coords, t, y = synth(batch, 2025, "cpu")
    X = torch.zeros(batch, 0)
    tr.step(X, coords, t, y)                       # warm-up (allocations, MKL init)
    n, t0 = 0, time.perf_counter()
    while True:
        tr.step(X, coords, t, y)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    return dict(value=batch * n / el, unit="obs/s", cores=cores, kind="port",
                sample=f"{n} train steps of batch {batch} (oracle/torch_port.py: the reference's op sequence, "
                       f"same model/config/optimizer, {el:.1f} s of CPU work, {cores} torch threads)")


def kernel_work(name, B, D, H, P_flat, nnz_pairs, Kt, Q=1):
    """Algorithmic work of one launch of a library kernel: (bound, amount, unit) or None.
    HBM kernels are priced in bytes that MUST cross HBM, MFMA kernels in flops of the mathematical
    product (DESIGN.md 'kernels')."""
    if name.startswith("adamw_ema_kernel"):
        return "hbm", 36.0 * P_flat, "B"            # read p,g,m,v,ema + write p,m,v,ema (4 B each)
    if name.startswith("sumsq_kernel"):
        return "hbm", 4.0 * P_flat, "B"
    if "rbf_build_kernel" in name:
        return "hbm", B * (12.0 + 4.0 * D), "B"
    if "dw_all_kernel" in name:
        # grouped dW_l = dZ_l^T a_{l-1} (l >= 1, + temporal rows of dW0^T) and the per-knot gather of dW0^T
        fl = sum(2.0 * B * H[i] * H[i - 1] for i in range(1, len(H))) + 2.0 * B * Kt * H[0]
        return "mfma", fl + 2.0 * nnz_pairs * H[0], "flop"
    if "l1_window_bwd_kernel" in name:
        return "mfma", 2.0 * nnz_pairs * H[0], "flop"    # dW0^T rows: one fma per non-zero (obs,knot) x H
    if "l1_window_fwd_kernel" in name:
        return "mfma", 2.0 * (nnz_pairs + B * Kt) * H[0], "flop"
    if "l1_tail_kernel" in name:
        # layer 0 on the window path + forward and backward of the layers after it, in one launch
        fl = sum(2.0 * B * H[i] * H[i - 1] for i in range(1, len(H))) + 2.0 * B * H[-1] * Q
        return "mfma", 2.0 * fl + 2.0 * (nnz_pairs + B * Kt) * H[0], "flop"
    if "tail_fwd_bwd_kernel" in name:
        # forward z = a W^T and backward dA = dZ W of the layers after the first, in one launch
        fl = sum(2.0 * B * H[i] * H[i - 1] for i in range(1, len(H))) + 2.0 * B * H[-1] * Q
        return "mfma", 2.0 * fl, "flop"
    if "tail_fwd_kernel" in name or "tail_bwd_kernel" in name:
        # Linear layers after the first (+ output layer): forward z = a W^T, backward dA = dZ W
        fl = sum(2.0 * B * H[i] * H[i - 1] for i in range(1, len(H))) + 2.0 * B * H[-1] * Q
        return "mfma", fl, "flop"
    if "gemm_tn_grouped_kernel" in name:
        # dW_l = dZ_l^T a_{l-1} for l >= 1, plus the temporal rows of dW0^T
        fl = sum(2.0 * B * H[i] * H[i - 1] for i in range(1, len(H))) + 2.0 * B * Kt * H[0]
        return "mfma", fl, "flop"
    if name.startswith("gemm_f32_kernel") and "M=" in name:
        f = dict(kv.split("=") for kv in name.split() if "=" in kv)
        return "mfma", 2.0 * int(f["M"]) * int(f["N"]) * int(f["K"]), "flop"
    return None


def self_launch(n, argv, script=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1) and relay rank 0's JSON line.  Called before
    this process has touched a GPU (importing torch does not initialise HIP); the parent never does -- it only waits --
    and no process is replaced (no exec).  Exit code: 0 when every rank exits 0, else the first failing rank's."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0 inherits stdout (its one JSON line is the record); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for o in pending:              # a rank that died leaves its partners waiting in a collective
                        procs[o].terminate()
            time.sleep(0.2)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU mini-batch (reference YAML: 4096)")
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS))
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: the Linear layers after the first on the bf16 matrix cores (BASELINE config C3); the "
                         "headline `value` is always the f32 line, bf16 runs are labelled as such")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a hipGraph (N = 1).  Off by default: on this ROCm the eager "
                         "launch chain is ~4 %% faster at B = 4096 (0.180 vs 0.187 ms), the host stays ahead")
    ap.add_argument("--no-graph", action="store_true", help="(kept for older command lines; eager is the default)")
    ap.add_argument("--dense", action="store_true", help="force the materialising (dense) kernels")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not overlap the next batch's gather + binning with the current step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--clock-warmup-ms", type=float, default=150.0,
                    help="milliseconds of an unrelated kernel (the standalone feature builder) before the warm-up steps, "
                         "to leave idle clocks; 0 = off")
    ap.add_argument("--no-sweep", action="store_true", help="skip the extra per-GPU batch sizes (N = 1 only)")
    ap.add_argument("--windows", type=int, default=10,
                    help="timed windows of --steps steps each; `value` is the median window (>= 1)")
    ap.add_argument("--dp-mode", default="shard", choices=["shard", "allreduce", "auto"],
                    help="N > 1: gradient exchange of the headline line (the other mode is reported beside it); auto = "
                         "whichever of the two is faster in a short calibration on this machine (both times reported; "
                         "not the default: in the two-rank gloo rehearsal the second engine of the calibration left "
                         "the first one slower)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # STNF_BENCH_BACKEND=gloo + STNF_BENCH_ONE_GPU=1: rehearsal of the multi-rank flow on a one-GPU box
        backend = os.environ.get("STNF_BENCH_BACKEND", "nccl")
        if os.environ.get("STNF_BENCH_ONE_GPU"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from stnf.models import STInterpMLP
    from stnf.engine import TrainStep
    from stnf import _native as N

    wl = WORKLOADS[args.workload]
    B = args.batch
    torch.manual_seed(0)                                     # identical initial weights on all ranks
    model = STInterpMLP(p=0, k_spatial_centers=wl["k_spatial_centers"],
                        k_temporal_centers=wl["k_temporal_centers"], hidden_dims=wl["hidden_dims"],
                        dropout=args.dropout, layernorm=True).to(dev)
    model.train()
    n_obs = wl["n_obs"]
    if "sites" in wl:
        coords, t, y = synth_sites(wl["sites"], wl["times"], 2025 + rank, dev)
    else:
        coords, t, y = synth(n_obs, 2025 + rank, dev)       # each rank owns its shard of observations
    batches_per_epoch = max(n_obs // B, 1)
    dp_auto = world > 1 and args.dp_mode == "auto"
    if args.dp_mode == "auto":
        args.dp_mode = "shard"                 # calibrated against all-reduce below (N > 1)
    shard = world > 1 and args.dp_mode == "shard"

    def make_engine(shard_optimizer, mdl=None):
        return TrainStep(model if mdl is None else mdl, lr=2e-2, weight_decay=5e-4, grad_clip=10.0,
                         ema_decay=1.0 - 1.0 / (10.0 * batches_per_epoch), max_batch=B,
                         use_graph=args.graph and world == 1, force_dense=args.dense, dtype=args.dtype,
                         shard_optimizer=shard_optimizer)
    eng = make_engine(shard)
    dp_fallback = None
    perm = torch.randperm(n_obs, device=dev)

    def batch(i):
        idx = perm[(i % batches_per_epoch) * B:(i % batches_per_epoch) * B + B]
        return coords[idx], t[idx], y[idx]

    def run(k0, k, e=None):
        # the observation shard stays resident in HBM; a step takes the index slice of its batch
        e = eng if e is None else e
        for i in range(k0, k0 + k):
            j, jn = i % batches_per_epoch, (i + 1) % batches_per_epoch
            # the next batch's rows are announced so that its gather + binning run on a side stream
            # while this step computes (software pipelining of the batch preparation)
            e.step_indexed(coords, t, y, perm[j * B:j * B + B], global_rows=B * world,
                           next_idx=None if args.no_pipeline else perm[jn * B:jn * B + B])

    # device warm-up BEFORE the W warm-up steps: the chip's power state takes ~15 ms of load to settle and falls back
    # within 20 ms of idling (tools/startup_latency.py: a step's kernels all run ~6 % slower during the first ~100
    # steps after an idle period, whatever the data), and W = 5 steps are 0.6 ms.  0.15 s of the standalone feature
    # builder -- no training step, nothing of the timed work, the timed model untouched -- recovers a third of that
    # for a 20-step timed region (0.132 -> 0.130 ms per step; a 200-step region reads 0.124).  --clock-warmup-ms 0
    # switches it off; the figure is in `config`.
    if args.clock_warmup_ms > 0:
        cw_feats = torch.empty(4096, (model.input_dim + 31) // 32 * 32, device=dev)
        cw_c, cw_t = coords[:4096].contiguous(), t[:4096].contiguous().view(-1)
        t_cw = time.perf_counter()
        while (time.perf_counter() - t_cw) * 1e3 < args.clock_warmup_ms:
            for _ in range(50):
                N.rbf_build(cw_c, cw_t, None, model.spatial_basis.centers, model.spatial_basis._bandwidths, "wendland",
                            model.temporal_basis.centers, model.temporal_basis.bandwidths, cw_feats)
            torch.cuda.synchronize()
        del cw_feats
    try:
        run(0, args.warmup)
        torch.cuda.synchronize()
    except Exception as e:                                   # noqa: BLE001
        # N > 1 only: the sharded exchange (in-place reduce-scatter / all-gather) was refused by this build's
        # collectives library -- a deterministic refusal is the same on every rank, so all of them fall back to the
        # all-reduce exchange together; the line says so.  Anything else is re-raised.
        if not shard:
            raise
        dp_fallback = f"{type(e).__name__}: {e}"[:300]
        shard = False
        args.dp_mode = "allreduce"
        eng = make_engine(False)
        run(0, args.warmup)
    # --dp-mode auto (N > 1): the sharded exchange is three collectives per step, the all-reduce one; which of them is
    # faster for an 11 MB gradient depends on the machine's collectives (latency against the optimiser traffic saved).
    # Both are timed on 100 steps (barrier + synchronize brackets, MAX over ranks -- so every rank decides alike), the
    # all-reduce engine on a second model of the same shape so that the headline model stays bound to ITS engine.
    dp_calibration = None
    if dp_auto and dp_fallback is None:
        def quick(e, k=100):
            run(0, 5, e)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(5, k, e)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            tm = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            return tm.item() / k * 1e3
        t_shard = quick(eng)
        torch.manual_seed(0)
        model_b = STInterpMLP(p=0, k_spatial_centers=wl["k_spatial_centers"],
                              k_temporal_centers=wl["k_temporal_centers"], hidden_dims=wl["hidden_dims"],
                              dropout=args.dropout, layernorm=True).to(dev)
        model_b.train()
        eng_b = make_engine(False, model_b)
        t_ar = quick(eng_b)
        del eng_b, model_b
        gc.collect()
        torch.cuda.synchronize()
        dp_calibration = {"steps": 100, "shard_ms_per_step": t_shard, "allreduce_ms_per_step": t_ar}
        if t_ar < t_shard:
            shard = False
            args.dp_mode = "allreduce"
            eng = make_engine(False)
            run(0, args.warmup)
            torch.cuda.synchronize()
        dp_calibration["chosen"] = args.dp_mode
    # >= 1 timed windows of EXACTLY --steps steps, each bracketed by barrier + synchronize on both sides and reduced
    # with MAX over ranks; `value` is the median window (a 20-step window is 2.4 ms: one window alone moves by a few
    # per cent from run to run with the chip's clock state), the fastest and slowest window are reported beside it
    def timed_window(k0):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(k0, args.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = tmax.item()
        return dt
    n_win = max(1, args.windows)
    win_s = [timed_window(args.warmup + w * args.steps) for w in range(n_win)]
    el = sorted(win_s)[(n_win - 1) // 2]                       # median window (lower median for an even count)
    loss = eng.mean_loss()

    # ---- per-kernel device time of the step, live, with HIP events on the launch stream.  Every
    # rank runs these steps (they contain the gradient all-reduce); rank 0 reports.
    D = model.input_dim
    H = wl["hidden_dims"]
    Kt = model.k_temporal
    c, tt, yy = batch(0)
    c, tt, yy = c.contiguous(), tt.contiguous().view(-1), yy.contiguous()
    n_prof = 10
    eng.time_allreduce = world > 1
    N.profile_enable(True)
    for _ in range(n_prof):
        eng._enqueue(None, c, tt, yy, B, B * world)
    torch.cuda.synchronize()
    eng.time_allreduce = False
    # the step's collectives (all-reduce, or reduce-scatter + clip-partials all-reduce + all-gather): sum of their
    # event brackets per step
    allreduce_ms = (sum(e0.elapsed_time(e1) for e0, e1 in eng.allreduce_events) / n_prof if world > 1 else None)
    eng.allreduce_events = []
    recs = N.profile_collect() if rank == 0 else None      # the step's launches only: collected before anything else runs
    N.profile_enable(False)
    # what the same event bracket reads around a launch that does no work (a one-thread kernel on a scratch word): the
    # part of every per-kernel figure below that is the bracket, not the kernel (rocprofv3's kernel trace has none)
    bracket_us = None
    if rank == 0:
        scratch_i = torch.zeros(1, dtype=torch.int32, device=dev)
        N.profile_enable(True)
        for _ in range(20):
            N.step_advance(scratch_i)
        torch.cuda.synchronize()
        br = sorted(ms for _, ms in N.profile_collect())
        N.profile_enable(False)
        bracket_us = br[len(br) // 2] * 1e3
    if world > 1:
        dist.barrier()

    # ---- N > 1: the configuration BASELINE names for the 8-GPU run (C4: 4 resolutions, 49 728 knots, 12.85 M
    # parameters = a 51 MB gradient) at the per-GPU batch SURVEY.md 8(e) sizes for it (>= 16 384 rows), beside the
    # headline workload above.  Every rank takes part (the step contains the all-reduce).
    def weak_line(wname, B4, mode):
        """One more weak-scaling line at N > 1: workload `wname` at B4 rows per GPU with gradient exchange `mode`,
        timed like the headline (barrier + synchronize on both sides, MAX over ranks), plus the time of its
        collectives per step.  Every rank takes part."""
        w4 = WORKLOADS[wname]
        n4 = max(w4["n_obs"] // world, 4 * B4)
        torch.manual_seed(0)
        m4 = STInterpMLP(p=0, k_spatial_centers=w4["k_spatial_centers"], k_temporal_centers=w4["k_temporal_centers"],
                         hidden_dims=w4["hidden_dims"], dropout=args.dropout, layernorm=True).to(dev)
        m4.train()
        c4c, c4t, c4y = synth(n4, 4025 + rank, dev)
        e4 = TrainStep(m4, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999, max_batch=B4, dtype=args.dtype,
                       shard_optimizer=(mode == "shard"))
        perm4 = torch.randperm(n4, device=dev)
        nb4 = n4 // B4

        def run4(k0, k):
            for i in range(k0, k0 + k):
                j, jn = i % nb4, (i + 1) % nb4
                e4.step_indexed(c4c, c4t, c4y, perm4[j * B4:j * B4 + B4], global_rows=B4 * world,
                                next_idx=None if args.no_pipeline else perm4[jn * B4:jn * B4 + B4])
        k4 = max(10, min(args.steps, 40))
        run4(0, 5)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t4 = time.perf_counter()
        run4(5, k4)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        el4 = torch.tensor([time.perf_counter() - t4], device=dev, dtype=torch.float64)
        dist.all_reduce(el4, op=dist.ReduceOp.MAX)
        e4.time_allreduce = True
        run4(5 + k4, 10)
        torch.cuda.synchronize()
        ar4 = sum(a0.elapsed_time(a1) for a0, a1 in e4.allreduce_events) / 10
        dist.barrier()
        res = {"workload": w4["name"], "per_gpu_batch": B4, "global_batch": B4 * world, "n_obs_per_gpu": n4,
               "dp_mode": mode, "obs_per_s": world * B4 * k4 / el4.item(), "ms_per_step": el4.item() / k4 * 1e3,
               "steps": k4, "gradient_bytes": 4 * e4.flat.numel(), "collectives_ms_per_step": ar4, "scaling": "weak",
               "dtype": args.dtype}
        del e4, m4, c4c, c4t, c4y, perm4
        gc.collect()
        torch.cuda.synchronize()
        return res

    def c5_sharded_line():
        """BASELINE config C5 on N GPUs: the 10 M-point prediction grid (100 000 sites x 100 times), sites sharded over
        the ranks with no collective on the data path, the (T, S_r, Q) blocks all-gathered at the end
        (stnf.distributed.sharded_predict_grid; replaces the per-slice loop of scripts/train_st_interp.py:1232-1248)."""
        from stnf.engine import Predictor
        from stnf import distributed as DD
        S5, T5 = 100_000, 100
        g5 = torch.Generator().manual_seed(5)
        c5 = torch.rand(S5, 2, generator=g5).to(dev)                 # the same sites on every rank
        tv5 = torch.arange(T5, device=dev, dtype=torch.float32) / (T5 - 1)
        model.eval()
        pr = Predictor(model)
        out = DD.sharded_predict_grid(pr.predict_grid, c5, tv5)      # warm-up (+ first-call attributes)
        res = {"points": S5 * T5, "sites": S5, "times": T5, "n_gpus": world}
        for label, fn in (("with_final_allgather", lambda: DD.sharded_predict_grid(pr.predict_grid, c5, tv5)),
                          ("shards_only", lambda: pr.predict_grid(c5[slice(*DD.shard_range(S5, rank, world))], tv5))):
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            dt = torch.tensor([(time.perf_counter() - t1) / 3], device=dev, dtype=torch.float64)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            res[label] = {"points_per_s": S5 * T5 / dt.item(), "ms_per_call": dt.item() * 1e3}
        del out
        model.train()
        return res

    # ---- N > 1: beside the headline line (the reference's batch of 4096 rows per GPU, where the 11 MB gradient
    # all-reduce is as long as the step), (a) the configuration BASELINE names for the 8-GPU run -- C4: 4 resolutions,
    # 49 728 knots, 12.85 M parameters = a 51 MB gradient -- at the per-GPU batch SURVEY.md 8(e) sizes for it
    # (>= 16 384 rows), and (b) the headline model at 65 536 rows per GPU, where the same collective is ~10 % of the step
    extra_lines = {}
    if world > 1:
        other = "allreduce" if args.dp_mode == "shard" else "shard"
        jobs = [("other_dp_mode_line", args.workload, B, other),
                ("c4_weak_scaling_line", "c4", max(16384, B), args.dp_mode),
                ("c4_weak_scaling_line_other_dp_mode", "c4", max(16384, B), other),
                ("c2_b65536_weak_scaling_line", "c2", 65536, args.dp_mode)]
        for key, wname, bb, mode in jobs:
            if wname == args.workload and bb == B and mode == args.dp_mode:
                continue
            try:
                extra_lines[key] = weak_line(wname, bb, mode)
            except Exception as e:                      # noqa: BLE001  (the headline line must survive these extras)
                extra_lines[key] = {"error": f"{type(e).__name__}: {e}"}
        try:
            extra_lines["inference_c5_10M_sharded"] = c5_sharded_line()
        except Exception as e:                          # noqa: BLE001
            extra_lines["inference_c5_10M_sharded"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        agg = {}
        for name, ms in recs:
            a = agg.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += ms
        per_step = sorted(((nm, cnt / n_prof, tot / cnt * 1e3, tot / n_prof * 1e3) for nm, (cnt, tot) in agg.items()),
                          key=lambda r: -r[3])      # (name, launches/step, avg us, us/step)
        kernels_us = {nm.replace("(", "").replace(")", "")[:80]: round(us, 2) for nm, _, _, us in per_step}
        # exact count of non-zero (observation, knot) pairs of this batch, for the window kernels
        phi = model.spatial_basis(c)
        nnz = int((phi != 0).sum().item())
        del phi
        P_flat = eng.flat.numel()
        dom = None
        for nm, lps, avg_us, us in per_step:
            w = kernel_work(nm.replace("(stdadk::", "").replace("(", ""), B, D, H, P_flat, nnz, Kt)
            if w is not None:
                dom = (nm, avg_us, w)
                break
        roof = None
        if dom:
            nm, avg_raw, (bound, amount, _) = dom
            # the kernel's duration = what its event bracket reads minus what the same bracket reads around a launch
            # that does no work (5 us on MI355X with fence-free events): agrees with the rocprofv3 kernel-trace average
            # of the same command (profiles/) to ~1 us; the raw figure stays in the line
            avg_us = max(avg_raw - (bracket_us or 0.0), 1e-3)
            if bound == "hbm":
                ach = amount / (avg_us * 1e-6) / 1e9
                roof = {"kernel": nm, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": avg_us,
                        "algorithmic_bytes_per_launch": amount}
            else:
                ach = amount / (avg_us * 1e-6) / 1e12
                roof = {"kernel": nm, "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                        "avg_launch_us": avg_us, "algorithmic_flops_per_launch": amount}
            roof["traffic"], roof["traffic_source"] = pmc_traffic(nm, args.workload, B, args.dtype)
            roof["avg_launch_us_event_bracket_raw"] = avg_raw
            roof["event_bracket_of_an_empty_launch_us"] = bracket_us
        # ---- the standalone materialising feature builder ("RBF-build GB/s")
        feats = torch.empty(B, (D + 31) // 32 * 32, device=dev)
        t_rbf = time_events(lambda: N.rbf_build(c, tt, None, model.spatial_basis.centers,
                                                model.spatial_basis._bandwidths, "wendland",
                                                model.temporal_basis.centers, model.temporal_basis.bandwidths,
                                                feats), 50)
        rbf_bytes = B * (12 + 4 * D)                         # 12 B read + 4*D written / obs
        rbf_gbs = rbf_bytes / t_rbf / 1e9
        ldf = (D + 31) // 32 * 32

        def rbf_grid(rows):          # threads of one launch: column tiles of 1024 x 4 rows per workgroup x 256 threads
            return -(-ldf // 1024) * -(-rows // 4) * 256
        rbf_traffic, rbf_traffic_src = pmc_traffic("rbf_build_kernel", args.workload, B, args.dtype, rbf_grid(B))
        del feats
        # the same builder on a footprint well past the 256 MiB Infinity Cache (FETCH/WRITE_SIZE and a short
        # timed loop both see cache hits below it): rows so that the written features are >= 640 MB
        B_big = max(B, -(-640_000_000 // (4 * ((D + 31) // 32 * 32))) // 1024 * 1024 + 1024)
        cb, tb_, _ = synth(B_big, 7, dev)
        feats_big = torch.empty(B_big, (D + 31) // 32 * 32, device=dev)
        t_big = time_events(lambda: N.rbf_build(cb, tb_.view(-1), None, model.spatial_basis.centers,
                                                model.spatial_basis._bandwidths, "wendland",
                                                model.temporal_basis.centers, model.temporal_basis.bandwidths,
                                                feats_big), 20)
        big_bytes = B_big * (12 + 4 * D)
        rbf_big = {"bound": "hbm", "achieved": big_bytes / t_big / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": big_bytes / t_big / 1e9 / HBM_PEAK_GBS, "rows": B_big, "bytes_per_launch": big_bytes,
                   "avg_launch_us": t_big * 1e6,
                   "note": "footprint past the 256 MiB Infinity Cache: every byte of the write stream reaches HBM"}
        rbf_big["traffic"], rbf_big["traffic_source"] = pmc_traffic("rbf_build_kernel", args.workload, B, args.dtype,
                                                                    rbf_grid(B_big))
        del feats_big, cb, tb_
        Q = model.output_dim
        floors = step_floors(B, H, Kt, Q, P_flat, nnz / B)
        ms_step = el / args.steps * 1e3
        out = {
            "metric": "train-step samples/sec (obs points/sec)", "value": args.gpus * B * args.steps / el,
            "value_windows": {"n": n_win, "steps_each": args.steps, "statistic": "median window",
                              "min": args.gpus * B * args.steps / max(win_s), "max": args.gpus * B * args.steps / min(win_s),
                              "ms_per_step_each": [round(w / args.steps * 1e3, 5) for w in win_s]},
            "unit": "obs/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": wl["name"], "per_gpu_batch": B, "global_batch": B * args.gpus,
                       "n_obs_per_gpu": n_obs, "dropout": args.dropout, "layernorm": True,
                       "optimizer": "AdamW lr 2e-2 wd 5e-4 clip 10 + EMA",
                       "path": (("index-window layer 1 (compact support) + " if eng.uses_window
                                 else "materialised features + dense ") +
                                ("fp32 MFMA MLP" if args.dtype == "f32" else
                                 "bf16-operand MFMA MLP after the first layer (fp32 accumulate / LayerNorm / master weights)")),
                       "hipgraph": bool(eng.use_graph),
                       "batch_preparation": "pipelined on a side stream" if (eng.uses_window and not args.no_pipeline
                                                                              and not eng.use_graph) else "in the step",
                       "device_warmup_ms_before_warmup_steps": args.clock_warmup_ms,
                       "parallelism": f"dp{args.gpus}",
                       "gradient_exchange": ("none (one GPU)" if world == 1 else
                                             ("reduce-scatter + sharded AdamW/EMA + all-gather" if shard
                                              else "all-reduce + replicated AdamW/EMA"))},
            "roofline": roof,
            "rbf_build": {"bound": "hbm", "achieved": rbf_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": rbf_gbs / HBM_PEAK_GBS, "bytes_per_obs": 12 + 4 * D, "avg_launch_us": t_rbf * 1e6,
                          "traffic": rbf_traffic, "traffic_source": rbf_traffic_src,
                          "note": "one launch of stdadk_rbf_build_f32 over all column tiles; at this batch the "
                                  "features fit the 256 MiB Infinity Cache -- rbf_build_past_l3 is the HBM figure"},
            "rbf_build_past_l3": rbf_big,
            "step_floor_us": {"hbm": round(floors["hbm_us"], 2), "mfma_f32": round(floors["mfma_us"], 2),
                              "hbm_bytes": floors["hbm_bytes"], "flops": floors["flops"],
                              "step_us": round(ms_step * 1e3, 2),
                              "frac_of_step": round(max(floors["hbm_us"], floors["mfma_us"]) / (ms_step * 1e3), 4),
                              "note": "floors of the WHOLE step (bench.step_floors): bytes that must cross HBM at "
                                      "8 TB/s, flops of the mathematical products at the fp32 MFMA peak"},
            "kernels_us_per_step": kernels_us,
            "kernels_us_note": "event brackets per launch; each includes the bracket of an empty launch "
                               "(roofline.event_bracket_of_an_empty_launch_us), which the rocprofv3 averages in profiles/ do not",
            "kernel_time_us_per_step": round(sum(r[3] for r in per_step), 1),
            "nonzero_obs_knot_pairs_per_obs": nnz / B,
            "final_mean_loss": loss,
        }
        if world > 1:
            # the one collective of the path, timed with events around it on the step's stream (rank 0's view;
            # it includes waiting for the slowest rank to arrive)
            if dp_calibration:
                out["dp_mode_calibration"] = dp_calibration
            if dp_fallback:
                out["dp_mode_fallback"] = {"requested": "shard", "used": "allreduce", "error": dp_fallback}
            out["collectives"] = {"dp_mode": args.dp_mode, "ms_per_step": allreduce_ms, "gradient_bytes": 4 * P_flat,
                                  "algorithm_bandwidth_GBs": 4 * P_flat / (allreduce_ms * 1e-3) / 1e9 if allreduce_ms else None,
                                  "note": "event brackets on the step's stream around every collective of a step (rank 0: "
                                          "includes waiting for the slowest rank)"}
            out.update(extra_lines)
        if args.gpus == 1 and not args.no_sweep:
            def timed(b2, k2, model_kw=None, eng_kw=None, graph=None, data=None):
                """obs/s of the same fused step for another batch size / objective / knot mode (`data`: another
                resident observation set (coords, t, y, perm) instead of the headline's)."""
                torch.manual_seed(0)
                coords_, t_, y_, perm_ = data if data is not None else (coords, t, y, perm)
                n_obs_ = coords_.shape[0]
                mk = dict(p=0, k_spatial_centers=wl["k_spatial_centers"],
                          k_temporal_centers=wl["k_temporal_centers"], hidden_dims=wl["hidden_dims"],
                          dropout=args.dropout, layernorm=True)
                mk.update(model_kw or {})
                m2 = STInterpMLP(**mk).to(dev)
                m2.train()
                e2 = TrainStep(m2, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999, max_batch=b2,
                               use_graph=args.graph if graph is None else graph, **(eng_kw or {}))
                nb2 = max(n_obs_ // b2, 1)
                def sl(i):
                    return perm_[(i % nb2) * b2:(i % nb2) * b2 + b2]
                pipe = not args.no_pipeline
                for i in range(5):
                    e2.step_indexed(coords_, t_, y_, sl(i), next_idx=sl(i + 1) if pipe else None)
                torch.cuda.synchronize()
                # median of three windows of k2 steps: one window alone is a few ms, and a single host hiccup on the
                # box (the step is four launches per ~0.1 ms) once read a variant at half its rate
                wins = []
                for w in range(3):
                    t1 = time.perf_counter()
                    for i in range(5 + w * k2, 5 + (w + 1) * k2):
                        e2.step_indexed(coords_, t_, y_, sl(i), next_idx=sl(i + 1) if pipe else None)
                    torch.cuda.synchronize()
                    wins.append(time.perf_counter() - t1)
                dt = sorted(wins)[1]
                res = {"obs_per_s": b2 * k2 / dt, "ms_per_step": dt / k2 * 1e3,
                       "ms_per_step_windows": [round(x / k2 * 1e3, 5) for x in wins],
                       "path": "window" if e2.uses_window else "materialised"}
                # engines are released HERE, outside any timed region: an engine that ran from a hipGraph holds the
                # graph's private memory pool, and when Python's cycle collector got round to it in the middle of a LATER
                # variant's timed loop, that loop stood still for ~70 ms (round 3: the first bf16 line at 16 384 rows read
                # 12-16 M obs/s instead of 77 M)
                del e2, m2
                gc.collect()
                torch.cuda.synchronize()
                return res
            # the survey's per-GPU batch sweep: same model, same step, other batch sizes
            out["batch_sweep"] = {str(b2): timed(b2, 40) for b2 in (16384, 65536) if b2 != B and b2 <= n_obs}
            # the same step replayed from a hipGraph / launched eagerly (whichever `value` did not use)
            out["other_launch_mode"] = dict(timed(B, 100, graph=not args.graph), hipgraph=not args.graph)
            # BASELINE config C3's "bf16 MLP with MFMA": the same step with bf16 operands in the Linear layers after
            # the first (fp32 accumulation / LayerNorm / loss / master weights); labelled, never `value`
            if args.dtype == "f32":
                out["bf16_mlp"] = {"dtype": "bf16 operands, f32 accumulate", "note": "TrainStep(dtype='bf16'); parity: "
                                   "tests/test_gpu_bf16.py (emulation of the operand rounding + float64 goldens)",
                                   **{str(b2): timed(b2, 40 if b2 > B else 100, eng_kw=dict(dtype="bf16"))
                                      for b2 in (B, 16384, 65536) if b2 <= n_obs}}
            # the north_star's wording, "1M-point / 3-resolution-basis config at 1 GPU": BASELINE config C3's shape
            # (KAUST 2b: 10 000 sites x 100 times = 1 M rows resident in HBM, 3-resolution basis) at the same batch,
            # fp32 and with bf16 operands after layer 0; same per-step work as the headline, another gather footprint
            if args.workload != "c3":
                w3 = WORKLOADS["c3"]
                d3 = synth_sites(w3["sites"], w3["times"], 2025, dev)
                d3 = d3 + (torch.randperm(d3[0].shape[0], device=dev),)
                mk3 = dict(k_spatial_centers=w3["k_spatial_centers"], k_temporal_centers=w3["k_temporal_centers"],
                           hidden_dims=w3["hidden_dims"])
                c3v = {"workload": w3["name"], "rows_resident": int(d3[0].shape[0]), "per_gpu_batch": B,
                       "f32": timed(B, 200, mk3, data=d3), "bf16": timed(B, 200, mk3, dict(dtype="bf16"), data=d3)}
                del d3
            else:
                c3v = None
            # the "next" rows of SURVEY.md §8(f) on the same workload and batch: multi-quantile objectives
            # (N3) and learnable knots (N2, materialising path); reported beside, never as, `value`
            taus = [0.05, 0.25, 0.5, 0.75, 0.95]
            out["variants"] = {
                **({"c3_1M_rows_f32": dict(c3v["f32"], workload=c3v["workload"], rows_resident=c3v["rows_resident"]),
                    "c3_1M_rows_bf16": dict(c3v["bf16"], workload=c3v["workload"], rows_resident=c3v["rows_resident"],
                                            dtype="bf16 operands after layer 0, f32 accumulate")} if c3v else {}),
                "multi_quantile_q5_noncrossing": timed(B, 60, dict(output_dim=5), dict(
                    loss="pinball", quantile_levels=taus, non_crossing_weight=0.5)),
                "multi_quantile_q5_delta_head": timed(B, 60, dict(output_dim=5, use_delta_reparameterization=True),
                                                      dict(loss="pinball", quantile_levels=taus,
                                                           non_crossing_lambda=0.05)),
                "learnable_knots": timed(B, 30, dict(spatial_learnable=True, gradient_damping=True,
                                                     damping_threshold=0.0, damping_strength=5.0),
                                         dict(domain_penalty_weight=0.01)),
                # sparse-group lasso on the first layer (train_st_interp.py:674-691)
                "sparse_group_penalty": timed(B, 60, None, dict(sparsity_penalty_type="sparse_group",
                                                                sparsity_lambda_l1=1e-4, sparsity_lambda_group=1e-3)),
            }
            # the reference's SHIPPED YAML (configs/*.yaml: 227 GMM-initialised learnable knots, 5 quantiles
            # with the delta-free head, batch 4096): scattered knots => materialising kernels
            np.random.seed(0)
            site = coords[:20000].cpu().numpy()
            out["variants"]["shipped_yaml_227_gmm_learnable_mq5"] = timed(
                B, 60, dict(k_spatial_centers=[25, 81, 121], spatial_learnable=True, spatial_init_method="gmm",
                            train_coords=site, gradient_damping=True, damping_threshold=0.0, damping_strength=5.0,
                            output_dim=5),
                dict(loss="pinball", quantile_levels=taus, non_crossing_weight=0.5, domain_penalty_weight=0.01))
            out["variants"]["ref_default_227_uniform_mse"] = timed(B, 60, dict(k_spatial_centers=[25, 81, 121]))
            # DA-STDK with NON-GRID knots at the C2 table size (random_site initialiser: knots drawn from the sites,
            # learnable): the window kernels over per-level knot cell lists against the materialising kernels
            np.random.seed(1)
            sc_kw = dict(spatial_learnable=True, spatial_init_method="random_site", train_coords=site,
                         gradient_damping=True, damping_threshold=0.0, damping_strength=5.0)
            out["variants"]["scattered_knots_learnable"] = {
                "window_cell_lists": timed(B, 30, sc_kw, dict(domain_penalty_weight=0.01)),
                "materialised": timed(B, 15, sc_kw, dict(domain_penalty_weight=0.01, force_dense=True))}
            # A10: dense-grid inference (forward only, eval mode) on the
            # resident observations, as the dense-grid prediction callers run it
            from stnf.engine import Predictor
            model.eval()
            pr = Predictor(model)                               # default chunk (262 144 rows), eager launches
            n_inf = n_obs
            ci, ti = coords[:n_inf].contiguous(), t[:n_inf].contiguous()
            for _ in range(3):
                pr.predict(ci, ti)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                pr.predict(ci, ti)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 20
            out["variants"]["inference_forward_only"] = {"obs_per_s": n_inf / dt, "ms_per_call": dt * 1e3,
                                                         "rows_per_call": n_inf, "path": "window"}
            # the same forward on a site x time grid (what the reference's dense-grid callers loop over: the same
            # sites at every time): the per-site half of layer 0 is evaluated once per site
            S_g, T_g = min(n_obs, 100000), 20
            tv = torch.arange(T_g, device=dev, dtype=torch.float32) / (T_g - 1)
            cg = coords[:S_g].contiguous()
            for _ in range(2):
                pr.predict_grid(cg, tv)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                pr.predict_grid(cg, tv)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 5
            out["variants"]["inference_site_x_time_grid"] = {"obs_per_s": S_g * T_g / dt, "ms_per_call": dt * 1e3,
                                                             "sites": S_g, "times": T_g, "path": "window"}
            # BASELINE config C5: 10 M-point dense prediction grid (100 000 sites x 100 times), forward only, one GPU:
            # the site x time decomposition, and row by row through the chunked Predictor, eager and replayed from
            # a hipGraph per chunk
            S5, T5 = 100_000, 100
            g5 = torch.Generator().manual_seed(5)
            c5 = torch.rand(S5, 2, generator=g5).to(dev)
            tv5 = (torch.arange(T5, device=dev, dtype=torch.float32) / (T5 - 1))
            c5_res = {"points": S5 * T5, "sites": S5, "times": T5}
            pr.predict_grid(c5, tv5)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                yg5 = pr.predict_grid(c5, tv5)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 3
            c5_res["site_x_time_grid"] = {"points_per_s": S5 * T5 / dt, "ms_per_call": dt * 1e3}
            cc5, tt5 = c5.repeat(T5, 1), tv5.repeat_interleave(S5)
            for label, prd in (("rows_eager", pr), ("rows_hipgraph", Predictor(model, use_graph=True))):
                yr5 = prd.predict(cc5, tt5)
                yr5 = prd.predict(cc5, tt5)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    yr5 = prd.predict(cc5, tt5)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / 2
                c5_res[label] = {"points_per_s": S5 * T5 / dt, "ms_per_call": dt * 1e3,
                                 "max_abs_diff_vs_grid": float((yr5.view(T5, S5, -1) - yg5).abs().max())}
            del cc5, tt5, yr5, yg5
            out["variants"]["inference_c5_10M"] = c5_res
            model.train()
        if args.gpus == 1 and not args.no_cpu_baseline:
            # the reference's CPU path at all host cores this process may use AND at 8 threads
            out["cpu_baseline"] = cpu_baseline(wl, B, args.dropout)
            if out["cpu_baseline"]["cores"] != 8:
                out["cpu_baseline_8_threads"] = cpu_baseline(wl, B, args.dropout, cores=8)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
