"""Host-side cost (enqueue only) of the per-step collectives of the data-parallel step, on a one-rank RCCL group:
how much of a 0.11 ms step the host needs just to ISSUE reduce-scatter + all-reduce(2 KB) + all-gather."""
import os, time
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
n = 2756100 // 8 * 8
flat = torch.zeros(n, device=dev); grad = torch.zeros(n, device=dev); small = torch.zeros(512, device=dev)
mine = grad[:n]
for _ in range(20):
    dist.reduce_scatter_tensor(mine, grad); dist.all_reduce(small); dist.all_gather_into_tensor(flat, flat[:n])
torch.cuda.synchronize()
k = 2000
for name, fn in (("reduce_scatter_tensor", lambda: dist.reduce_scatter_tensor(mine, grad)),
                 ("all_reduce(512 floats)", lambda: dist.all_reduce(small)),
                 ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(flat, flat[:n])),
                 ("all_reduce(flat)", lambda: dist.all_reduce(grad))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    t1 = time.perf_counter() - t0
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    print(f"{name:26s} host enqueue {t1 / k * 1e6:6.1f} us/call, wall {t2 / k * 1e6:6.1f} us/call", flush=True)
dist.destroy_process_group()
