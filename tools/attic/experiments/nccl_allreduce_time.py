"""How long do the collectives of the data-parallel step take on a single-rank RCCL group (no wire: pure launch / bookkeeping cost)?"""
import datetime, os, time
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.zeros(1, device=dev).add_(1); torch.cuda.synchronize()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
def timeit(name, fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:50s} host {1e3 * (t1 - t0) / n:8.3f} ms/call, with drain {1e3 * (t2 - t0) / n:8.3f} ms/call", flush=True)
big = torch.zeros(17_400_000, device=dev)
small = torch.zeros(62_000, device=dev)
timeit("all_reduce 69.6 MB in place", lambda: dist.all_reduce(big))
timeit("all_reduce 248 KB in place", lambda: dist.all_reduce(small))
timeit("div_ 69.6 MB", lambda: big.div_(1.0))
parts = [torch.zeros(n, device=dev) for n in (27 * 144, 128 * 351, 128, 128 * 128, 128, 3 * 128, 3)]
timeit("cat of 7 head tensors", lambda: torch.cat([p.reshape(-1) for p in parts]))
def head():
    flat = torch.cat([p.reshape(-1) for p in parts]); dist.all_reduce(flat); flat.div_(1.0)
    off = 0
    for p in parts:
        n = p.numel(); p.copy_(flat[off:off + n].view_as(p)); off += n
timeit("head: cat + all_reduce + div + copy-back", head)
timeit("barrier", lambda: dist.barrier())
dist.destroy_process_group()
