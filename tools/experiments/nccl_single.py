"""RCCL bring-up of the N>1 code paths on a 1-GPU box: a world_size-1 "nccl" process group (the collectives degenerate to copies, but
they run through RCCL's streams, work handles and device-pointer checks): all-gather of a rendered tile (sync and async), the in-place
all-reduce of the field's gradient buffer inside the data-parallel fused train step, broadcast_parameters."""
import datetime
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
torch.set_num_threads(16)  # as bench.main does (train_bench halves it for its loop): with one OpenMP thread per host core the process outruns its cgroup CPU quota and is
                            # throttled until the next 100-ms period (stalls of ~90 ms every few iterations)
from text2nerf_amd import generate_rays  # noqa: E402
from text2nerf_amd.parallel import all_gather_tiles, broadcast_parameters, render_sharded  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.zeros(1, device=dev).add_(1)
torch.cuda.synchronize()
t0 = time.time()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
print("process group up in %.1f s" % (time.time() - t0), flush=True)
field, params, aabb = bench.build_field(dev, scene="S1-soft", seed=0)
field.frame_width = 400
rays = generate_rays(400, 400, [400.0, 400.0, 200, 200], torch.eye(4).numpy(), device=dev)
with torch.no_grad():
    rgb, depth, _, _ = field(rays, white_bg=True, is_train=False, N_samples=-1)
    tile = torch.cat([rgb, depth[:, None]], 1)
    g = all_gather_tiles(tile)
    assert torch.equal(g, tile)
    out = torch.empty_like(tile)
    w = dist.all_gather_into_tensor(out, tile, async_op=True)
    rgb2, _, _, _ = field(rays, white_bg=True, is_train=False, N_samples=-1)
    w.wait()
    assert torch.equal(out, tile) and torch.equal(rgb2, rgb)
    s_rgb, s_depth = render_sharded(rays, lambda r: field(r, white_bg=True, is_train=False, N_samples=-1)[:2])
    assert torch.equal(s_rgb, rgb) and torch.equal(s_depth, depth)
    # round 3: interleaved 8-row bands (frame_width given) and the rank evidence bench.py gathers
    s_rgb, s_depth = render_sharded(rays, lambda r: field(r, white_bg=True, is_train=False, N_samples=-1)[:2], frame_width=400)
    assert torch.equal(s_rgb, rgb) and torch.equal(s_depth, depth)
ids = [None]
dist.all_gather_object(ids, dict(bench.device_identity(dev), rank=0))
print("rank evidence:", ids, flush=True)
broadcast_parameters(field.parameters())
dist.barrier()
torch.cuda.synchronize()
print("all-gather (sync, async, sharded render) and broadcast ok", flush=True)
r = bench.train_bench(dev, iters=5, warmup=2, fused_step=True, dist=dist)
print({k: v for k, v in r.items() if "iters_per_s" in k or "ms_per_iter" in k}, flush=True)
for iters in (5, 30):   # (the first run of this path pays one-time costs: 5 iterations show them, 30 the steady state)
    r = bench.train_bench(dev, iters=iters, warmup=2, fused_optim=True, dist=dist)
    print("autograd + TVAdam(field), data-parallel:", iters, {k: v for k, v in r.items() if "iters_per_s" in k or "ms_per_iter" in k}, flush=True)
r = bench.train_bench(dev, iters=30, warmup=2, dist=dist)
print("autograd + torch Adam, data-parallel:", {k: v for k, v in r.items() if "iters_per_s" in k or "ms_per_iter" in k}, flush=True)
dist.barrier()
dist.destroy_process_group()
print("nccl single-rank path ok")
