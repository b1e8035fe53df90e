"""Weak-mode step pattern of bench.py (render a frame, all-gather its [rays, 4] tile asynchronously, wait for the previous gather) on a
single-rank RCCL group, with per-step end times: are there host-side stalls around the collectives?"""
import datetime, os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from text2nerf_amd import generate_rays  # noqa: E402
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.zeros(1, device=dev).add_(1); torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "nccl"
if mode != "none":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
field, params, aabb = bench.build_field(dev, scene="S1-soft", seed=0)
field.frame_width = 800
rays = generate_rays(800, 800, [800.0, 800.0, 400, 400], torch.eye(4).numpy(), device=dev)
pending = []
def step():
    with torch.no_grad():
        rgb, depth, z, w = field(rays, white_bg=True, is_train=False, N_samples=-1)
        if mode == "none":
            return
        tile = torch.cat([rgb, depth[:, None]], 1)
        out = torch.empty((tile.shape[0], 4), dtype=tile.dtype, device=tile.device)
        if mode == "sync":
            dist.all_gather_into_tensor(out, tile)
            return
        work = dist.all_gather_into_tensor(out, tile, async_op=True)
        pending.append((work, out, tile))
        if len(pending) > 1:
            pending.pop(0)[0].wait()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); ends = []
for _ in range(40):
    step(); ends.append(round((time.perf_counter() - t0) * 1e3, 1))
while pending: pending.pop(0)[0].wait()
torch.cuda.synchronize()
print(mode, "total %.1f ms for 40 steps; host-side step end times:" % ((time.perf_counter() - t0) * 1e3), ends, flush=True)
if mode != "none": dist.destroy_process_group()
