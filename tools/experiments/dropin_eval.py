"""The reference's evaluation call (all rays of one 800x800 image from host memory, no hints, 5-tuple returned) through the mirror:
per-frame time with and without the automatic raster detection."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from text2nerf_amd import OctreeRender_trilinear_fast, generate_rays  # noqa: E402
torch.set_num_threads(8)
dev = torch.device("cuda", 0)
field, params, aabb = bench.build_field(dev, scene="S1-soft", seed=0)
rays = generate_rays(800, 800, [800.0, 800.0, 400, 400], torch.eye(4).numpy(), device=dev).cpu()
for auto in (True, False, True):
    field.auto_frame_width = auto
    for where, r in (("host rays", rays), ("device rays", rays.to(dev))):
        with torch.no_grad():
            for _ in range(3): OctreeRender_trilinear_fast(r, field, chunk=4096, N_samples=-1, white_bg=True, is_train=False, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): out = OctreeRender_trilinear_fast(r, field, chunk=4096, N_samples=-1, white_bg=True, is_train=False, device=dev)
            torch.cuda.synchronize()
        print(f"auto_frame_width={auto!s:5s} {where:12s}: {(time.perf_counter() - t0) / 20 * 1e3:7.2f} ms per evaluation call", flush=True)
