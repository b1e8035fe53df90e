"""Host-side time of the fused train step at a small batch (GPU work is short: the loop time is the host's): cProfile by tottime."""
import cProfile, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
torch.set_num_threads(8)
dev = torch.device("cuda", 0)
from text2nerf_amd import synth  # noqa: E402
from text2nerf_amd.optim import TVAdam  # noqa: E402
field, params, aabb = bench.build_field(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rays = torch.from_numpy(synth.frame_rays_np(512, 512)[:: 512 * 512 // B][:B].copy())
rgb_t = torch.rand(B, 3); dep_t = torch.rand(B) * 5 + 1
opt = TVAdam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=field)
tv = [(field.density_plane, 0.1), (field.app_plane, 0.01)]
SPEC = len(sys.argv) > 2 and sys.argv[2] == "spec"
def it():
    return field.train_step(rays, rgb_t, dep_t, opt, N_samples=259, white_bg=True, tv=tv, speculative=SPEC)
for k in range(10): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(100): it()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("speculative", SPEC); print("batch %d: host loop %.3f ms/iter, with drain %.3f ms/iter" % (B, (t1 - t0) / 100 * 1e3, (t2 - t0) / 100 * 1e3))
pr = cProfile.Profile(); pr.enable()
for k in range(100): it()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
