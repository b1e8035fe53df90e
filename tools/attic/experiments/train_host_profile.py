"""cProfile of the fused train step's host side (what the Python loop costs per iteration when the GPU is not the limit)."""
import cProfile, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
dev = torch.device("cuda", 0)
from text2nerf_amd import synth  # noqa: E402
from text2nerf_amd.optim import TVAdam  # noqa: E402
field, params, aabb = bench.build_field(dev)
n_samples = 259
poses = bench.reference_poses("local_fixed")
allrays = torch.from_numpy(np.concatenate([synth.frame_rays_np(512, 512, c2w=p) for p in poses]))
allrgb = torch.rand(allrays.shape[0], 3)
alldepth = torch.rand(allrays.shape[0]) * 3 + 1
opt = TVAdam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=field)
perm = torch.from_numpy(np.random.permutation(allrays.shape[0]))
tv_terms = [(field.density_plane, 0.1), (field.app_plane, 0.01)]
def it(k):
    idx = perm[(k * 16384) % (perm.numel() - 16384):][:16384]
    return field.train_step(allrays[idx], allrgb[idx], alldepth[idx], opt, N_samples=n_samples, white_bg=True, tv=tv_terms)[3]
for k in range(5): it(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(50): it(5 + k)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host loop %.3f ms/iter, with drain %.3f ms/iter" % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
pr = cProfile.Profile(); pr.enable()
for k in range(50): it(60 + k)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
