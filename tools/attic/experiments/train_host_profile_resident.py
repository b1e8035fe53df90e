"""Host side of the fused train step with the training set resident in HBM: host loop time vs drained time, cProfile by own time."""
import cProfile, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
torch.set_num_threads(4)
dev = torch.device("cuda", 0)
from text2nerf_amd import synth  # noqa: E402
from text2nerf_amd.optim import TVAdam  # noqa: E402
field, params, aabb = bench.build_field(dev)
poses = bench.reference_poses("local_fixed")
allrays = torch.from_numpy(np.concatenate([synth.frame_rays_np(512, 512, c2w=p) for p in poses]))
g = np.random.Generator(np.random.PCG64(1024))
with torch.no_grad():
    sub = allrays[::4].to(dev)
    rgb_s, dep_s, _, _ = field(sub, white_bg=True, is_train=False, N_samples=259)
allrgb = (rgb_s.cpu().repeat_interleave(4, 0)[: allrays.shape[0]] + torch.from_numpy(g.normal(0, 0.05, (allrays.shape[0], 3)).astype(np.float32))).clamp(0, 1)
alldepth = dep_s.cpu().repeat_interleave(4, 0)[: allrays.shape[0]] + torch.from_numpy(g.normal(0, 0.05, (allrays.shape[0],)).astype(np.float32))
opt = TVAdam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=field)
perm = torch.from_numpy(np.random.permutation(allrays.shape[0])).to(dev)
R_, G_, D_ = allrays.to(dev), allrgb.to(dev), alldepth.to(dev)
tv = [(field.density_plane, 0.1), (field.app_plane, 0.01)]
B = 16384
def it(k):
    idx = perm[(k * B) % (perm.numel() - B):][:B]
    return field.train_step(R_[idx], G_[idx], D_[idx], opt, N_samples=259, white_bg=True, tv=tv)
for k in range(5): it(k)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(60): it(5 + k)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host loop %.3f ms/iter, with drain %.3f ms/iter" % ((t1 - t0) / 60 * 1e3, (t2 - t0) / 60 * 1e3))
# host-only cost: the same loop with the GPU never the limiter is not available; cProfile by own time instead
pr = cProfile.Profile(); pr.enable()
for k in range(60): it(70 + k)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
