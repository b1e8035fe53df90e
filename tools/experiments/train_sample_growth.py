"""Appearance / evaluated sample counts of the bench's C3 train loop every ten fused steps (the noisy targets turn the field into fog after ~45 steps)."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, bench, numpy as np
# monkeypatch: run the fused step loop and print appearance counts
dev = torch.device("cuda:0")
from text2nerf_amd import synth
from text2nerf_amd.optim import TVAdam
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
np.random.seed(1024); torch.manual_seed(1024)
perm = torch.from_numpy(np.random.permutation(allrays.shape[0]))
tv = [(field.density_plane, 0.1), (field.app_plane, 0.01)]
for k in range(70):
    idx = perm[(k * 16384) % (perm.numel() - 16384):][:16384]
    field.train_step(allrays[idx], allrgb[idx], alldepth[idx], opt, N_samples=259, white_bg=True, tv=tv)
    if k % 10 == 9: print(k + 1, field.stats())
