"""Soak: N fused train steps of the bench's C3 loop (training set resident in HBM); loss, sample counts, finite parameters, device memory."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from text2nerf_amd import synth  # noqa: E402
from text2nerf_amd.optim import TVAdam  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
field, params, aabb = bench.build_field(dev)
poses = bench.reference_poses("local_fixed")
allrays = torch.from_numpy(np.concatenate([synth.frame_rays_np(512, 512, c2w=p) for p in poses]))
g = np.random.Generator(np.random.PCG64(1024))
with torch.no_grad():
    rgb_s, dep_s, _, _ = field(allrays[::4].to(dev), white_bg=True, is_train=False, N_samples=259)
allrgb = (rgb_s.cpu().repeat_interleave(4, 0)[: allrays.shape[0]] + torch.from_numpy(g.normal(0, 0.05, (allrays.shape[0], 3)).astype(np.float32))).clamp(0, 1)
alldepth = dep_s.cpu().repeat_interleave(4, 0)[: allrays.shape[0]] + torch.from_numpy(g.normal(0, 0.05, (allrays.shape[0],)).astype(np.float32))
R_, G_, D_ = allrays.to(dev), allrgb.to(dev), alldepth.to(dev)
opt = TVAdam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=field)
perm = torch.from_numpy(np.random.permutation(allrays.shape[0])).to(dev)
tv = [(field.density_plane, 0.1), (field.app_plane, 0.01)]
for k in range(N):
    idx = perm[(k * 16384) % (perm.numel() - 16384):][:16384]
    losses = field.train_step(R_[idx], G_[idx], D_[idx], opt, N_samples=259, white_bg=True, tv=tv)
    if k % 50 == 49 or k == N - 1:
        torch.cuda.synchronize()
        finite = all(bool(torch.isfinite(p).all()) for p in field.parameters())
        print(k + 1, "loss", [round(float(v), 5) for v in losses], field.stats(), "finite", finite,
              "mem GiB %.2f / reserved %.2f" % (torch.cuda.memory_allocated(dev) / 2**30, torch.cuda.memory_reserved(dev) / 2**30), flush=True)
        assert finite
