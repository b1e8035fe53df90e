"""Child process of tests/test_memory.py: device memory torch has RESERVED (what the process takes from the GPU, not what is in use)
after (a) three C2 frames (300^3, 800x800, budgeted lists) and (b) 300 fused train steps of the bench's C3 loop on noisy targets —
the run in which the appearance list triples around step 60-100 and shrinks again (profiles/round4_train_soak.txt). One JSON line."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from text2nerf_amd import synth  # noqa: E402
from text2nerf_amd.optim import TVAdam  # noqa: E402

GiB = float(1 << 30)
dev = torch.device("cuda:0")
out = {}
field, params, aabb = bench.build_field(dev)
field.materialize_weights = False
field.frame_width = 800
rays = torch.from_numpy(synth.frame_rays_np(800, 800)).to(dev)
with torch.no_grad():
    for _ in range(3):
        rgb, depth, _, _ = field(rays)
torch.cuda.synchronize()
out["frame_reserved_GiB"] = torch.cuda.memory_reserved(dev) / GiB
out["frame_allocated_GiB"] = torch.cuda.memory_allocated(dev) / GiB
out["frame_list_retry"] = field.stats()["list_retry"]
del rays, rgb, depth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
field.frame_width = 0
poses = bench.reference_poses("local_fixed")
allrays = torch.from_numpy(np.concatenate([synth.frame_rays_np(512, 512, c2w=p) for p in poses]))
g = np.random.Generator(np.random.PCG64(1024))
with torch.no_grad():
    rgb_s, dep_s, _, _ = field(allrays[::4].to(dev), white_bg=True, is_train=False, N_samples=259)
allrgb = (rgb_s.cpu().repeat_interleave(4, 0)[: allrays.shape[0]] + torch.from_numpy(g.normal(0, 0.05, (allrays.shape[0], 3)).astype(np.float32))).clamp(0, 1)
alldepth = dep_s.cpu().repeat_interleave(4, 0)[: allrays.shape[0]] + torch.from_numpy(g.normal(0, 0.05, (allrays.shape[0],)).astype(np.float32))
R_, G_, D_ = allrays.to(dev), allrgb.to(dev), alldepth.to(dev)
opt = TVAdam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=field)
np.random.seed(1024)
perm = torch.from_numpy(np.random.permutation(allrays.shape[0])).to(dev)
tv = [(field.density_plane, 0.1), (field.app_plane, 0.01)]
peak_app, peak_res = 0, 0.0
for k in range(steps):
    idx = perm[(k * 16384) % (perm.numel() - 16384):][:16384]
    losses = field.train_step(R_[idx], G_[idx], D_[idx], opt, N_samples=259, white_bg=True, tv=tv)
    if k % 25 == 24 or k == steps - 1:
        torch.cuda.synchronize()
        peak_app = max(peak_app, field.stats()["appearance"])
        peak_res = max(peak_res, torch.cuda.memory_reserved(dev) / GiB)
out["train_steps"] = steps
out["train_peak_appearance_samples"] = peak_app
out["train_reserved_GiB_peak"] = peak_res
out["train_reserved_GiB_end"] = torch.cuda.memory_reserved(dev) / GiB
out["train_allocated_GiB_end"] = torch.cuda.memory_allocated(dev) / GiB
out["finite"] = all(bool(torch.isfinite(p).all()) for p in field.parameters())
print(json.dumps(out), flush=True)
