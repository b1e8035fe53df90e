"""Wall time of the phases of the C3-shaped train step with a device sync after each (fused TV+Adam path)."""
import sys, time, os, gc
gc.disable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
torch.set_num_threads(max(1, min(bench.HOST_CORES, 16)))
from text2nerf_amd import OctreeRender_trilinear_fast, synth
from text2nerf_amd.losses import TVLoss, TransMittanceLoss_mask
from text2nerf_amd.optim import TVAdam
dev = torch.device("cuda:0")
field, params, aabb = bench.build_field(dev)
poses = synth.local_fixed_like_poses(9)
allrays = torch.from_numpy(np.concatenate([synth.frame_rays_np(512, 512, c2w=p) for p in poses]))
opt = TVAdam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
tv, tl = TVLoss(), TransMittanceLoss_mask(dev)
np.random.seed(1024); torch.manual_seed(1024)
perm = torch.from_numpy(np.random.permutation(allrays.shape[0]))
with torch.no_grad():
    sub = allrays[::4].to(dev); rgb_s, dep_s, _, _ = field(sub, white_bg=True, is_train=False, N_samples=259)
allrgb = rgb_s.cpu().repeat_interleave(4, 0)[:allrays.shape[0]]; alldepth = dep_s.cpu().repeat_interleave(4, 0)[:allrays.shape[0]]
def sync(): torch.cuda.synchronize(); return time.perf_counter()
acc = np.zeros(6); n = 0
for k in range(30):
    t = [sync()]
    idx = perm[k * 16384:(k + 1) * 16384]
    rays, rgb_t, dep_t = allrays[idx], allrgb[idx].to(dev), alldepth[idx].to(dev); t.append(sync())
    rgb, _, depth, w, z = OctreeRender_trilinear_fast(rays, field, chunk=16384, N_samples=259, is_train=True, device=dev); t.append(sync())
    loss = torch.mean((rgb - rgb_t) ** 2) + 0.005 * torch.mean((depth - dep_t) ** 2) + 1e3 * tl(w, (z - dep_t[:, None] + 0.1) < 0); t.append(sync())
    opt.zero_grad(); t.append(sync())
    loss.backward(); t.append(sync())
    opt.step(tv=[(field.density_plane, 0.1), (field.app_plane, 0.01)]); t.append(sync())
    if k >= 5:
        acc += np.diff(t) * 1e3; n += 1
print("data %.2f fwd %.2f loss %.2f zero %.2f bwd %.2f opt %.2f  total %.2f ms" % (tuple(acc / n) + (acc.sum() / n,)))
