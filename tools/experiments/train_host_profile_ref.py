"""Host side of the reference-form train step (autograd + the model's TV terms + torch.optim.Adam): host loop time vs drained time, and a
cProfile of where the Python time goes."""
import cProfile, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
torch.set_num_threads(8)
dev = torch.device("cuda", 0)
from text2nerf_amd import OctreeRender_trilinear_fast, synth, to_device_async  # noqa: E402
from text2nerf_amd.losses import TVLoss, TransMittanceLoss_mask  # noqa: E402
field, params, aabb = bench.build_field(dev)
fused_hint = len(sys.argv) > 1 and sys.argv[1] == "fused"
n_samples = 259
poses = bench.reference_poses("local_fixed")
allrays = torch.from_numpy(np.concatenate([synth.frame_rays_np(512, 512, c2w=p) for p in poses]))
with torch.no_grad():
    sub = allrays[::4].to(dev)
    rgb_s, dep_s, _, _ = field(sub, white_bg=True, is_train=False, N_samples=n_samples)
allrgb = rgb_s.cpu().repeat_interleave(4, 0)[: allrays.shape[0]].clamp(0, 1)
alldepth = dep_s.cpu().repeat_interleave(4, 0)[: allrays.shape[0]]
opt = torch.optim.Adam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), fused=True if fused_hint else None)
tv, tl = TVLoss(), TransMittanceLoss_mask(dev)
perm = torch.from_numpy(np.random.permutation(allrays.shape[0]))
def it(k):
    idx = perm[(k * 16384) % (perm.numel() - 16384):][:16384]
    rays, rgb_t, dep_t = allrays[idx], to_device_async(allrgb[idx], dev), to_device_async(alldepth[idx], dev)
    rgb, _, depth, w, z = OctreeRender_trilinear_fast(rays, field, chunk=16384, N_samples=n_samples, white_bg=True, ndc_ray=False, device=dev, is_train=True)
    loss = torch.mean((rgb - rgb_t) ** 2) + 0.005 * torch.mean((depth - dep_t) ** 2)
    loss = loss + 1e3 * tl(w, (z - dep_t[:, None] + 0.1) < 0)
    loss = loss + field.TV_loss_density(tv) * 0.1 + field.TV_loss_app(tv) * 0.01
    opt.zero_grad(); loss.backward(); opt.step()
    return loss
for k in range(5): it(k)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(20): it(5 + k)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host loop %.3f ms/iter, with drain %.3f ms/iter (torch fused Adam: %s)" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3, fused_hint))
pr = cProfile.Profile(); pr.enable()
for k in range(20): it(30 + k)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
