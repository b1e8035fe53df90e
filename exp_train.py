import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
dev = torch.device("cuda:0")
t0=time.time(); r = bench.train_bench(dev, iters=10, warmup=3); print(r, time.time()-t0)
# phase timing of one iteration
from text2nerf_amd import OctreeRender_trilinear_fast, synth
from text2nerf_amd.losses import TVLoss, TransMittanceLoss_mask
field, params, aabb = bench.build_field(dev)
rays = torch.from_numpy(synth.frame_rays_np(128,128)).contiguous()
opt = torch.optim.Adam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
tv = TVLoss()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for k in range(4):
    t=[sync()]
    rgb,_,depth,w,z = OctreeRender_trilinear_fast(rays, field, chunk=16384, N_samples=259, is_train=True, device=dev); t.append(sync())
    loss = rgb.mean() + depth.mean()*0.01 + (w*w).mean(); t.append(sync())
    loss = loss + field.TV_loss_density(tv)*0.1 + field.TV_loss_app(tv)*0.01; t.append(sync())
    opt.zero_grad(); loss.backward(); t.append(sync())
    opt.step(); t.append(sync())
    print("fwd %.1f loss %.1f tv %.1f bwd %.1f adam %.1f ms" % tuple((t[i+1]-t[i])*1e3 for i in range(5)))
