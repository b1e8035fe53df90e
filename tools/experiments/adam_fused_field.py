"""foreach vs fused torch Adam on the gradients of an actual render backward of the tiny field (test_f1's scenario): where do they differ?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.conftest import TINY
from tests.test_hip_parity import make_field, dev
from text2nerf_amd import synth
params = synth.make_field_params(3, TINY["grid"], aabb=TINY["aabb"])
rays = torch.from_numpy(synth.frame_rays_np(16, 16, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0))))
def grads():
    f = make_field(params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    torch.manual_seed(1)
    rgb, depth, z, w = f(rays, is_train=True, white_bg=True, N_samples=40)
    (rgb.sum() + 0.1 * depth.sum()).backward()
    return f
fa, fb = grads(), grads()
fa.optimizer_hints = False
oa = torch.optim.Adam(fa.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
ob = torch.optim.Adam(fb.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
print("fused flags:", [g.get("fused") for g in oa.param_groups], [g.get("fused") for g in ob.param_groups])
before = {k: v.detach().clone() for k, v in fa.state_dict().items()}
oa.step(); ob.step()
for (k, a), (_, b) in zip(fa.named_parameters(), fb.named_parameters()):
    g = a.grad
    d = (a.detach() - b.detach()).abs()
    big = d > 1e-6
    print(f"{k:28s} max diff {float(d.max()):.3e}  n(diff>1e-6) {int(big.sum()):6d}  |g| of those: min {float(g[big].abs().min()) if big.any() else 0:.3e} max {float(g[big].abs().max()) if big.any() else 0:.3e}  moved(foreach) {float((a.detach()-before[k]).abs().max()):.3e}")
