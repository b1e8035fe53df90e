import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.conftest import TINY
from tests.test_hip_parity import make_field, dev
from text2nerf_amd import synth
params = synth.make_field_params(11, TINY["grid"], density_scale=0.9, aabb=TINY["aabb"])
f = make_field(params, TINY["grid"], TINY["aabb"], TINY["near_far"])
tiny = dict(np.load("tests/golden/tiny.npz"))
rays = torch.from_numpy(tiny["tiny_rays"]).to(dev())
outs = []
for i in range(6):
    with torch.no_grad():
        r = f(rays, white_bg=True, is_train=False, N_samples=70 if i % 2 else -1)
    outs.append((r[0].clone(), r[1].clone(), f.stats()))
for i in range(2, 6):
    print(i, torch.equal(outs[i][0], outs[i - 2][0]), float((outs[i][0] - outs[i - 2][0]).abs().max()), outs[i][2])
