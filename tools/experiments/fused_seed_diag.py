"""One seed of the fused-vs-composed campaign: where do the parameters differ after 1 / 2 / 3 steps, and what do the GRADIENTS of step 1 look like
at those elements (fused phases = 1 buffer against the composed backward's)? python tools/experiments/fused_seed_diag.py SEED"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from text2nerf_amd import synth
from text2nerf_amd.optim import TVAdam
from tests.test_hip_parity import make_field, dev
seed = int(sys.argv[1])
g = np.random.Generator(np.random.PCG64(9000 + seed))
grid = [int(g.integers(9, 60)) for _ in range(3)]
lo = (-g.uniform(2.0, 9.0, 3)).astype(np.float32); hi = g.uniform(2.0, 9.0, 3).astype(np.float32)
aabb = [lo.tolist(), hi.tolist()]
near_far = [float(g.uniform(0.05, 1.0)), float(g.uniform(6.0, 14.0))]
params = synth.make_field_params(9100 + seed, grid, density_scale=float(g.uniform(0.5, 1.4)), aabb=aabb)
centre = tuple(float(v) for v in (lo + (hi - lo) * g.uniform(0.3, 0.7, 3)))
H, W = int(g.integers(9, 40)), int(g.integers(9, 40))
rays = torch.from_numpy(synth.frame_rays_np(H, W, c2w=synth.look_pose(float(g.uniform(-3, 3)), float(g.uniform(-1, 1)), centre)))
n = int(g.integers(5, rays.shape[0]))
rays = rays[torch.from_numpy(g.permutation(rays.shape[0])[:n])].contiguous()
rgb_t = torch.from_numpy(g.uniform(0, 1, (n, 3)).astype(np.float32)); dep_t = torch.from_numpy(g.uniform(1, 9, (n,)).astype(np.float32))
N = int(g.integers(16, 120))
print("grid", grid, "rays", n, "N", N)
def run(fused, steps):
    f = make_field(params, grid, aabb, near_far)
    o = TVAdam(f.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=f)
    for it in range(steps):
        torch.manual_seed(300 + it)
        f.train_step(rays, rgb_t, dep_t, o, N_samples=N, white_bg=True, tv=[(f.density_plane, 0.1), (f.app_plane, 0.01)], fused=fused, graph=False)
    if fused:
        f.__dict__["_fused_step"].sync()
    torch.cuda.synchronize()
    return f, o
for steps in (1, 2, 3):
    fa, oa = run(False, steps); fb, ob = run(True, steps)
    worst = []
    for (k, a), (_, b) in zip(fa.state_dict().items(), fb.state_dict().items()):
        d = (a - b).abs()
        worst.append((float(d.max()), k, int(d.argmax())))
    worst.sort(reverse=True)
    print("steps", steps, "worst:", [(round(w[0], 6), w[1], w[2]) for w in worst[:3]])
    if steps == 1:
        k, idx = worst[0][1], worst[0][2]
        # Adam state of that element in both runs (step 1: m = 0.1 g, v = 0.01 g^2 -> g = 10 m)
        pa = dict(fa.named_parameters())[k]; pb = dict(fb.named_parameters())[k]
        for tag, o, p in (("composed", oa, pa), ("fused", ob, pb)):
            st = o.state[p]
            print("   ", tag, "keys", [x for x in st.keys()])
