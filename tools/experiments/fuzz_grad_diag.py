"""Why does a fuzz gradient case disagree? Discrete events near their thresholds (appearance list: w vs 1e-4; head ReLUs; rgb clamp)
and forward differences, for given seeds of test_random_configuration_gradients_vs_oracle_autograd."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_torch as O
from tests.test_hip_fuzz import _field
from tests.test_hip_parity import dev
from text2nerf_amd import synth
for seed in [int(x) for x in sys.argv[1:]]:
    g = np.random.Generator(np.random.PCG64(5000 + seed))
    grid = [int(g.integers(9, 40)) for _ in range(3)]
    lo = (-g.uniform(2.0, 9.0, 3)).astype(np.float32); hi = g.uniform(2.0, 9.0, 3).astype(np.float32)
    aabb = [lo.tolist(), hi.tolist()]
    near_far = [float(g.uniform(0.05, 1.0)), float(g.uniform(4.0, 12.0))]
    params = synth.make_field_params(6000 + seed, grid, density_scale=float(g.uniform(0.5, 1.4)), aabb=aabb)
    f = _field(params, grid, aabb, near_far, 1.0)
    cfg = O.FieldConfig(aabb=aabb, grid_size=grid, near_far=near_far)
    centre = tuple(float(v) for v in (lo + (hi - lo) * g.uniform(0.3, 0.7, 3)))
    rays = torch.from_numpy(synth.frame_rays_np(12, 14, c2w=synth.look_pose(float(g.uniform(-3, 3)), float(g.uniform(-1, 1)), centre)))
    is_train = bool(seed % 2 == 0)
    n = int(g.integers(20, 80))
    ca = torch.from_numpy(g.uniform(-1, 1, (rays.shape[0], 3)).astype(np.float32))
    torch.manual_seed(seed); jit = torch.rand(rays.shape[0], 1) if is_train else None
    torch.manual_seed(seed)
    out = f(rays, is_train=is_train, white_bg=True, N_samples=n)
    P = O.params_from_numpy(params, requires_grad=True)
    o = O.forward(cfg, P, rays, white_bg=True, is_train=is_train, n_samples=n, jitter=jit)
    w_h, w_o = out[3].detach().cpu(), o[3].detach()
    print(f"seed {seed}: grid {grid} n {n} train {is_train}; fwd diffs rgb {float((out[0].detach().cpu()-o[0].detach()).abs().max()):.2e} "
          f"depth {float((out[1].detach().cpu()-o[1].detach()).abs().max()):.2e} w {float((w_h-w_o).abs().max()):.2e}")
    mh, mo = w_h > 1e-4, w_o > 1e-4
    print("   appearance masks differ at", int((mh != mo).sum()), "samples; closest |w - 1e-4| =", float((w_o - 1e-4).abs().min()))
    raw = o[0].detach()
    print("   rgb_map range (oracle):", float(raw.min()), float(raw.max()), " closest to the clamp bounds:", float(torch.minimum(raw.abs(), (raw - 1).abs()).min()))
    # saturated alphas: weights pattern
    T_after = 1 - w_o.sum(-1)
    print("   rays with acc > 0.999:", int((T_after < 1e-3).sum()), "of", rays.shape[0], " max weight", float(w_o.max()))
