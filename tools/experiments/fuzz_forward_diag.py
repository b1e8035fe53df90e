"""One seed of tests/test_hip_fuzz.py::test_random_frames_on_the_tile_marcher_vs_c_oracle in detail: the rays whose colour differs from
oracle_c by more than the budget, their weights next to the 1e-4 appearance threshold, the same frame through the per-ray marcher, and
the float64 PyTorch oracle as the referee.   python tools/experiments/fuzz_forward_diag.py SEED   (T2N_LIB selects the library build)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_torch as O  # noqa: E402
from oracle.oracle_c import COracle  # noqa: E402
from tests import test_hip_fuzz as F  # noqa: E402
from text2nerf_amd import synth  # noqa: E402

seed = int(sys.argv[1])
dev = torch.device("cuda:0")
g = np.random.Generator(np.random.PCG64(5000 + seed))
grid = [int(g.integers(9, 200)) for _ in range(3)]
lo = (-g.uniform(2.0, 9.0, 3)).astype(np.float32)
hi = g.uniform(2.0, 9.0, 3).astype(np.float32)
aabb = [lo.tolist(), hi.tolist()]
near_far = [float(g.uniform(0.05, 1.0)), float(g.uniform(4.0, 12.0))]
step_ratio = float(g.choice([0.5, 1.0]))
params = synth.make_field_params(6000 + seed, grid, density_scale=float(g.uniform(0.3, 1.6)), aabb=aabb)
f = F._field(params, grid, aabb, near_far, step_ratio)
cfg = O.FieldConfig(aabb=aabb, grid_size=grid, near_far=near_far, step_ratio=step_ratio)
co = COracle(cfg, params)
centre = tuple(float(v) for v in (lo + (hi - lo) * g.uniform(0.2, 0.8, 3)))
far_out = tuple(float(v) for v in (hi + g.uniform(0.5, 3.0, 3)))
print("seed", seed, "grid", grid, "step_ratio", step_ratio, "lib", os.environ.get("T2N_LIB", "main"))
P64 = O.params_from_numpy(params, dtype=torch.float64)
for cam in (centre, far_out):
    H, W = int(g.integers(8, 70)), int(g.integers(8, 90))
    pose = synth.look_pose(float(g.uniform(-3, 3)), float(g.uniform(-1, 1)), cam)
    if seed & 1:
        pose[:3, :3] = pose[:3, :3] @ np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], np.float32)
    rays = synth.frame_rays_np(H, W, c2w=pose)
    n = int(g.integers(17, 150)) if seed % 3 else -1
    N = n if n > 0 else cfg.n_samples
    c_rgb, c_depth, c_z, c_w = co.render(rays, n_samples=N, is_train=False, white_bg=True)
    rt = torch.from_numpy(rays).to(dev)
    out = {}
    for name, fw in (("tile", W), ("per-ray", 0)):
        f.frame_width = fw
        f.materialize_weights = True
        with torch.no_grad():
            rgb, depth, z, w = f(rt, is_train=False, white_bg=True, N_samples=n)
        out[name] = (rgb.cpu().numpy(), w.cpu().numpy())
    e_t = np.abs(out["tile"][0] - c_rgb).max(1)
    e_p = np.abs(out["per-ray"][0] - c_rgb).max(1)
    print(f"camera {cam}: {H}x{W}, N {N}: max |rgb - oracle_c| tile {e_t.max():.3e} per-ray {e_p.max():.3e}; rays over 1e-4: tile {(e_t > 1e-4).sum()} per-ray {(e_p > 1e-4).sum()}")
    bad = np.nonzero(e_t > 5e-5)[0]
    if len(bad):
        r64 = torch.from_numpy(rays[bad]).double()
        rgb64, _, _, w64 = O.forward(cfg, P64, r64, white_bg=True, is_train=False, n_samples=N)
        for k, r in enumerate(bad[:6]):
            wt, wc, wd = out["tile"][1][r], c_w[r], w64[k].numpy()
            near = np.nonzero((np.abs(wd - 1e-4) < 3e-7))[0]
            print(f"  ray {r}: err tile {e_t[r]:.3e} per-ray {e_p[r]:.3e}; |tile - f64| {np.abs(out['tile'][0][r] - rgb64[k].numpy()).max():.3e}  |oracle_c - f64| {np.abs(c_rgb[r] - rgb64[k].numpy()).max():.3e}"
                  f"  samples within 3e-7 of the threshold (f64): {[(int(i), float(wd[i]), float(wt[i]), float(wc[i])) for i in near]}  app counts tile/c/f64 {(wt > 1e-4).sum()} {(wc > 1e-4).sum()} {(wd > 1e-4).sum()}")
