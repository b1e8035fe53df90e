"""Offline fuzz campaign: the three randomised differential tests of tests/test_hip_fuzz.py over many more seeds than the suite
runs (python tools/fuzz_campaign.py FIRST COUNT). Prints every failing (test, seed) with the first line of its assertion."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_hip_fuzz as F  # noqa: E402



def discrete_margins(seed):
    """For a failed gradient case: how close the case sits to a discrete event, from the oracle in float64 — the head's ReLU inputs on
    the appearance samples, the appearance-list threshold |w - 1e-4|, rays whose colour sits on the clamp bound. A gradient comparison
    across implementations is only meaningful when none of these is within rounding (~1e-6): an input on the other side of one flips
    that sample's (or ray's) whole upstream gradient, ~1e-3 of a tensor's largest gradient on these 168-ray batches."""
    import numpy as np
    import torch
    from oracle import oracle_torch as O
    from text2nerf_amd import synth
    g = np.random.Generator(np.random.PCG64(5000 + seed))
    grid = [int(g.integers(9, 40)) for _ in range(3)]
    lo = (-g.uniform(2.0, 9.0, 3)).astype(np.float32)
    hi = g.uniform(2.0, 9.0, 3).astype(np.float32)
    aabb = [lo.tolist(), hi.tolist()]
    near_far = [float(g.uniform(0.05, 1.0)), float(g.uniform(4.0, 12.0))]
    params = synth.make_field_params(6000 + seed, grid, density_scale=float(g.uniform(0.5, 1.4)), aabb=aabb)
    cfg = O.FieldConfig(aabb=aabb, grid_size=grid, near_far=near_far)
    centre = tuple(float(v) for v in (lo + (hi - lo) * g.uniform(0.3, 0.7, 3)))
    rays = torch.from_numpy(synth.frame_rays_np(12, 14, c2w=synth.look_pose(float(g.uniform(-3, 3)), float(g.uniform(-1, 1)), centre))).double()
    is_train = bool(seed % 2 == 0)
    n = int(g.integers(20, 80))
    torch.manual_seed(seed)
    jit = torch.rand(rays.shape[0], 1).double() if is_train else None
    P = O.params_from_numpy(params, dtype=torch.float64)
    rgb, depth, z, w, aux = O.forward(cfg, P, rays, white_bg=True, is_train=is_train, n_samples=n, jitter=jit, return_aux=True)
    pts, _, _ = O.sample_ray(cfg, rays[:, :3], rays[:, 3:6], n, jit)
    m = aux["app_mask"]
    # a ray sits ON the clamp bound only if it has opacity (a background ray is exactly 1.0 in every implementation, with zero gradient:
    # it explains nothing) and its colour before the clamp is within rounding of 0 or 1 (163 of 168 background rays made any
    # failing seed a "knife edge")
    acc = w.sum(-1)
    raw = aux.get("rgb_pre_clamp", rgb)
    on_bound = (((raw - 1.0).abs() < 1e-6) | (raw.abs() < 1e-6)).any(-1) & (acc > 1e-6)
    out = {"w_vs_threshold": float((w - 1e-4).abs().min()), "rays_on_clamp_bound": int(on_bound.sum()),
           "background_rays": int((acc <= 1e-6).sum())}
    if m.any():
        f = O.app_feature(P, O.normalize_coord(cfg, pts)[m])
        x = torch.cat([f, O.positional_encoding(f, 6)], -1)
        h0 = x @ P["renderModule.mlp.0.weight"].T + P["renderModule.mlp.0.bias"]
        h1 = torch.relu(h0) @ P["renderModule.mlp.2.weight"].T + P["renderModule.mlp.2.bias"]
        out.update(min_abs_h0_pre=float(h0.abs().min()), min_abs_h1_pre=float(h1.abs().min()))
    knife = out["w_vs_threshold"] < 2e-6 or out["rays_on_clamp_bound"] > 0 or min(out.get("min_abs_h0_pre", 1.0), out.get("min_abs_h1_pre", 1.0)) < 2e-6
    return out, knife


first, count = int(sys.argv[1]), int(sys.argv[2])
bad = []
for name in ("test_random_configuration_vs_c_oracle", "test_random_frames_on_the_tile_marcher_vs_c_oracle",
             "test_random_configuration_gradients_vs_oracle_autograd"):
    fn = getattr(F, name)
    for seed in range(first, first + count):
        try:
            fn(seed)
        except BaseException as e:  # noqa: BLE001
            msg = (str(e).strip().splitlines() or [repr(e)])[0][:300]
            bad.append((name, seed, type(e).__name__, msg))
            print("FAIL", name, seed, type(e).__name__, msg, flush=True)
            if not isinstance(e, AssertionError):
                traceback.print_exc()
    print("done", name, flush=True)
print("failures:", len(bad))
for b in bad:
    print(b)
    if b[0] == "test_random_configuration_gradients_vs_oracle_autograd":
        mg, knife = discrete_margins(b[1])
        print("   ", "KNIFE-EDGE (a discrete event within rounding)" if knife else "UNEXPLAINED", mg)
