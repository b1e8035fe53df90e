"""Randomised differential test: the HIP forward against the plain-C oracle over many seeded configurations (grid shapes,
boxes, near/far, step ratios, poses inside / outside the box, eval and train, white / black background, sample counts).
Catches rare-path mismatches (borders, empty rays, saturated alphas) the fixed goldens cannot enumerate."""
import numpy as np
import pytest
import torch

from oracle import oracle_torch as O
from oracle.oracle_c import COracle
from tests.test_hip_parity import DEPTH_ATOL, RGB_ATOL, W_ATOL, W_RTOL, close, dev
from text2nerf_amd import synth

pytestmark = pytest.mark.gpu


def _field(params, grid, aabb, near_far, step_ratio):
    from text2nerf_amd import TensorVMSplit
    m = TensorVMSplit(torch.tensor(aabb, dtype=torch.float32), list(grid), dev(), density_n_comp=[16] * 3, appearance_n_comp=[48] * 3,
                      app_dim=27, near_far=near_far, shadingMode="MLP_Fea_noview", alphaMask_thres=1e-4, density_shift=-10,
                      distance_scale=25, pos_pe=0, view_pe=0, fea_pe=6, featureC=128, step_ratio=step_ratio, fea2denseAct="softplus")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_configuration_vs_c_oracle(seed):
    g = np.random.Generator(np.random.PCG64(1000 + seed))
    grid = [int(g.integers(9, 70)) for _ in range(3)]
    lo = (-g.uniform(2.0, 9.0, 3)).astype(np.float32)
    hi = g.uniform(2.0, 9.0, 3).astype(np.float32)
    aabb = [lo.tolist(), hi.tolist()]
    near_far = [float(g.uniform(0.05, 1.0)), float(g.uniform(4.0, 12.0))]
    step_ratio = float(g.choice([0.5, 1.0, 2.0]))
    params = synth.make_field_params(2000 + seed, grid, density_scale=float(g.uniform(0.3, 1.6)), aabb=aabb)
    f = _field(params, grid, aabb, near_far, step_ratio)
    cfg = O.FieldConfig(aabb=aabb, grid_size=grid, near_far=near_far, step_ratio=step_ratio)
    assert f.nSamples == cfg.n_samples
    co = COracle(cfg, params)
    # cameras: inside the box, outside looking in, and a few degenerate directions
    centre = tuple(float(v) for v in (lo + (hi - lo) * g.uniform(0.2, 0.8, 3)))
    far_out = tuple(float(v) for v in (hi + g.uniform(1.0, 6.0, 3)))
    rays = np.concatenate([
        synth.frame_rays_np(11, 13, c2w=synth.look_pose(float(g.uniform(-3, 3)), float(g.uniform(-1, 1)), centre)),
        synth.frame_rays_np(9, 7, c2w=synth.look_pose(float(g.uniform(-3, 3)), float(g.uniform(-1, 1)), far_out)),
        np.array([[centre[0], centre[1], centre[2], 0, 0, 1], [centre[0], centre[1], centre[2], 1, 0, 0],
                  [centre[0], centre[1], centre[2], 0, -1, 0], [lo[0] - 1, lo[1] - 1, lo[2] - 1, 0.577, 0.577, 0.577]], np.float32)])
    rt = torch.from_numpy(rays)
    for is_train, white in ((False, True), (True, bool(seed & 1))):
        n = -1 if not is_train else int(g.integers(5, 90))
        jit = None
        torch.manual_seed(seed)
        if is_train:
            jit = torch.rand(rays.shape[0], 1)
            coin = bool(torch.rand((1,)) < 0.5) if not white else None
            torch.manual_seed(seed)
        with torch.no_grad():
            rgb, depth, z, w = f(rt, is_train=is_train, white_bg=white, N_samples=n)
        N = n if n > 0 else cfg.n_samples
        add_bg = white or (is_train and bool(coin))
        c_rgb, c_depth, c_z, c_w = co.render(rays, n_samples=N, is_train=is_train, white_bg=add_bg,
                                             jitter=None if jit is None else jit.numpy())
        assert f.stats()["evaluated"] == co.last_stats["evaluated"], (seed, is_train)
        close(z, c_z, atol=0, msg=f"seed {seed}")
        close(w, c_w, atol=W_ATOL, rtol=W_RTOL, msg=f"seed {seed}")
        close(rgb, c_rgb, atol=RGB_ATOL, msg=f"seed {seed}")
        close(depth, c_depth, atol=DEPTH_ATOL * max(1.0, near_far[1] / 8.0), msg=f"seed {seed}")


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_configuration_gradients_vs_oracle_autograd(seed):
    from tests.test_hip_parity import _grad_check
    g = np.random.Generator(np.random.PCG64(5000 + seed))
    grid = [int(g.integers(9, 40)) for _ in range(3)]
    lo = (-g.uniform(2.0, 9.0, 3)).astype(np.float32)
    hi = g.uniform(2.0, 9.0, 3).astype(np.float32)
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
    torch.manual_seed(seed)
    jit = torch.rand(rays.shape[0], 1) if is_train else None
    torch.manual_seed(seed)
    out = f(rays, is_train=is_train, white_bg=True, N_samples=n)
    ((out[0] * ca.to(dev())).sum() + 0.1 * out[1].sum() + (out[3] ** 2).sum()).backward()
    P = O.params_from_numpy(params, requires_grad=True)
    o = O.forward(cfg, P, rays, white_bg=True, is_train=is_train, n_samples=n, jitter=jit)
    lo_ = (o[0] * ca).sum() + 0.1 * o[1].sum() + (o[3] ** 2).sum()
    if not lo_.requires_grad:        # no sample survived the box / z gate: the oracle's graph is empty, all gradients must be zero
        assert all(float(p.grad.abs().max()) == 0.0 for p in f.parameters())
        return
    lo_.backward()
    _grad_check(f, {k: (v.grad if v.grad is not None else torch.zeros_like(v)).numpy() for k, v in P.items()}, rel=5e-4)


@pytest.mark.parametrize("seed", list(range(12)) + [6109])   # 6109 (round-5 campaign): an 8x11 frame whose two tiles share ONE sub-list — every ray takes
                                                               # the per-ray reservation route and some lose the race for a list's tail
def test_random_frames_on_the_tile_marcher_vs_c_oracle(seed):
    """Whole row-major frames through the 8x8-pixel tile marcher (frame_width hint) against the plain-C oracle: random grid
    shapes (anisotropic, down to 9 texels), boxes, cameras inside / outside the box / rolled by 90 degrees, ragged image
    sizes, both output modes. Exercises the or-reduced tap ranges at grid borders, the table path and the direct-gather
    steps, the LDS-transposed row writes and tiles with dead lanes."""
    g = np.random.Generator(np.random.PCG64(5000 + seed))
    grid = [int(g.integers(9, 200)) for _ in range(3)]
    lo = (-g.uniform(2.0, 9.0, 3)).astype(np.float32)
    hi = g.uniform(2.0, 9.0, 3).astype(np.float32)
    aabb = [lo.tolist(), hi.tolist()]
    near_far = [float(g.uniform(0.05, 1.0)), float(g.uniform(4.0, 12.0))]
    step_ratio = float(g.choice([0.5, 1.0]))
    params = synth.make_field_params(6000 + seed, grid, density_scale=float(g.uniform(0.3, 1.6)), aabb=aabb)
    f = _field(params, grid, aabb, near_far, step_ratio)
    cfg = O.FieldConfig(aabb=aabb, grid_size=grid, near_far=near_far, step_ratio=step_ratio)
    co = COracle(cfg, params)
    centre = tuple(float(v) for v in (lo + (hi - lo) * g.uniform(0.2, 0.8, 3)))
    far_out = tuple(float(v) for v in (hi + g.uniform(0.5, 3.0, 3)))
    for cam in (centre, far_out):
        H, W = int(g.integers(8, 70)), int(g.integers(8, 90))
        pose = synth.look_pose(float(g.uniform(-3, 3)), float(g.uniform(-1, 1)), cam)
        if seed & 1:   # roll the camera by 90 degrees: image rows run along the other world axis
            pose[:3, :3] = pose[:3, :3] @ np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], np.float32)
        rays = synth.frame_rays_np(H, W, c2w=pose)
        n = int(g.integers(17, 150)) if seed % 3 else -1
        N = n if n > 0 else cfg.n_samples
        c_rgb, c_depth, c_z, c_w = co.render(rays, n_samples=N, is_train=False, white_bg=True)
        rt = torch.from_numpy(rays).to(dev())
        f.frame_width = W
        try:
            for materialise in (True, False):
                f.materialize_weights = materialise
                with torch.no_grad():
                    rgb, depth, z, w = f(rt, is_train=False, white_bg=True, N_samples=n)
                assert f.stats()["evaluated"] == co.last_stats["evaluated"], (seed, cam, materialise)
                close(rgb, c_rgb, atol=RGB_ATOL, msg=f"seed {seed}")
                close(depth, c_depth, atol=DEPTH_ATOL * max(1.0, near_far[1] / 8.0), msg=f"seed {seed}")
                if materialise:
                    close(z, c_z, atol=0, msg=f"seed {seed}")
                    close(w, c_w, atol=W_ATOL, rtol=W_RTOL, msg=f"seed {seed}")
        finally:
            f.frame_width = 0
            f.materialize_weights = True
