"""BASELINE configs[4] (C5) on one GPU: the 48 training views of the reference's 360-degree circle trajectory
(dataLoader/scene_util.py:167-367 `cam_traj_gen(96, 'circle', radius=0.2, for_training=True)`, captured in tests/golden/poses.npz by
tests/golden/make_golden_poses.py) rendered with bf16 factor storage through render_views (device-side ray generation + the 8x8-tile
marcher + the two-kernel appearance stage). A few views against the oracle on the bf16-ROUNDED factors at the path's tolerances, and
whole-trajectory properties: every view bitwise equal to the same view rendered alone and in fp32 storage of the rounded tensors,
finite, colours in [0, 1], the views differ from each other (the trajectory really turns)."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_torch as O
from tests.conftest import GOLDEN, TINY
from tests.test_hip_parity import DEPTH_ATOL, RGB_ATOL, dev, make_field

pytestmark = pytest.mark.gpu
H, W = 24, 32
INTR = [float(max(H, W)), float(max(H, W)), W // 2, H // 2]      # dataLoader/scene_gen.py:230-237


@pytest.fixture(scope="module")
def poses():
    return dict(np.load(os.path.join(GOLDEN, "poses.npz")))


def test_c5_circle_trajectory_bf16(poses, tiny_params):
    from text2nerf_amd import render_views, generate_rays
    P48 = poses["circle_train_96"]
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    f.factor_storage = "bf16"
    rgb, depth = render_views(f, P48, INTR, H, W, N_samples=-1, white_bg=True)
    assert rgb.shape == (48, H, W, 3) and depth.shape == (48, H, W)
    assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all())
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
    # the trajectory turns: opposite views differ
    assert float((rgb[0] - rgb[24]).abs().max()) > 1e-3
    # the fp32 render of the rounded tensors is the same picture bit for bit (bf16 -> fp32 widening is exact)
    rounded = {k: v.numpy() for k, v in O.round_factors_bf16(O.params_from_numpy(tiny_params)).items()}
    fr = make_field(rounded, TINY["grid"], TINY["aabb"], TINY["near_far"])
    for v in (0, 7, 19, 33, 47):
        one_rgb, one_depth = render_views(fr, P48[v:v + 1], INTR, H, W, N_samples=-1, white_bg=True)
        assert torch.equal(one_rgb[0], rgb[v]) and torch.equal(one_depth[0], depth[v]), v
    # oracle on the rounded factors, three views spread over the circle
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"])
    PR = O.params_from_numpy(rounded)
    for v in (0, 16, 40):
        rays = generate_rays(H, W, INTR, P48[v], device=dev()).cpu()
        o_rgb, o_depth, _, _ = O.forward(cfg, PR, rays)
        assert float((rgb[v].reshape(-1, 3).cpu() - o_rgb.clamp(0, 1)).abs().max()) <= RGB_ATOL, v
        assert float((depth[v].reshape(-1).cpu() - o_depth).abs().max()) <= DEPTH_ATOL, v


def test_c3_local_fixed_views_vs_oracle(poses, tiny_params):
    """the 9 `get_local_fixed_poses2` poses of the driver's default trajectory (scene_gen.py:242), fp32 storage"""
    from text2nerf_amd import render_views, generate_rays
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    rgb, depth = render_views(f, poses["local_fixed"], INTR, H, W, N_samples=-1, white_bg=True)
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"])
    P = O.params_from_numpy(tiny_params)
    for v in (0, 2, 5, 8):
        rays = generate_rays(H, W, INTR, poses["local_fixed"][v], device=dev()).cpu()
        o_rgb, o_depth, _, _ = O.forward(cfg, P, rays)
        assert float((rgb[v].reshape(-1, 3).cpu() - o_rgb.clamp(0, 1)).abs().max()) <= RGB_ATOL, v
        assert float((depth[v].reshape(-1).cpu() - o_depth).abs().max()) <= DEPTH_ATOL, v
