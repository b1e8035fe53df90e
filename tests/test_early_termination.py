"""Early ray termination (SURVEY.md 7 hard parts): an eval-only mode of both marchers. The reference never
terminates (models/tensorBase.py:19-26 raw2alpha, :494-505 compositing evaluate every in-box sample), so these tests bound the
deviation against the oracle instead of asking for parity: acc / colour < eps, depth < eps * z range, evaluated samples <= the
reference's; and they check that the mode is OFF wherever the reference's per-sample outputs are handed out (weights, z_vals,
training)."""
import ctypes as C

import numpy as np
import pytest
import torch

from text2nerf_amd import synth
from tests.conftest import TINY
from tests.test_hip_parity import DEPTH_ATOL, RGB_ATOL, dev, make_field

pytestmark = pytest.mark.gpu


def _oracle(params, rays_np, n_samples):
    from oracle import oracle_torch as O
    from oracle.oracle_c import COracle
    co = COracle(O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"]), params)
    rgb, depth, _, _ = co.render(rays_np, n_samples=n_samples, want_weights=False)
    return rgb, depth, co.last_stats


def _check(f, rays_np, o_rgb, o_depth, o_stats, zmax, expect_fewer):
    rays = torch.from_numpy(rays_np).to(dev())
    with torch.no_grad():
        rgb0, depth0, _, _ = f(rays)
    ev0 = f.stats()["evaluated"]
    assert ev0 == o_stats["evaluated"]                      # off (make_field): sample for sample
    f.early_termination = 1e-6
    with torch.no_grad():
        rgb1, depth1, _, _ = f(rays)
    ev1 = f.stats()["evaluated"]
    assert ev1 <= ev0
    if expect_fewer:
        assert ev1 < expect_fewer * ev0, (ev1, ev0)
    # against the oracle at the parity tolerances, and against the un-terminated render at the mode's own bound
    assert np.abs(rgb1.cpu().numpy() - o_rgb).max() <= RGB_ATOL
    assert np.abs(depth1.cpu().numpy() - o_depth).max() <= DEPTH_ATOL
    assert float((rgb1 - rgb0).abs().max()) <= 2e-6
    assert float((depth1 - depth0).abs().max()) <= 1e-6 * zmax + 1e-6
    return ev0, ev1, float((rgb1 - rgb0).abs().max()), float((depth1 - depth0).abs().max())


@pytest.mark.parametrize("frame_width", [0, 48])      # per-ray marcher (64-sample blocks) / 8x8-tile marcher
@pytest.mark.parametrize("density_scale", [0.9, 3.0])  # the tiny goldens' field / an opaque one (most rays die within a few steps)
def test_termination_bounds_vs_oracle(frame_width, density_scale):
    params = synth.make_field_params(11, TINY["grid"], density_scale=density_scale, aabb=TINY["aabb"])
    f = make_field(params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    f.materialize_weights = False
    f.frame_width = frame_width
    rays_np = synth.frame_rays_np(40, 48, c2w=synth.look_pose(0.2, -0.1, (0.3, 0.2, -1.5)))
    o_rgb, o_depth, o_stats = _oracle(params, rays_np, f.nSamples)
    zmax = TINY["near_far"][0] + float(f.stepSize) * f.nSamples
    # (the tiny box holds at most 33 steps: neither the per-ray marcher's 64-sample blocks nor the tile marcher's look at the transmittance
    # every 16 steps stops anything inside it; what is checked here is the bound, the saving is checked at 300^3 below)
    fewer = None
    ev0, ev1, dc, dd = _check(f, rays_np, o_rgb, o_depth, o_stats, zmax, fewer)
    print(f"frame_width {frame_width}, density x{density_scale}: evaluated {ev1} of {ev0}, max colour change {dc:.1e}, max depth change {dd:.1e}")


@pytest.mark.parametrize("frame_width", [0, 64])
def test_termination_in_fog_at_production_size(frame_width):
    """300^3, scene S2 (fog: ~16 % of space opaque), 518 samples per ray, a 64x64 view: both marchers stop most rays long before the far
    side of the box."""
    from oracle import oracle_torch as O
    from oracle.oracle_c import COracle
    aabb = [[-8.0] * 3, [8.0] * 3]
    params = synth.make_field_params(1, [300] * 3, scene="S2", aabb=aabb)
    f = make_field(params, [300] * 3, aabb, [0.5, 8.0])
    f.materialize_weights = False
    f.frame_width = frame_width
    rays_np = synth.frame_rays_np(64, 64)
    co = COracle(O.FieldConfig(aabb=aabb, grid_size=[300] * 3), params)
    o_rgb, o_depth, _, _ = co.render(rays_np, n_samples=f.nSamples, want_weights=False)
    ev0, ev1, dc, dd = _check(f, rays_np, o_rgb, o_depth, co.last_stats, 0.5 + float(f.stepSize) * f.nSamples, 0.9)
    print(f"S2 300^3, frame_width {frame_width}: evaluated {ev1} of {ev0} ({ev1 / ev0:.2f}), max colour change {dc:.1e}, max depth change {dd:.1e}")


def test_termination_is_off_where_weights_are_returned(tiny_params):
    """OctreeRender_trilinear_fast hands the reference's weights / z_vals rows to its caller (renderer.py:28-42) and training needs
    every sample: neither is terminated, whatever the field's setting."""
    from text2nerf_amd import OctreeRender_trilinear_fast
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    rays = torch.from_numpy(synth.frame_rays_np(16, 24))
    with torch.no_grad():
        ref = OctreeRender_trilinear_fast(rays, f, chunk=4096, N_samples=-1, white_bg=True, is_train=False, device=dev())
    ev0 = f.stats()["evaluated"]
    f.early_termination = 1e-4
    with torch.no_grad():
        got = OctreeRender_trilinear_fast(rays, f, chunk=4096, N_samples=-1, white_bg=True, is_train=False, device=dev())
    assert f.stats()["evaluated"] == ev0
    for a, b in zip(ref, got):
        if a is not None:
            assert torch.equal(a, b)
    torch.manual_seed(5)
    out0 = f(rays.to(dev()), is_train=True, N_samples=40)
    f.early_termination = 0.0
    torch.manual_seed(5)
    out1 = f(rays.to(dev()), is_train=True, N_samples=40)
    assert torch.equal(out0[0], out1[0]) and torch.equal(out0[3], out1[3])


def test_termination_threshold_is_bounded_by_the_appearance_threshold(tiny_params):
    """Above rayMarch_weight_thres a skipped sample could have been an appearance-list entry: rejected."""
    from text2nerf_amd import _lib
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    lib = _lib.load()
    h = f.sync_params()
    assert lib.t2n_field_set_early_termination(h, C.c_float(1e-3)) != 0
    assert lib.t2n_field_set_early_termination(h, C.c_float(-1.0)) != 0
    assert lib.t2n_field_set_early_termination(h, C.c_float(1e-4)) == 0
    assert lib.t2n_field_set_early_termination(h, C.c_float(0.0)) == 0


def test_zero_appearance_threshold_with_termination_requested(tiny_params):
    """rayMarch_weight_thres = 0 (`weight > thres`, models/tensorBase.py:477) is a legitimate reference setting; a requested
    termination threshold above it is clamped (0 = off) instead of failing every sync_params(), and a threshold lowered on a live
    handle takes the termination bound with it."""
    from text2nerf_amd import _lib
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    f.materialize_weights = False
    rays = torch.from_numpy(synth.frame_rays_np(16, 24)).to(dev())
    f.early_termination = 1e-6
    with torch.no_grad():
        f(rays)                                   # handle live with eps = 1e-6 under thres = 1e-4
    f.rayMarch_weight_thres = 0.0
    f.update_stepSize(f.gridSize.tolist() if hasattr(f.gridSize, "tolist") else list(f.gridSize))   # t2n_field_set_desc on the live handle: the bound follows
    with torch.no_grad():
        a = f(rays)
    ev = f.stats()["evaluated"]
    g = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])     # a fresh field built with thres = 0 and eps requested
    g.materialize_weights = False
    g.rayMarch_weight_thres = 0.0
    g.early_termination = 1e-6
    with torch.no_grad():
        c = g(rays)                               # sync_params() clamps instead of raising
    g.early_termination = 0.0
    with torch.no_grad():
        b = g(rays)
    assert g.stats()["evaluated"] == ev
    for x in (a, c):
        assert torch.equal(x[0], b[0]) and torch.equal(x[1], b[1])
