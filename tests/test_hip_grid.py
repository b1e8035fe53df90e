"""GPU parity of the coarse-to-fine / occupancy-mask entry points (t2n_compute_alpha, t2n_dense_alpha,
t2n_alpha_volume, t2n_upsample_bilinear, t2n_filter_rays_alpha and the host-side shrink) against the golden vectors
generated from the reference. Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest
import torch

from text2nerf_amd import synth
from tests.conftest import TINY
from tests.test_hip_parity import close, dev, make_field

pytestmark = pytest.mark.gpu


@pytest.fixture()
def field(tiny_params):
    return make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])


@pytest.fixture()
def cfield(tiny_params):
    return make_field(synth.concentrate_density({k: v.copy() for k, v in tiny_params.items()}), TINY["grid"], TINY["aabb"],
                      TINY["near_far"])


def test_compute_alpha(grid_ops, field):
    from text2nerf_amd import AlphaGridMask
    pts = torch.from_numpy(grid_ops["pts"]).to(dev())
    close(field.compute_alpha(pts, length=0.37), grid_ops["alpha_nomask"], atol=2e-6, rtol=1e-5)
    field.alphaMask = AlphaGridMask(dev(), torch.from_numpy(grid_ops["mask_aabb"]), torch.from_numpy(grid_ops["mask_volume"]))
    close(field.compute_alpha(pts, length=0.37), grid_ops["alpha_mask"], atol=2e-6, rtol=1e-5)
    # shape is preserved like the reference's .view(xyz_locs.shape[:-1])
    assert field.compute_alpha(pts.view(30, 100, 3)).shape == (30, 100)


def test_filter_rays_alpha(grid_ops, field):
    from text2nerf_amd import AlphaGridMask
    from text2nerf_amd._lib import T2NError
    rays = torch.from_numpy(grid_ops["filter_rays"])
    with pytest.raises(T2NError):
        field.filtering_rays(rays, torch.zeros(rays.shape[0], 3), bbox_only=False)
    field.alphaMask = AlphaGridMask(dev(), torch.from_numpy(grid_ops["mask_aabb"]), torch.from_numpy(grid_ops["filter_volume"]))
    kept = field.filtering_rays(rays, torch.zeros(rays.shape[0], 3), N_samples=24, bbox_only=False)
    assert np.array_equal(kept[0].numpy(), grid_ops["filter_rays"][grid_ops["filter_keep"]])


def test_dense_alpha_update_mask_and_shrink(grid_ops, cfield):
    g = grid_ops["dense_alpha"].shape
    alpha, xyz = cfield.getDenseAlpha(g)
    close(alpha, grid_ops["dense_alpha"], atol=2e-6, rtol=1e-5)
    assert xyz.shape == (*g, 3)
    cfield.alphaMask_thres = float(grid_ops["mask_thres"])
    new_aabb = cfield.updateAlphaMask(g)
    assert np.array_equal(cfield.alphaMask.alpha_volume[0, 0].cpu().numpy(), grid_ops["upd_volume"])
    close(new_aabb, grid_ops["upd_new_aabb"], atol=0)
    rays = torch.from_numpy(grid_ops["filter_rays"]).to(dev())
    rgb, depth, _, _ = cfield(rays, is_train=False, white_bg=True)
    close(rgb, grid_ops["upd_rgb"], atol=1e-4)
    close(depth, grid_ops["upd_depth"], atol=2e-4)
    # shrink to that box: cropped factors bit-exact, corrected aabb, new step size / sample count, render parity
    cfield.shrink(new_aabb)
    assert cfield.gridSize.tolist() == grid_ops["shrink_grid"].tolist()
    close(cfield.aabb, grid_ops["shrink_aabb"], atol=0)
    assert cfield.nSamples == int(grid_ops["shrink_step"][1]) and float(cfield.stepSize) == grid_ops["shrink_step"][0]
    close(cfield.density_plane[0], grid_ops["shrink_dplane0"], atol=0)
    close(cfield.app_line[2], grid_ops["shrink_aline2"], atol=0)
    cfield.alphaMask = None
    rgb, depth, _, _ = cfield(rays, is_train=False, white_bg=True)
    close(rgb, grid_ops["shrink_rgb"], atol=1e-4)
    close(depth, grid_ops["shrink_depth"], atol=2e-4)


def test_upsample_volume_grid(grid_ops, field):
    res = grid_ops["up_res"].tolist()
    field.upsample_volume_grid(res)
    for i in range(3):
        close(field.density_plane[i], grid_ops[f"up_density_plane{i}"], atol=1e-6, rtol=1e-6)
        close(field.density_line[i], grid_ops[f"up_density_line{i}"], atol=1e-6, rtol=1e-6)
    close(field.app_plane[1], grid_ops["up_app_plane1"], atol=1e-6, rtol=1e-6)
    close(field.app_line[0], grid_ops["up_app_line0"], atol=1e-6, rtol=1e-6)
    assert field.nSamples == int(grid_ops["up_step"][1]) and float(field.stepSize) == grid_ops["up_step"][0]
    assert all(p.requires_grad for p in field.parameters())          # still leaf nn.Parameters for the optimiser
    rays = torch.from_numpy(grid_ops["filter_rays"]).to(dev())
    rgb, depth, _, _ = field(rays, is_train=False, white_bg=True)
    close(rgb, grid_ops["up_rgb"], atol=1e-4)
    close(depth, grid_ops["up_depth"], atol=2e-4)


def test_upsample_300_matches_oracle(field):
    """A production-sized resize (24x20x16 -> 300^3 planes would be 69 MB; use 150x130x110) against the oracle restatement."""
    from oracle import oracle_torch as O
    P = O.params_from_numpy({k: v.detach().cpu().numpy() for k, v in field.state_dict().items()})
    res = [150, 130, 110]
    up = O.upsample_field(P, res)
    field.upsample_volume_grid(res)
    sd = field.state_dict()
    for k in ("density_plane.0", "density_plane.2", "app_plane.1", "density_line.1", "app_line.2"):
        close(sd[k], up[k].numpy(), atol=1e-6, rtol=1e-6, msg=k)
