"""The module-level names and stage methods of the reference beside the render call (VERDICT r3 "names the mirror does not export"):
TensorBase.sample_ray / sample_ray_ndc (models/tensorBase.py:293-323) against the reference's own outputs (golden G2) and the oracle,
positional_encoding / SHRender / RGBRender (:11-39) against golden G5, the ray_utils helpers (dataLoader/ray_utils.py:9-21,45-63,
129-171) against closed forms and numpy."""
import numpy as np
import pytest
import torch

from tests.conftest import TINY
from tests.test_hip_parity import close, dev, make_field

pytestmark = pytest.mark.gpu


def test_sample_ray_matches_the_reference(tiny, tiny_params):
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    rays = torch.from_numpy(tiny["tiny_rays"])
    pts, z, valid = f.sample_ray(rays[:, :3], rays[:, 3:6], is_train=False, N_samples=-1)
    assert pts.shape == tiny["g2_eval_pts"].shape and valid.dtype == torch.bool
    close(pts, tiny["g2_eval_pts"], atol=0)
    close(z, np.broadcast_to(tiny["g2_eval_z"], tuple(z.shape)), atol=0)
    assert np.array_equal(valid.cpu().numpy(), tiny["g2_eval_valid"])
    torch.manual_seed(77)                      # the draw of the golden (CPU generator, one value per ray)
    pts, z, valid = f.sample_ray(rays[:, :3], rays[:, 3:6], is_train=True, N_samples=40)
    close(pts, tiny["g2_train_pts"], atol=0)
    close(z, tiny["g2_train_z"], atol=0)
    assert np.array_equal(valid.cpu().numpy(), tiny["g2_train_valid"])
    assert isinstance(f, __import__("text2nerf_amd").TensorBase)


def test_sample_ray_ndc_matches_its_definition(tiny_params):
    """models/tensorBase.py:293-302 restated: pts = o + d * linspace(near, far, N) (one row for all rays), mask = inside the box."""
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    g = torch.Generator().manual_seed(4)
    o = (torch.rand(300, 3, generator=g) * 2 - 1) * torch.tensor([6.0, 5.0, 5.0])
    d = torch.nn.functional.normalize(torch.randn(300, 3, generator=g), dim=-1)
    pts, interpx, valid = f.sample_ray_ndc(o.to(dev()), d.to(dev()), is_train=False, N_samples=37)
    near, far = TINY["near_far"]
    zr = torch.linspace(near, far, 37)
    assert interpx.shape == (1, 37) and torch.equal(interpx.cpu()[0], zr)
    want = o[:, None, :] + d[:, None, :] * zr[None, :, None]
    close(pts, want.numpy(), atol=0)
    lo, hi = torch.tensor(TINY["aabb"][0]), torch.tensor(TINY["aabb"][1])
    assert torch.equal(valid.cpu(), ~((lo > want) | (want > hi)).any(-1))
    torch.manual_seed(9)
    pts_t, ix_t, _ = f.sample_ray_ndc(o.to(dev()), d.to(dev()), is_train=True, N_samples=37)
    assert ix_t.shape == (1, 37) and float((ix_t.cpu()[0] - zr).min()) >= 0 and float((ix_t.cpu()[0] - zr).max()) < (far - near) / 37
    close(pts_t, (o[:, None, :] + d[:, None, :] * ix_t.cpu()[0][None, :, None]).numpy(), atol=0)


def test_module_level_heads(tiny):
    from text2nerf_amd import RGBRender, SHRender, positional_encoding
    af = torch.from_numpy(tiny["g5_appfeat"]).to(dev())
    vd = torch.from_numpy(tiny["g5_viewdirs"]).to(dev())
    close(positional_encoding(af, 6), tiny["g5_pe"], atol=2e-6)       # device sin / cos against the CPU's
    close(SHRender(None, vd, af), tiny["g5_rgb_sh"], atol=2e-6)
    assert RGBRender(None, vd, af) is af
    with pytest.raises(Exception):
        positional_encoding(af.cpu(), 6)


def test_ray_utils_helpers(tiny):
    from text2nerf_amd import depth2dist, get_ray_directions, get_ray_directions_blender, ndc2dist, sample_pdf
    d = get_ray_directions(6, 8, [9.0, 7.5], center=[4, 3])
    b = get_ray_directions_blender(6, 8, [9.0, 7.5], center=[4, 3])
    close(b, tiny["g1_dirs_raw"] * np.array([1, -1, -1], np.float32), atol=1e-7)
    assert torch.equal(b[..., 0], d[..., 0])
    g = torch.Generator().manual_seed(2)
    z = torch.sort(torch.rand(5, 9, generator=g) * 4 + 1, dim=-1).values.to(dev())
    cos = torch.rand(5, generator=g).to(dev())
    dd = depth2dist(z, cos)
    assert dd.shape == (5, 9) and torch.allclose(dd[:, :-1], (z[:, 1:] - z[:, :-1]) * cos[:, None]) and torch.allclose(dd[:, -1], 1e10 * cos)
    p = torch.rand(5, 9, 3, generator=g).to(dev())
    nd = ndc2dist(p, cos)
    assert torch.allclose(nd[:, :-1], (p[:, 1:] - p[:, :-1]).norm(dim=-1)) and torch.allclose(nd[:, -1], 1e10 * cos)
    # sample_pdf, deterministic quantiles: the inverse CDF of a piecewise-constant density, against numpy
    bins = torch.linspace(2.0, 6.0, 9).repeat(4, 1).to(dev())
    w = (torch.rand(4, 8, generator=g) + 0.1).to(dev())
    s = sample_pdf(bins, w, 16, det=True).cpu().numpy()
    wn = w.cpu().numpy().astype(np.float64) + 1e-5
    cdf = np.concatenate([np.zeros((4, 1)), np.cumsum(wn / wn.sum(-1, keepdims=True), -1)], -1)
    u = np.linspace(0.0, 1.0, 16)
    for r in range(4):
        want = np.interp(u, cdf[r], bins[r].cpu().numpy().astype(np.float64))
        np.testing.assert_allclose(s[r], want, atol=2e-5)
    assert sample_pdf(bins, w, 7).shape == (4, 7)
