"""dda / ray_marcher (dataLoader/ray_utils.py:174-228): oracle and HIP kernel against goldens produced by the reference
(tests/golden/make_golden_marcher.py). Elementwise fp32 in the reference's operation order: compared to 1 ulp-level atol."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_torch as O
from tests.conftest import GOLDEN, TINY


@pytest.fixture(scope="module")
def gm():
    return dict(np.load(os.path.join(GOLDEN, "marcher.npz"), allow_pickle=False))


def _cases(rays):
    r8 = torch.cat([rays, torch.full((rays.shape[0], 1), 0.5), torch.full((rays.shape[0], 1), 8.0)], 1)
    return [("lin", rays, dict(n=17, lindisp=False, bbox=True)), ("disp", rays, dict(n=9, lindisp=True, bbox=True)),
            ("nf", r8, dict(n=12, lindisp=False, bbox=False))]


def _ok(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    np.testing.assert_allclose(a[fin], b[fin], rtol=2e-6, atol=2e-6)


def test_oracle_marcher_vs_reference(tiny, gm):
    rays = torch.from_numpy(tiny["tiny_rays"])
    bbox = torch.tensor(TINY["aabb"], dtype=torch.float32)
    tmin, tmax = O.dda(rays[:, :3], rays[:, 3:6], bbox)
    _ok(tmin, gm["dda_tmin"]); _ok(tmax, gm["dda_tmax"])
    for tag, r, c in _cases(rays):
        xyz, z = O.ray_marcher(r, c["n"], c["lindisp"], None, bbox if c["bbox"] else None)
        _ok(z, gm[f"m_{tag}_z"]); _ok(xyz, gm[f"m_{tag}_xyz"])
    xyz, z = O.ray_marcher(rays, 10, False, torch.from_numpy(gm["m_pert_u"]) * 1.0, bbox)
    _ok(z, gm["m_pert_z"]); _ok(xyz, gm["m_pert_xyz"])


@pytest.mark.gpu
def test_hip_marcher_vs_reference(tiny, gm):
    from text2nerf_amd import dda, ray_marcher
    rays = torch.from_numpy(tiny["tiny_rays"])
    bbox = torch.tensor(TINY["aabb"], dtype=torch.float32)
    tmin, tmax = dda(rays[:, :3].cuda(), rays[:, 3:6].cuda(), bbox)
    assert tmin.shape == (rays.shape[0], 1) and tmin.is_cuda
    _ok(tmin, gm["dda_tmin"]); _ok(tmax, gm["dda_tmax"])
    for tag, r, c in _cases(rays):
        xyz, ro, rd, z = ray_marcher(r, N_samples=c["n"], lindisp=c["lindisp"], perturb=0, bbox_3D=bbox if c["bbox"] else None)
        _ok(z, gm[f"m_{tag}_z"]); _ok(xyz, gm[f"m_{tag}_xyz"])
        assert torch.equal(ro.cpu(), r[:, :3]) and torch.equal(rd.cpu(), r[:, 3:6])
    torch.manual_seed(11)                       # CPU rays: the draws come from the CPU generator like the reference's
    xyz, ro, rd, z = ray_marcher(rays, N_samples=10, lindisp=False, perturb=1.0, bbox_3D=bbox)
    _ok(z, gm["m_pert_z"]); _ok(xyz, gm["m_pert_xyz"])
