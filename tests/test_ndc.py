"""The NDC branch (ndc_ray=True; not used by the Text2NeRF driver): sample_ray_ndc and forward's ndc lines
(models/tensorBase.py:293-302,441-446), ndc_rays_blender / ndc_rays (dataLoader/ray_utils.py:88-124) — oracle and HIP path
against goldens produced by the reference (tests/golden/make_golden_ndc.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_torch as O
from tests.conftest import GOLDEN, TINY


@pytest.fixture(scope="module")
def gn():
    return dict(np.load(os.path.join(GOLDEN, "ndc.npz"), allow_pickle=False))


@pytest.mark.parametrize("tag", ["mlp", "sh"])
def test_oracle_ndc_forward(gn, tiny_params, tiny_params_sh, tag):
    params = tiny_params if tag == "mlp" else tiny_params_sh
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"],
                        shading_mode="MLP_Fea_noview" if tag == "mlp" else "SH")
    rays = torch.from_numpy(gn["ndc_rays_in"])
    rgb, depth, z, w = O.forward(cfg, O.params_from_numpy(params), rays, ndc=True)
    assert np.array_equal(z.numpy(), gn[f"ndc_{tag}_eval_z"])
    np.testing.assert_allclose(rgb.numpy(), gn[f"ndc_{tag}_eval_rgb"], atol=5e-6)
    np.testing.assert_allclose(w.numpy(), gn[f"ndc_{tag}_eval_w"], atol=2e-6, rtol=2e-5)
    np.testing.assert_allclose(depth.numpy(), gn[f"ndc_{tag}_eval_depth"], atol=2e-5)
    rgb, depth, z, w = O.forward(cfg, O.params_from_numpy(params), rays, ndc=True, is_train=True, n_samples=40,
                                 jitter=torch.from_numpy(gn[f"ndc_{tag}_jit"]))
    assert np.array_equal(z.numpy(), gn[f"ndc_{tag}_train_z"])
    np.testing.assert_allclose(rgb.numpy(), gn[f"ndc_{tag}_train_rgb"], atol=5e-6)
    np.testing.assert_allclose(w.numpy(), gn[f"ndc_{tag}_train_w"], atol=2e-6, rtol=2e-5)


def test_oracle_ndc_rays(gn):
    o, d = torch.from_numpy(gn["nr_o"]), torch.from_numpy(gn["nr_d"])
    a, b = O.ndc_rays(378, 504, 400.0, 1.0, o, d, blender=True)
    np.testing.assert_allclose(a.numpy(), gn["nr_blender_o"], rtol=3e-6, atol=3e-6)
    np.testing.assert_allclose(b.numpy(), gn["nr_blender_d"], rtol=3e-6, atol=3e-6)
    a, b = O.ndc_rays(378, 504, 400.0, 1.0, o, -d, blender=False)
    np.testing.assert_allclose(a.numpy(), gn["nr_cv_o"], rtol=3e-6, atol=3e-6)
    np.testing.assert_allclose(b.numpy(), gn["nr_cv_d"], rtol=3e-6, atol=3e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["mlp", "sh"])
def test_hip_ndc_forward_and_gradients(gn, tiny_params, tiny_params_sh, tag):
    from tests.test_hip_parity import DEPTH_ATOL, RGB_ATOL, W_ATOL, W_RTOL, _grad_check, close, make_field
    params = tiny_params if tag == "mlp" else tiny_params_sh
    shading = "MLP_Fea_noview" if tag == "mlp" else "SH"
    f = make_field(params, TINY["grid"], TINY["aabb"], TINY["near_far"], shading=shading)
    rays = torch.from_numpy(gn["ndc_rays_in"])
    with torch.no_grad():
        rgb, depth, z, w = f(rays, ndc_ray=True)
    assert tuple(z.shape) == (1, gn[f"ndc_{tag}_eval_z"].shape[1])
    close(z, gn[f"ndc_{tag}_eval_z"], atol=0)
    close(w, gn[f"ndc_{tag}_eval_w"], atol=W_ATOL, rtol=W_RTOL)
    close(rgb, gn[f"ndc_{tag}_eval_rgb"], atol=RGB_ATOL)
    close(depth, gn[f"ndc_{tag}_eval_depth"], atol=DEPTH_ATOL)
    # train mode: the shared jitter row comes from the DEVICE generator here (the reference's chunk lives on the GPU), so the
    # oracle is fed the depth row the call returned
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"], shading_mode=shading)
    torch.manual_seed(4)
    if tag == "mlp":
        out = f(rays, ndc_ray=True, is_train=True, N_samples=40)
    else:
        with torch.no_grad():
            out = f(rays, ndc_ray=True, is_train=True, N_samples=40)
    zrow = out[2].detach().cpu()
    near, far = TINY["near_far"]
    jit = (zrow - torch.linspace(near, far, 40).unsqueeze(0)) / ((far - near) / 40)
    assert float(jit.min()) > -1e-4 and float(jit.max()) < 1 + 1e-4
    P = O.params_from_numpy(params, requires_grad=(tag == "mlp"))
    o = O.forward(cfg, P, rays, ndc=True, is_train=True, n_samples=40, jitter=jit)
    # (the oracle rebuilds z as linspace + jit * scale: equal to the returned row up to 1 ulp; compare at tolerance)
    close(out[2], o[2].detach().numpy(), atol=2e-6)
    close(out[3], o[3].detach().numpy(), atol=2e-5, rtol=2e-4)
    close(out[0], o[0].detach().numpy(), atol=RGB_ATOL)
    if tag == "mlp":
        ((out[0] ** 2).sum() + (out[1] * 0.1).sum() + (out[3] ** 2).sum()).backward()
        ((o[0] ** 2).sum() + (o[1] * 0.1).sum() + (o[3] ** 2).sum()).backward()
        _grad_check(f, {k: v.grad.numpy() for k, v in P.items()}, rel=1e-3)


@pytest.mark.gpu
def test_hip_ndc_rays(gn):
    from text2nerf_amd import ndc_rays, ndc_rays_blender
    o, d = torch.from_numpy(gn["nr_o"]).cuda(), torch.from_numpy(gn["nr_d"]).cuda()
    a, b = ndc_rays_blender(378, 504, 400.0, 1.0, o, d)
    np.testing.assert_allclose(a.cpu().numpy(), gn["nr_blender_o"], rtol=3e-6, atol=3e-6)
    np.testing.assert_allclose(b.cpu().numpy(), gn["nr_blender_d"], rtol=3e-6, atol=3e-6)
    a, b = ndc_rays(378, 504, 400.0, 1.0, o, -d)
    np.testing.assert_allclose(a.cpu().numpy(), gn["nr_cv_o"], rtol=3e-6, atol=3e-6)
    np.testing.assert_allclose(b.cpu().numpy(), gn["nr_cv_d"], rtol=3e-6, atol=3e-6)
