"""TensorVM (models/tensoRF.py:4-136), the stacked-coefficient variant, against goldens produced by the reference
(tests/golden/make_golden_vm.py): the oracle via vm_to_split on CPU; the HIP path (same kernels as TensorVMSplit, parameters
read in place through slices) on the MI355X, forward and gradients."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import oracle_torch as O
from tests.conftest import GOLDEN, TINY
from text2nerf_amd import synth

sys.path.insert(0, GOLDEN)
GRID = [20, 20, 20]


def vm_state(sd):
    out = {k: v for k, v in sd.items() if not k.startswith(("density_", "app_"))}
    out["plane_coef"] = np.concatenate([np.concatenate([sd[f"app_plane.{k}"], sd[f"density_plane.{k}"]], 1) for k in range(3)], 0)
    out["line_coef"] = np.concatenate([np.concatenate([sd[f"app_line.{k}"], sd[f"density_line.{k}"]], 1) for k in range(3)], 0)
    return out


@pytest.fixture(scope="module")
def gv():
    return dict(np.load(os.path.join(GOLDEN, "vm.npz"), allow_pickle=False))


@pytest.fixture(scope="module")
def vm_params():
    return vm_state(synth.make_field_params(13, GRID, density_scale=0.9, aabb=TINY["aabb"]))


def test_oracle_vm_vs_reference(tiny, gv, vm_params):
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=GRID, near_far=TINY["near_far"])
    P = O.vm_to_split(O.params_from_numpy(vm_params), 16, 48)
    rays = torch.from_numpy(tiny["tiny_rays"])
    rgb, depth, z, w = O.forward(cfg, P, rays)
    assert np.array_equal(z.numpy(), np.broadcast_to(gv["vm_eval_z"], z.shape))
    np.testing.assert_allclose(rgb.numpy(), gv["vm_eval_rgb"], atol=5e-6)
    np.testing.assert_allclose(depth.numpy(), gv["vm_eval_depth"], atol=2e-5)
    np.testing.assert_allclose(w.numpy(), gv["vm_eval_w"], atol=2e-6, rtol=2e-5)


def _make_vm(vm_params):
    from text2nerf_amd import TensorVM
    m = TensorVM(torch.tensor(TINY["aabb"]), GRID, torch.device("cuda:0"), density_n_comp=16, appearance_n_comp=48, app_dim=27,
                 near_far=TINY["near_far"], shadingMode="MLP_Fea_noview", alphaMask_thres=1e-4, density_shift=-10,
                 distance_scale=25, pos_pe=0, view_pe=0, fea_pe=6, featureC=128, step_ratio=1.0, fea2denseAct="softplus")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vm_params.items()}, strict=True)
    return m


@pytest.mark.gpu
def test_hip_tensorvm_forward_and_gradients_vs_reference(tiny, gv, vm_params):
    from tests.test_hip_parity import DEPTH_ATOL, RGB_ATOL, W_ATOL, W_RTOL, close
    m = _make_vm(vm_params)
    assert sorted(m.state_dict().keys()) == gv["vm_state_keys"].tolist()
    assert [len(g["params"]) if isinstance(g["params"], list) else 1 for g in m.get_optparam_groups()][:2] == [1, 1]
    rays = torch.from_numpy(tiny["tiny_rays"])
    with torch.no_grad():
        rgb, depth, z, w = m(rays)
    close(z, np.broadcast_to(gv["vm_eval_z"], z.shape), atol=0)
    close(rgb, gv["vm_eval_rgb"], atol=RGB_ATOL)
    close(depth, gv["vm_eval_depth"], atol=DEPTH_ATOL)
    close(w, gv["vm_eval_w"], atol=W_ATOL, rtol=W_RTOL)
    torch.manual_seed(77)
    rgb, depth, z, w = m(rays, is_train=True, white_bg=True, N_samples=30)
    close(z, gv["vm_train_z"], atol=0)
    close(rgb, gv["vm_train_rgb"], atol=RGB_ATOL)
    dev = rgb.device
    ca, cb, cw = [torch.from_numpy(gv[k]).to(dev) for k in ("vm_ca", "vm_cb", "vm_cw")]
    ((rgb * ca).sum() + (depth * cb).sum() + (w * cw).sum()).backward()
    bad = {}
    for k, p in m.named_parameters():
        g = gv["vm_grad_" + k]
        assert p.grad is not None and tuple(p.grad.shape) == g.shape, k
        err = float(np.abs(p.grad.detach().cpu().numpy() - g).max()) / (float(np.abs(g).max()) + 1e-12)
        if err > 2e-4:
            bad[k] = err
    assert not bad, bad
    # an optimiser step on the stacked tensors re-uploads transparently; checkpoint kwargs round-trip
    opt = torch.optim.Adam(m.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    opt.step()
    with torch.no_grad():
        rgb2 = m(rays)[0]
    assert bool(torch.isfinite(rgb2).all()) and not torch.equal(rgb2.cpu(), torch.from_numpy(gv["vm_eval_rgb"]))
    kw = m.get_kwargs()
    assert kw["density_n_comp"] == 16 and kw["appearance_n_comp"] == 48


def test_oracle_vm_upsample_vs_reference(tiny, gv, vm_params):
    """TensorVM.upsample_volume_grid (models/tensoRF.py:126-136): the oracle's align_corners bilinear resize reproduces the reference's
    resized stacked tensors and the render after them."""
    P = O.params_from_numpy(vm_params)
    pl = P["plane_coef"]
    up_p = O.upsample_bilinear(pl.reshape(1, -1, 20, 20), 27, 27).reshape(3, -1, 27, 27)
    up_l = O.upsample_bilinear(P["line_coef"].reshape(1, -1, 20, 1), 27, 1).reshape(3, -1, 27, 1)
    np.testing.assert_allclose(up_p.numpy(), gv["vm_up27_plane"], atol=2e-6)
    np.testing.assert_allclose(up_l.numpy(), gv["vm_up27_line"], atol=2e-6)
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=[27] * 3, near_far=TINY["near_far"])
    assert cfg.n_samples == int(gv["vm_up27_nsamples"])
    P2 = dict(P, plane_coef=torch.from_numpy(gv["vm_up27_plane"]), line_coef=torch.from_numpy(gv["vm_up27_line"]))
    rgb, depth, _, _ = O.forward(cfg, O.vm_to_split(P2, 16, 48), torch.from_numpy(tiny["tiny_rays"]))
    np.testing.assert_allclose(rgb.numpy(), gv["vm_up27_rgb"], atol=5e-6)


@pytest.mark.gpu
def test_hip_tensorvm_upsample_volume_grid(tiny, gv, vm_params):
    from tests.test_hip_parity import DEPTH_ATOL, RGB_ATOL, close
    m = _make_vm(vm_params)
    rays = torch.from_numpy(tiny["tiny_rays"])
    with torch.no_grad():
        m(rays)                                  # a native field at 20^3 exists before the resize
    m.upsample_volume_grid([27, 27, 27])
    assert tuple(m.plane_coef.shape) == (3, 64, 27, 27) and tuple(m.line_coef.shape) == (3, 64, 27, 1)
    close(m.plane_coef, gv["vm_up27_plane"], atol=2e-6)
    close(m.line_coef, gv["vm_up27_line"], atol=2e-6)
    assert m.nSamples == int(gv["vm_up27_nsamples"]) and m.gridSize.tolist() == [27, 27, 27]
    with torch.no_grad():
        rgb, depth, _, _ = m(rays)
    close(rgb, gv["vm_up27_rgb"], atol=RGB_ATOL)
    close(depth, gv["vm_up27_depth"], atol=DEPTH_ATOL)
