"""TensorCP (models/tensoRF.py:306-434), the line-only CP field, against goldens produced by the reference
(tests/golden/make_golden_cp.py): the oracle's independent CP restatement on CPU; on the MI355X the HIP path, which embeds
the CP field exactly into the VM-split kernels (plane_0 = Ly (x) Lx, line_0 = Lz) and lets autograd carry the gradients
back to the lines."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_torch as O
from tests.conftest import GOLDEN, TINY
from text2nerf_amd import synth


def cp_state(sd):
    out = {k: v for k, v in sd.items() if k.startswith(("density_line", "app_line", "renderModule"))}
    for k in range(3):
        out[f"density_line.{k}"] = (out[f"density_line.{k}"] * 1.5).astype(np.float32)
        out[f"app_line.{k}"] = (out[f"app_line.{k}"] * 3.0).astype(np.float32)
    g = np.random.Generator(np.random.PCG64(23))
    out["basis_mat.weight"] = g.uniform(-0.14, 0.14, (27, 48)).astype(np.float32)
    return out


@pytest.fixture(scope="module")
def gc():
    return dict(np.load(os.path.join(GOLDEN, "cp.npz"), allow_pickle=False))


@pytest.fixture(scope="module")
def cp_params():
    return cp_state(synth.make_field_params(17, TINY["grid"], density_scale=0.9, aabb=TINY["aabb"]))


def test_oracle_cp_vs_reference(tiny, gc, cp_params):
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"])
    rays = torch.from_numpy(tiny["tiny_rays"])
    rgb, depth, z, w = O.forward(cfg, O.params_from_numpy(cp_params), rays)
    np.testing.assert_allclose(rgb.numpy(), gc["cp_eval_rgb"], atol=5e-6)
    np.testing.assert_allclose(depth.numpy(), gc["cp_eval_depth"], atol=2e-5)
    np.testing.assert_allclose(w.numpy(), gc["cp_eval_w"], atol=2e-6, rtol=2e-5)
    assert float(w.sum()) > 10


@pytest.mark.gpu
def test_hip_tensorcp_forward_and_gradients_vs_reference(tiny, gc, cp_params):
    from tests.test_hip_parity import DEPTH_ATOL, RGB_ATOL, W_ATOL, W_RTOL, close
    from text2nerf_amd import TensorCP
    m = TensorCP(torch.tensor(TINY["aabb"]), TINY["grid"], torch.device("cuda:0"), density_n_comp=[16] * 3,
                 appearance_n_comp=[48] * 3, app_dim=27, near_far=TINY["near_far"], shadingMode="MLP_Fea_noview",
                 alphaMask_thres=1e-4, density_shift=-10, distance_scale=25, pos_pe=0, view_pe=0, fea_pe=6, featureC=128,
                 step_ratio=1.0, fea2denseAct="softplus")
    assert sorted(m.state_dict().keys()) == gc["cp_state_keys"].tolist()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in cp_params.items()}, strict=True)
    rays = torch.from_numpy(tiny["tiny_rays"])
    with torch.no_grad():
        rgb, depth, z, w = m(rays)
        again = m(rays)[0]
    assert torch.equal(rgb, again)
    close(rgb, gc["cp_eval_rgb"], atol=RGB_ATOL)
    close(depth, gc["cp_eval_depth"], atol=DEPTH_ATOL)
    close(w, gc["cp_eval_w"], atol=W_ATOL, rtol=W_RTOL)
    torch.manual_seed(78)
    rgb, depth, z, w = m(rays, is_train=True, white_bg=True, N_samples=30)
    close(z, gc["cp_train_z"], atol=0)
    close(rgb, gc["cp_train_rgb"], atol=RGB_ATOL)
    dev = rgb.device
    ca, cb, cw = [torch.from_numpy(gc[k]).to(dev) for k in ("cp_ca", "cp_cb", "cp_cw")]
    ((rgb * ca).sum() + (depth * cb).sum() + (w * cw).sum()).backward()
    bad = {}
    for k, p in m.named_parameters():
        g = gc["cp_grad_" + k]
        assert p.grad is not None and tuple(p.grad.shape) == g.shape, k
        err = float(np.abs(p.grad.detach().cpu().numpy() - g).max()) / (float(np.abs(g).max()) + 1e-12)
        if err > 3e-4:
            bad[k] = err
    assert not bad, bad
    opt = torch.optim.Adam(m.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    opt.step()
    with torch.no_grad():
        rgb2 = m(rays)[0]
    assert bool(torch.isfinite(rgb2).all()) and not torch.equal(rgb2, again)


def _cp(device, cp_params):
    from text2nerf_amd import TensorCP
    m = TensorCP(torch.tensor(TINY["aabb"]), TINY["grid"], device, density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, app_dim=27,
                 near_far=TINY["near_far"], shadingMode="MLP_Fea_noview", alphaMask_thres=1e-4, density_shift=-10, distance_scale=25,
                 pos_pe=0, view_pe=0, fea_pe=6, featureC=128, step_ratio=1.0, fea2denseAct="softplus")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in cp_params.items()}, strict=True)
    return m


def test_tensorcp_shrink_host_logic_vs_reference(gc, cp_params):
    """TensorCP.shrink (models/tensoRF.py:387-416) is host index arithmetic + slicing: cropped lines, corrected aabb, grid and sample
    count against the reference's."""
    m = _cp("cpu", cp_params)
    m.shrink(torch.from_numpy(gc["cp_shrink_in"]))
    np.testing.assert_allclose(m.aabb.numpy(), gc["cp_shrink_aabb"], atol=1e-6)
    assert m.gridSize.tolist() == gc["cp_shrink_grid"].tolist() and m.nSamples == int(gc["cp_shrink_nsamples"])
    for k in range(3):
        assert np.array_equal(m.density_line[k].detach().numpy(), gc[f"cp_shrink_density_line.{k}"])
        assert np.array_equal(m.app_line[k].detach().numpy(), gc[f"cp_shrink_app_line.{k}"])


@pytest.mark.gpu
def test_hip_tensorcp_render_after_shrink(tiny, gc, cp_params):
    from tests.test_hip_parity import DEPTH_ATOL, RGB_ATOL, close
    m = _cp(torch.device("cuda:0"), cp_params)
    rays = torch.from_numpy(tiny["tiny_rays"])
    with torch.no_grad():
        m(rays)
    m.shrink(torch.from_numpy(gc["cp_shrink_in"]))
    with torch.no_grad():
        rgb, depth, _, _ = m(rays)
    close(rgb, gc["cp_shrink_rgb"], atol=RGB_ATOL)
    close(depth, gc["cp_shrink_depth"], atol=DEPTH_ATOL)
