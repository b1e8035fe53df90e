"""The view-dependent heads MLPRender_Fea / MLPRender (models/tensorBase.py:62-86,137-159) against goldens produced by the
reference (tests/golden/make_golden_heads.py): the oracle on CPU; the HIP general-head path (csrc/t2n_heads.hip) on the
MI355X, forward and gradients. MLP_PE is rejected: the reference's class cannot run (see make_golden_heads_cases.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import oracle_torch as O
from tests.conftest import GOLDEN, TINY
from text2nerf_amd import synth

sys.path.insert(0, GOLDEN)
from make_golden_heads_cases import HEADS  # noqa: E402


@pytest.fixture(scope="module")
def gh():
    return dict(np.load(os.path.join(GOLDEN, "heads.npz"), allow_pickle=False))


def _params(kw):
    return synth.make_field_params(31, TINY["grid"], density_scale=0.9, aabb=TINY["aabb"], shading_mode=kw["shadingMode"],
                                   fea_pe=kw["fea_pe"], view_pe=kw["view_pe"], pos_pe=kw["pos_pe"])


@pytest.mark.parametrize("tag", list(HEADS))
def test_oracle_heads_vs_reference(tiny, gh, tag):
    kw = HEADS[tag]
    params = _params(kw)
    assert params["renderModule.mlp.0.weight"].shape[1] == int(gh[f"{tag}_in_mlpC"])
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"], shading_mode=kw["shadingMode"],
                        fea_pe=kw["fea_pe"], view_pe=kw["view_pe"], pos_pe=kw["pos_pe"])
    rays = torch.from_numpy(tiny["tiny_rays"])
    rgb, depth, z, w = O.forward(cfg, O.params_from_numpy(params), rays)
    np.testing.assert_allclose(rgb.numpy(), gh[f"{tag}_eval_rgb"], atol=5e-6)
    np.testing.assert_allclose(depth.numpy(), gh[f"{tag}_eval_depth"], atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(HEADS))
def test_hip_heads_forward_and_gradients_vs_reference(tiny, gh, tag):
    from tests.test_hip_parity import DEPTH_ATOL, RGB_ATOL, close, dev
    from text2nerf_amd import TensorVMSplit
    kw = HEADS[tag]
    m = TensorVMSplit(torch.tensor(TINY["aabb"]), TINY["grid"], dev(), density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, app_dim=27,
                      near_far=TINY["near_far"], alphaMask_thres=1e-4, density_shift=-10, distance_scale=25, featureC=128,
                      step_ratio=1.0, fea2denseAct="softplus", **kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in _params(kw).items()}, strict=True)
    rays = torch.from_numpy(tiny["tiny_rays"])
    with torch.no_grad():
        rgb, depth, z, w = m(rays)
    close(rgb, gh[f"{tag}_eval_rgb"], atol=RGB_ATOL)
    close(depth, gh[f"{tag}_eval_depth"], atol=DEPTH_ATOL)
    torch.manual_seed(55)
    rgb, depth, z, w = m(rays, is_train=True, white_bg=True, N_samples=36)
    close(rgb, gh[f"{tag}_train_rgb"], atol=RGB_ATOL)
    ca = torch.from_numpy(gh[f"{tag}_ca"]).to(dev())
    ((rgb * ca).sum() + 0.1 * depth.sum() + (w ** 2).sum()).backward()
    bad = {}
    for k, p in m.named_parameters():
        g = gh[f"{tag}_grad_" + k]
        assert p.grad is not None and tuple(p.grad.shape) == g.shape, k
        err = float(np.abs(p.grad.detach().cpu().numpy() - g).max()) / (float(np.abs(g).max()) + 1e-12)
        if err > 3e-4:
            bad[k] = err
    assert not bad, bad


@pytest.mark.gpu
def test_mlp_pe_head_is_rejected():
    from tests.test_hip_parity import dev
    from text2nerf_amd import TensorVMSplit
    from text2nerf_amd._lib import T2NError
    with pytest.raises(T2NError):
        TensorVMSplit(torch.tensor([[-1.0] * 3, [1.0] * 3]), [8, 8, 8], dev(), shadingMode="MLP_PE")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(HEADS))
def test_hip_heads_in_several_passes_equal_one_pass(tiny, tag):
    """the general heads' forward reads no count on the host and allocates nothing — the activation scratch is part of
    the caller's workspace and the rows are taken in passes of its capacity, every kernel clipping to a device-side plan. With ONE
    scratch row per ray (instead of 32) a frame's ~7 appearance rows per ray need several passes: bit-identical colours, and the
    sub-launch form of the reference call (chunked rays) agrees as well."""
    from tests.test_hip_parity import dev
    from text2nerf_amd import TensorVMSplit, OctreeRender_trilinear_fast
    kw = HEADS[tag]
    def make():
        m = TensorVMSplit(torch.tensor(TINY["aabb"]), TINY["grid"], dev(), density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, app_dim=27,
                          near_far=TINY["near_far"], alphaMask_thres=1e-4, density_shift=-10, distance_scale=25, featureC=128,
                          step_ratio=1.0, fea2denseAct="softplus", **kw)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in _params(kw).items()}, strict=True)
        return m
    rays = torch.from_numpy(synth.frame_rays_np(40, 48, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0))))     # 1 920 rays
    a, b = make(), make()
    b.head_scratch_rows_per_ray = 1
    with torch.no_grad():
        ra = a(rays.to(dev()))
        rb = b(rays.to(dev()))
        rc = OctreeRender_trilinear_fast(rays, b, chunk=512, N_samples=-1, white_bg=True, is_train=False, device=dev())
    rows = a.stats()["appearance"]
    assert rows > 3 * (rays.shape[0] + 256), rows        # more rows than one pass of the small scratch holds, several times over
    assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1])
    # (the chunked call marches per ray where the whole frame may take the tile marcher: same samples, weights to ~2e-6)
    assert float((ra[0] - rc[0]).abs().max()) <= 2e-5 and float((ra[1] - rc[2]).abs().max()) <= 2e-4
