"""bf16 factor storage (BASELINE.json configs[4]) on the MI355X: the render must equal the fp32 render of the bf16-rounded
factor tensors BIT FOR BIT (bf16 -> fp32 widening is exact and the arithmetic after the loads is the same code), match
golden G11 (the reference itself on rounded tensors) within the path's tolerances, and its gradients must match the
reference's autograd at the rounded point."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_torch as O
from tests.conftest import GOLDEN, TINY
from tests.test_hip_parity import DEPTH_ATOL, RGB_ATOL, W_ATOL, W_RTOL, close, dev, make_field

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g11():
    return dict(np.load(os.path.join(GOLDEN, "bf16.npz"), allow_pickle=False))


def _rounded(params):
    return {k: v.numpy() for k, v in O.round_factors_bf16(O.params_from_numpy(params)).items()}


def test_bf16_storage_equals_fp32_render_of_rounded_factors_bitwise(tiny, tiny_params):
    rays = torch.from_numpy(tiny["tiny_rays"])
    fh = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    fh.factor_storage = "bf16"
    fr = make_field(_rounded(tiny_params), TINY["grid"], TINY["aabb"], TINY["near_far"])
    for kw in (dict(is_train=False, white_bg=True, N_samples=-1), dict(is_train=True, white_bg=True, N_samples=40)):
        with torch.no_grad():
            torch.manual_seed(5)
            a = fh(rays, **kw)
            torch.manual_seed(5)
            b = fr(rays, **kw)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        assert fh.stats() == fr.stats()
    # switching back re-uploads the unrounded tensors
    fh.factor_storage = "fp32"
    f32 = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    with torch.no_grad():
        assert torch.equal(fh(rays)[0], f32(rays)[0])


def test_g11_bf16_storage_vs_reference_on_rounded_factors(tiny, tiny_params, g11):
    rays = torch.from_numpy(tiny["tiny_rays"])
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    f.factor_storage = "bf16"
    with torch.no_grad():
        rgb, depth, z, w = f(rays, is_train=False, white_bg=True, N_samples=-1)
    close(rgb, g11["g11_eval_rgb"], atol=RGB_ATOL)
    close(depth, g11["g11_eval_depth"], atol=DEPTH_ATOL)
    close(w, g11["g11_eval_w"], atol=W_ATOL, rtol=W_RTOL)
    torch.manual_seed(123)
    with torch.no_grad():
        rgb, depth, z, w = f(rays, is_train=True, white_bg=True, N_samples=40)
    close(z, g11["g11_train_z"], atol=0)
    close(rgb, g11["g11_train_rgb"], atol=RGB_ATOL)
    close(w, g11["g11_train_w"], atol=W_ATOL, rtol=W_RTOL)


def test_g11_gradients_at_the_rounded_point(tiny, tiny_params, g11):
    rays = torch.from_numpy(tiny["tiny_rays"])
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    f.factor_storage = "bf16"
    torch.manual_seed(321)
    rgb, depth, z, w = f(rays, is_train=True, white_bg=True, N_samples=40)
    ca, cb, cw = [torch.from_numpy(g11[k]).to(dev()) for k in ("g11_ca", "g11_cb", "g11_cw")]
    ((rgb * ca).sum() + (depth * cb).sum() + (w * cw).sum()).backward()
    bad = {}
    for k, p in f.named_parameters():
        g = g11["g11_grad_" + k]
        err = float(np.abs(p.grad.detach().cpu().numpy() - g).max()) / (float(np.abs(g).max()) + 1e-12)
        if err > 2e-4:
            bad[k] = err
    assert not bad, f"gradient mismatch (max abs err / max |g|): {bad}"


def test_bf16_storage_full_frame_300():
    """300^3 production shape: bf16 storage == fp32 render of the rounded tensors bitwise on a strided 800x800 frame."""
    from text2nerf_amd import synth
    aabb = [[-8.0] * 3, [8.0] * 3]
    params = synth.make_field_params(0, [300] * 3, scene="S1-soft", aabb=aabb)
    rays = torch.from_numpy(synth.frame_rays_np(800, 800, stride=4))
    fh = make_field(params, [300] * 3, aabb, [0.5, 8.0])
    fh.factor_storage = "bf16"
    fr = make_field(_rounded(params), [300] * 3, aabb, [0.5, 8.0])
    with torch.no_grad():
        a = fh(rays)
        b = fr(rays)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert fh.stats() == fr.stats() and fh.stats()["appearance"] > 0
