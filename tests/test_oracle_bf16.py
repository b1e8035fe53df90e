"""Pin the oracle on bf16-rounded factor tensors against golden G11 (the reference itself run with its plane / line tensors
rounded to bf16, tests/golden/make_golden_bf16.py). CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_torch as O
from tests.conftest import GOLDEN, TINY


@pytest.fixture(scope="module")
def g11():
    return dict(np.load(os.path.join(GOLDEN, "bf16.npz"), allow_pickle=False))


def test_g11_oracle_on_bf16_rounded_factors(tiny, tiny_params, g11):
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"])
    P = O.round_factors_bf16(O.params_from_numpy(tiny_params))
    assert not torch.equal(P["density_plane.0"], O.params_from_numpy(tiny_params)["density_plane.0"])
    assert torch.equal(P["basis_mat.weight"], O.params_from_numpy(tiny_params)["basis_mat.weight"])
    rays = torch.from_numpy(tiny["tiny_rays"])
    rgb, depth, z, w = O.forward(cfg, P, rays, white_bg=True, is_train=False)
    np.testing.assert_allclose(rgb.numpy(), g11["g11_eval_rgb"], atol=5e-6)
    np.testing.assert_allclose(depth.numpy(), g11["g11_eval_depth"], atol=2e-5)
    np.testing.assert_allclose(w.numpy(), g11["g11_eval_w"], atol=2e-6, rtol=2e-5)
    # the rounding matters: the fp32 golden differs by far more than the tolerance
    assert float(np.abs(tiny["g6_eval_rgb"] - g11["g11_eval_rgb"]).max()) > 1e-4
    torch.manual_seed(123)
    jit = torch.rand(rays.shape[0], 1)
    rgb, depth, z, w = O.forward(cfg, P, rays, white_bg=True, is_train=True, n_samples=40, jitter=jit)
    assert np.array_equal(z.numpy(), g11["g11_train_z"])
    np.testing.assert_allclose(rgb.numpy(), g11["g11_train_rgb"], atol=5e-6)
    np.testing.assert_allclose(w.numpy(), g11["g11_train_w"], atol=2e-6, rtol=2e-5)
