"""Pin the plain-C oracle (oracle/oracle_c.c) against the golden vectors generated from the reference, and against the
PyTorch oracle on a 300^3 batch. CPU only."""
import numpy as np
import pytest

from oracle import oracle_torch as O
from oracle.oracle_c import COracle
from text2nerf_amd import synth
from tests.conftest import TINY


@pytest.fixture(scope="module")
def co(tiny_params):
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"])
    return COracle(cfg, tiny_params)


@pytest.mark.parametrize("tag,kw", [("eval", dict(is_train=False, white_bg=True, n_samples=-1)),
                                    ("eval70", dict(is_train=False, white_bg=True, n_samples=70)),
                                    ("evalblack", dict(is_train=False, white_bg=False, n_samples=-1))])
def test_c_oracle_forward_eval(tiny, co, tag, kw):
    rgb, depth, z, w = co.render(tiny["tiny_rays"], **kw)
    np.testing.assert_array_equal(z, np.broadcast_to(tiny[f"g6_{tag}_z"], z.shape))
    np.testing.assert_allclose(w, tiny[f"g6_{tag}_w"], atol=2e-6, rtol=2e-5)
    np.testing.assert_allclose(rgb, tiny[f"g6_{tag}_rgb"], atol=1e-5)
    np.testing.assert_allclose(depth, tiny[f"g6_{tag}_depth"], atol=5e-5)
    if tag != "eval70":
        assert co.last_stats["evaluated"] == int((tiny["g2_eval_valid"] & (tiny["g2_eval_pts"][..., 2] > 2.0)).sum())


def test_c_oracle_forward_train(tiny, co):
    rgb, depth, z, w = co.render(tiny["tiny_rays"], n_samples=40, is_train=True, jitter=tiny["g6_train_jitter"])
    np.testing.assert_array_equal(z, tiny["g6_train_z"])
    np.testing.assert_allclose(w, tiny["g6_train_w"], atol=2e-6, rtol=2e-5)
    np.testing.assert_allclose(rgb, tiny["g6_train_rgb"], atol=1e-5)
    np.testing.assert_allclose(depth, tiny["g6_train_depth"], atol=5e-5)


def test_c_oracle_sh_head(tiny, tiny_params_sh):
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"], shading_mode="SH")
    rgb, depth, _, _ = COracle(cfg, tiny_params_sh).render(tiny["tiny_rays"])
    np.testing.assert_allclose(rgb, tiny["g6_sh_rgb"], atol=1e-5)
    np.testing.assert_allclose(depth, tiny["g6_sh_depth"], atol=5e-5)


def test_c_oracle_big300(big300):
    aabb = [[-8.0] * 3, [8.0] * 3]
    cfg = O.FieldConfig(aabb=aabb, grid_size=[300] * 3)
    co = COracle(cfg, synth.make_field_params(0, [300] * 3, scene="S1-soft", aabb=aabb))
    rays = synth.frame_rays_np(800, 800)[big300["S1-soft_idx"]]
    rgb, depth, z, w = co.render(rays)
    np.testing.assert_allclose(rgb, big300["S1-soft_rgb"], atol=1e-5)
    np.testing.assert_allclose(depth, big300["S1-soft_depth"], atol=1e-4)
    np.testing.assert_allclose(w.sum(-1), big300["S1-soft_acc"], atol=2e-5)
    assert abs(co.last_stats["appearance"] - int(big300["S1-soft_napp"].sum())) <= 2
