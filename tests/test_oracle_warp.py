"""Pin oracle/oracle_warp.py (f-3: depth-aware median filtering, DIBR forward warp) against goldens produced by the
reference itself (tests/golden/make_golden_warp.py). CPU only."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle_warp as OW
from tests.conftest import GOLDEN
from text2nerf_amd import synth

sys.path.insert(0, GOLDEN)
from make_golden_warp_cases import FILTER_CASES, H, W, pose44, warp_poses  # noqa: E402


@pytest.fixture(scope="module")
def gw():
    return dict(np.load(os.path.join(GOLDEN, "warp.npz"), allow_pickle=False))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_sparse_bilateral_filtering_bit_exact(gw, tag):
    c = FILTER_CASES[tag]
    rgb, depth = synth.rgbd_frame(c["seed"], H, W, holes=c["holes"])
    photo, dkeep, states = OW.sparse_bilateral_filtering(depth, rgb, c["filter_size"], 0.02, c["num_iter"])
    assert np.array_equal(states[1], gw[f"filt_{tag}_depth1"])
    assert np.array_equal(dkeep, gw[f"filt_{tag}_depth"])
    assert np.array_equal(photo, gw[f"filt_{tag}_photo"])
    assert (dkeep != depth).sum() > 100          # the filter did something


def _views():
    poses = [pose44(p) for p in warp_poses()]
    frames = [synth.rgbd_frame(31 + v, H, W) for v in range(3)]
    intrinsic = [float(max(H, W)), float(max(H, W)), W // 2, H // 2]
    return poses, frames, intrinsic


def test_forward_warp_vs_reference(gw):
    poses, frames, intr = _views()
    K = np.eye(3, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = intr
    f8, known, dep, flow = OW.forward_warp((frames[1][0] * 255).astype(np.uint8), frames[1][1], np.linalg.inv(poses[1]),
                                           np.linalg.inv(poses[3]), K)
    np.testing.assert_allclose(flow, gw["fw_flow"], atol=1e-9)
    assert np.array_equal(known, gw["fw_mask"])
    assert np.abs(f8.astype(int) - gw["fw_frame"].astype(int)).max() <= 1 and (f8 != gw["fw_frame"]).mean() < 1e-3
    np.testing.assert_allclose(dep, gw["fw_depth"], rtol=1e-9, atol=1e-12)


def test_multiview_warp_vs_reference(gw):
    poses, frames, intr = _views()
    mask, img, dep = OW.bilinear_splat_warping_multiview([f[0] for f in frames], [f[1] for f in frames], np.stack(poses[:3]),
                                                         poses[3], H, W, intr)
    assert np.array_equal(mask, gw["warp_mask"])
    assert np.abs(img - gw["warp_image"]).max() <= 1.0 / 255 + 1e-7 and (img != gw["warp_image"]).mean() < 1e-3
    np.testing.assert_allclose(dep, gw["warp_depth"], rtol=1e-9, atol=1e-12)


def test_hole_filling_vs_reference(gw):
    known = gw["fill_in_mask"].astype(np.int64)
    img, dep = gw["warp_image"].copy(), gw["warp_depth"].copy()
    holes = (known == 0) & (gw["warp_mask"] == 1)
    img[holes] = 1.0
    dep[holes] = 0.0
    f_img, f_map, f_dep = OW.dibr_filter_mask2(img, known, dep)
    assert np.array_equal(f_map, gw["fill_mask"])
    assert np.array_equal(f_img, gw["fill_image"])
    np.testing.assert_allclose(f_dep, gw["fill_depth"], rtol=1e-13, atol=0)
    assert (f_map != known).sum() > 200


def _fill1_inputs(gw):
    """The inputs tests/golden/make_golden_fill.py fed the reference's dibr_filter_mask (rebuilt from warp.npz and a seed)."""
    known = gw["fill_in_mask"].astype(np.int64)
    img = gw["warp_image"].copy()
    img[(known == 0) & (gw["warp_mask"] == 1)] = 1.0
    g = np.random.Generator(np.random.PCG64(43))
    known2 = (g.uniform(0, 1, known.shape) < 0.55).astype(np.int64)
    img2 = gw["warp_image"].copy()
    img2[known2 == 0] = 1.0
    return {"a": (img, known), "b": (img2, known2)}


def test_four_stage_hole_filling_vs_reference(gw):
    """dibr_filter_mask (utils.py:345-392): 5x5 scan, 3x3 scan, border lines, erase scan — bit-exact against the reference's output."""
    gf = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fill.npz"))
    for tag, (img, known) in _fill1_inputs(gw).items():
        f_img, f_map = OW.dibr_filter_mask(img, known)
        assert np.array_equal(f_map, gf[f"fill1_{tag}_mask"])
        assert np.array_equal(f_img, gf[f"fill1_{tag}_image"])
    assert int((gf["fill1_b_image"] == 255).all(-1).sum()) >= 1      # the erase stage fires in case b
