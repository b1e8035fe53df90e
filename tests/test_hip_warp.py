"""f-3 on the MI355X: the HIP depth-aware median filter and DIBR forward warp against the goldens produced by the reference
(tests/golden/warp.npz) and against the oracle on larger seeded frames. Filter: bit-exact (outputs are copies of input
samples). Warp: fp64 atomics sum in a different order than numpy add.at -> depth within 1e-9 relative, uint8 image within
one level on < 0.1 % of the pixels (round-half-even ties), masks exact."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import oracle_warp as OW
from tests.conftest import GOLDEN
from text2nerf_amd import synth

sys.path.insert(0, GOLDEN)
from make_golden_warp_cases import FILTER_CASES, H, W, pose44, warp_poses  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gw():
    return dict(np.load(os.path.join(GOLDEN, "warp.npz"), allow_pickle=False))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_filter_vs_reference_golden_bit_exact(gw, tag):
    from text2nerf_amd.warp import sparse_bilateral_filtering
    c = FILTER_CASES[tag]
    rgb, depth = synth.rgbd_frame(c["seed"], H, W, holes=c["holes"])
    photos, depths = sparse_bilateral_filtering(depth.copy(), rgb.copy(), filter_size=c["filter_size"], depth_threshold=0.02,
                                                num_iter=c["num_iter"], HR=False, mask=None)
    assert len(photos) == len(depths) == c["num_iter"] and photos[0] is photos[-1]
    assert np.array_equal(depths[0], depth)
    assert np.array_equal(depths[1], gw[f"filt_{tag}_depth1"])
    assert np.array_equal(depths[-1], gw[f"filt_{tag}_depth"])
    assert np.array_equal(photos[-1], gw[f"filt_{tag}_photo"])


def test_filter_vs_oracle_larger_frame_and_tensor_io():
    from text2nerf_amd.warp import sparse_bilateral_filtering
    rgb, depth = synth.rgbd_frame(5, 97, 131, n_boxes=6, holes=15)        # ragged vs the 16x16 tiles
    o_photo, o_keep, o_states = OW.sparse_bilateral_filtering(depth, rgb, [7, 5, 5, 3, 3], 0.02, 5)
    dev = torch.device("cuda:0")
    photos, depths = sparse_bilateral_filtering(torch.from_numpy(depth).to(dev), torch.from_numpy(rgb).to(dev),
                                                filter_size=[7, 5, 5, 3, 3], depth_threshold=0.02, num_iter=5)
    assert isinstance(photos[-1], torch.Tensor) and photos[-1].is_cuda
    assert np.array_equal(photos[-1].cpu().numpy(), o_photo)
    for a, b in zip(depths, o_states):
        assert np.array_equal(a.cpu().numpy(), b)
    with pytest.raises(Exception):
        sparse_bilateral_filtering(depth, rgb, mask=np.ones_like(depth))


def _views(h, w, seed0=31):
    poses = [pose44(p) for p in warp_poses()]
    frames = [synth.rgbd_frame(seed0 + v, h, w) for v in range(3)]
    return poses, frames, [float(max(h, w)), float(max(h, w)), w // 2, h // 2]


def _check_warp(mask, img, dep, r_mask, r_img, r_dep):
    assert np.array_equal(np.asarray(mask), r_mask)
    assert img.dtype == np.float32 and dep.dtype == np.float64
    assert np.abs(img - r_img).max() <= 1.0 / 255 + 1e-7 and (img != r_img).mean() < 1e-3
    np.testing.assert_allclose(dep, r_dep, rtol=1e-9, atol=1e-12)


def test_multiview_warp_vs_reference_golden(gw):
    from text2nerf_amd.warp import bilinear_splat_warping_multiview
    poses, frames, intr = _views(H, W)
    mask, img, dep = bilinear_splat_warping_multiview([f[0] for f in frames], [f[1] for f in frames], np.stack(poses[:3]), poses[3],
                                                      H, W, intr, masks=None)
    _check_warp(mask, img, dep, gw["warp_mask"], gw["warp_image"], gw["warp_depth"])
    assert mask.dtype == np.int64 and 0.9 < mask.mean() <= 1.0


def test_multiview_warp_vs_oracle_512_with_masks():
    from text2nerf_amd.warp import bilinear_splat_warping_multiview
    h, w = 192, 256
    poses, frames, intr = _views(h, w, seed0=51)
    g = np.random.Generator(np.random.PCG64(9))
    masks = [g.uniform(0, 1, (h, w)) > 0.3 for _ in range(3)]
    args = ([f[0] for f in frames], [f[1] for f in frames], np.stack(poses[:3]), poses[3], h, w, intr)
    r = OW.bilinear_splat_warping_multiview(*args, masks=masks)
    o = bilinear_splat_warping_multiview(*args, masks=masks)
    _check_warp(o[0], o[1], o[2], r[0], r[1], r[2])


def test_hole_filling_vs_reference_golden_and_oracle(gw):
    from text2nerf_amd.warp import dibr_filter_mask2
    known = gw["fill_in_mask"].astype(np.int64)
    img, dep = gw["warp_image"].copy(), gw["warp_depth"].copy()
    holes = (known == 0) & (gw["warp_mask"] == 1)
    img[holes] = 1.0
    dep[holes] = 0.0
    f_img, f_map, f_dep = dibr_filter_mask2(img.copy(), known.copy(), output_depth=dep.copy())
    assert f_map.dtype == np.int64 and np.array_equal(f_map, gw["fill_mask"])
    assert np.array_equal(f_img, gw["fill_image"])
    np.testing.assert_allclose(f_dep, gw["fill_depth"], rtol=1e-13, atol=0)
    # larger frame, many holes, no depth: against the oracle
    rgb, d = synth.rgbd_frame(71, 150, 203, n_boxes=5)
    g = np.random.Generator(np.random.PCG64(72))
    k2 = (g.uniform(0, 1, d.shape) > 0.25).astype(np.int64)
    rgb[k2 == 0] = 1.0
    o_img, o_map = OW.dibr_filter_mask2(rgb, k2)
    h_img, h_map = dibr_filter_mask2(rgb.copy(), k2.copy())
    assert np.array_equal(h_map, o_map) and np.array_equal(h_img, o_img) and (o_map != k2).sum() > 3000


def test_four_stage_hole_filling_vs_reference_golden_and_oracle(gw):
    """dibr_filter_mask (utils.py:345-392) through t2n_dibr_filter_mask: the reference's goldens bit-exact, then a larger frame with
    many holes against the oracle."""
    import os
    from tests.test_oracle_warp import _fill1_inputs
    from text2nerf_amd.warp import dibr_filter_mask
    gf = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fill.npz"))
    for tag, (img, known) in _fill1_inputs(gw).items():
        f_img, f_map = dibr_filter_mask(img.copy(), known.copy())
        assert f_map.dtype == np.int64 and np.array_equal(f_map, gf[f"fill1_{tag}_mask"])
        assert np.array_equal(f_img, gf[f"fill1_{tag}_image"])
    rgb, d = synth.rgbd_frame(73, 131, 177, n_boxes=5)
    g = np.random.Generator(np.random.PCG64(74))
    k2 = (g.uniform(0, 1, d.shape) > 0.45).astype(np.int64)
    rgb[k2 == 0] = 1.0
    o_img, o_map = OW.dibr_filter_mask(rgb, k2)
    h_img, h_map = dibr_filter_mask(rgb.copy(), k2.copy())
    assert np.array_equal(h_map, o_map) and np.array_equal(h_img, o_img)
    assert (o_map != k2).sum() > 3000 and int((o_img == 255).all(-1).sum()) > 0
