"""f-2: frame post-processing (renderer.py:91-113, utils.py:241-257). CPU: properties of the oracle restatement (cv2 is not
installed here, so the JET table itself is unpinned — see oracle_torch.jet_table_bgr). GPU: the HIP kernel == the oracle,
byte for byte."""
import numpy as np
import pytest
import torch

from oracle import oracle_torch as O


def _frame(seed=0, n=5000):
    g = np.random.Generator(np.random.PCG64(seed))
    rgb = g.uniform(-0.1, 1.1, (n, 3)).astype(np.float32)
    depth = g.uniform(0.0, 9.5, (n,)).astype(np.float32)
    depth[:7] = [np.nan, np.inf, -np.inf, 0.0, 1.2, 2.0, 8.0]
    gt = g.uniform(0, 1, (n, 3)).astype(np.float32)
    return rgb, depth, gt


def test_jet_table_shape_and_landmarks():
    t = O.jet_table_bgr()
    assert t.shape == (256, 3) and t.dtype == np.uint8
    assert tuple(t[0]) == (128, 0, 0) and tuple(t[255]) == (0, 0, 128)          # dark blue -> dark red (B,G,R)
    assert t[96, 1] == 255 and t[96, 2] == 2 and t[31, 1] == 0 and t[32, 1] == 0 and t[33, 1] == 4
    assert t[:, 1].max() == 255 and t[128, 1] == 255
    # monotone ramps: blue falls after its plateau, red rises before its plateau
    assert np.all(np.diff(t[96:160, 0].astype(int)) <= 0) and np.all(np.diff(t[96:160, 2].astype(int)) >= 0)


def _jet_fixture():
    import os
    from tests.conftest import GOLDEN
    rows = [l.split() for l in open(os.path.join(GOLDEN, "jet_bgr_table.txt")) if l.strip() and not l.startswith("#")]
    return np.array(rows, dtype=np.uint8)


def test_jet_table_equals_committed_table():
    """tests/golden/jet_bgr_table.txt: the 256-entry COLORMAP_JET table (B, G, R), pinned to OpenCV's published definition
    (modules/imgproc/src/colormap.cpp, class Jet) — not to a cv2 run: see the file's header."""
    t = _jet_fixture()
    assert t.shape == (256, 3)
    assert np.array_equal(O.jet_table_bgr(), t)
    assert tuple(t[0]) == (128, 0, 0) and tuple(t[128]) == (126, 255, 130) and tuple(t[255]) == (0, 0, 128)


@pytest.mark.gpu
def test_hip_kernel_reproduces_the_jet_table_byte_exact():
    """every one of the 256 indices through the HIP post-processing kernel: depth values that map to index i exactly"""
    from text2nerf_amd import postprocess_frame
    dev = torch.device("cuda:0")
    mi, ma = 0.0, 255.0
    depth = torch.arange(256, dtype=torch.float32) + 0.25          # (d - mi) / (ma - mi + 1e-8) * 255 -> i + 0.25 -> index i
    rgb = torch.zeros(256, 3)
    _, d8, _ = postprocess_frame(rgb.to(dev).reshape(16, 16, 3), depth.to(dev).reshape(16, 16), [mi, ma], push_depth=None)
    assert np.array_equal(d8.cpu().numpy().reshape(256, 3), _jet_fixture())


def test_oracle_postprocess_semantics():
    rgb, depth, gt = _frame()
    r8, d8, psnr = O.postprocess_frame(rgb, depth, [0.5, 8.0], push_depth=2.0, gt_rgb=gt)
    assert r8.dtype == np.uint8 and r8.min() == 0 and r8.max() == 255
    assert r8[np.argmax(rgb[:, 0] > 1.0), 0] == 255
    # nan -> 0 -> below mi -> index 0 ; depth 2.0 -> 0.8 -> (0.8-0.5)/7.5*255 = 10.2 -> 10
    t = O.jet_table_bgr()
    assert tuple(d8[0]) == tuple(t[0]) and tuple(d8[5]) == tuple(t[10])
    assert 0 < psnr < 20
    r8b, d8b, none = O.postprocess_frame(rgb, depth, [0.5, 8.0], push_depth=None)
    assert none is None and np.array_equal(r8, r8b) and not np.array_equal(d8, d8b)


@pytest.mark.gpu
@pytest.mark.parametrize("push", [2.0, None])
def test_hip_postprocess_matches_oracle(push):
    from text2nerf_amd import postprocess_frame
    rgb, depth, gt = _frame(3, 64 * 48)
    dev = torch.device("cuda:0")
    r8, d8, psnr = postprocess_frame(torch.from_numpy(rgb).to(dev).reshape(64, 48, 3), torch.from_numpy(depth).to(dev).reshape(64, 48),
                                     [0.5, 8.0], push_depth=push, gt_rgb=torch.from_numpy(gt).reshape(64, 48, 3))
    o_r8, o_d8, o_psnr = O.postprocess_frame(rgb, depth, [0.5, 8.0], push_depth=push, gt_rgb=gt)
    assert r8.shape == (64, 48, 3) and d8.shape == (64, 48, 3)
    assert np.array_equal(r8.cpu().numpy().reshape(-1, 3), o_r8)
    assert np.array_equal(d8.cpu().numpy().reshape(-1, 3), o_d8)
    assert abs(psnr - o_psnr) < 1e-5


@pytest.mark.gpu
def test_evaluation_frames_on_device(tiny_params):
    from tests.conftest import TINY
    from tests.test_hip_parity import make_field
    from text2nerf_amd import evaluation_frames, synth
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    poses = [synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0)), synth.look_pose(-0.2, 0.1, (0.0, 0.0, -1.5))]
    H, W = 24, 32
    r8, d8, ps = evaluation_frames(f, poses, [32.0, 32.0, W // 2, H // 2], H, W, TINY["near_far"])
    assert r8.shape == (2, H, W, 3) and d8.shape == (2, H, W, 3) and r8.dtype == torch.uint8 and ps == []
    # against the oracle render + oracle post-processing of view 0
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"])
    rays = torch.from_numpy(synth.frame_rays_np(H, W, c2w=poses[0]))
    o_rgb, o_depth, _, _ = O.forward(cfg, O.params_from_numpy(tiny_params), rays)
    o_r8, o_d8, _ = O.postprocess_frame(o_rgb.numpy(), o_depth.numpy(), TINY["near_far"], push_depth=2.0)
    diff = np.abs(r8[0].cpu().numpy().reshape(-1, 3).astype(int) - o_r8.astype(int))
    assert diff.max() <= 1        # uint8 truncation of values within 1e-4
