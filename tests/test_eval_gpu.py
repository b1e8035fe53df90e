"""The drop-in evaluation loops on the MI355X: `evaluation` / `evaluation_path` (reference signatures, renderer.py:44-197) driving the
real HIP renderer with two views in flight on alternating streams, against the oracle's render + post-processing; and the trajectory
renderer with and without frames in flight (bitwise the same pictures)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import oracle_torch as O
from tests.conftest import TINY
from tests.test_hip_parity import dev, make_field
from text2nerf_amd import synth

pytestmark = pytest.mark.gpu
H, W = 24, 32


class _Dataset:
    def __init__(self, split, poses):
        self.split, self.img_wh, self.near_far = split, (W, H), TINY["near_far"]
        rays = torch.stack([torch.from_numpy(synth.frame_rays_np(H, W, c2w=p)) for p in poses])
        g = torch.Generator().manual_seed(1)
        self.all_rays_split = self.all_rays_gen_split = rays
        self.all_rays_sprt_split = rays[:2]
        self.all_rgbs_gen_split = torch.rand(len(poses), H * W, 3, generator=g)
        self.directions = torch.from_numpy(synth.ray_directions_np(H, W, float(W), float(W), W // 2, H // 2))
        self.focal = [float(W), float(W)]


def test_evaluation_loops_on_the_gpu(tmp_path, tiny_params):
    from text2nerf_amd import OctreeRender_trilinear_fast, evaluation, evaluation_path
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    poses = [synth.look_pose(0.2 * v - 0.3, -0.1, (0.2, 0.1, -1.0)) for v in range(4)]
    ds = _Dataset("train", poses)
    args = SimpleNamespace(batch_size=4096, push_depth=2.0)
    out = tmp_path / "views"
    psnrs = evaluation(ds, f, args, OctreeRender_trilinear_fast, str(out), N_vis=-1, prtx="t_", N_samples=-1, white_bg=True,
                       compute_extra_metrics=True, device=dev(), N_iter=3, preview=False)
    assert len(psnrs) == 4 and f.materialize_weights is True
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"])
    P = O.params_from_numpy(tiny_params)
    for v in range(4):
        o_rgb, o_depth, _, _ = O.forward(cfg, P, ds.all_rays_split[v])
        r8, d8, ps = O.postprocess_frame(o_rgb.numpy().reshape(H, W, 3), o_depth.numpy().reshape(H, W), TINY["near_far"], push_depth=2.0,
                                         gt_rgb=ds.all_rgbs_gen_split[v].view(H, W, 3).numpy())
        img = np.asarray(Image.open(out / "rgbs" / f"t_{v:03d}_rgb.png")).astype(np.int32)
        dep = np.asarray(Image.open(out / "depths" / f"t_{v:03d}_depth.png")).astype(np.int32)
        assert img.shape == (H, W, 3) and np.abs(img - r8.astype(np.int32)).max() <= 1          # truncation to uint8 of values 1e-7 apart
        assert np.mean(np.abs(dep - d8.astype(np.int32)) > 8) < 0.01                            # JET steps at 255 x depth boundaries
        assert abs(psnrs[v] - ps) < 1e-3
    # a camera path: rays from get_rays (HIP) per pose, rgb | depth side by side
    dt = _Dataset("test", poses)
    res = evaluation_path(dt, f, [p[:3] for p in poses[:3]], OctreeRender_trilinear_fast, str(tmp_path / "path"), prtx="p_", white_bg=True,
                          device=dev())
    assert res == [] and sorted(os.listdir(tmp_path / "path")) == ["p_000.png", "p_001.png", "p_002.png", "rgbd"]
    rgbd = np.asarray(Image.open(tmp_path / "path" / "rgbd" / "p_002.png"))
    assert rgbd.shape == (H, 2 * W, 3)


def test_frames_in_flight_render_the_same_pictures(tiny_params):
    from text2nerf_amd import render_views
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    poses = np.stack([synth.look_pose(0.15 * v, -0.05 * v, (0.2, 0.1, -1.0)) for v in range(7)])
    intr = [float(W), float(W), W // 2, H // 2]
    a_rgb, a_dep = render_views(f, poses, intr, H, W, frames_in_flight=1)
    for n in (2, 3):
        b_rgb, b_dep = render_views(f, poses, intr, H, W, frames_in_flight=n)
        assert torch.equal(a_rgb, b_rgb) and torch.equal(a_dep, b_dep), n
    # at full size too: the C2 frame, five views, two in flight
    aabb = [[-8.0] * 3, [8.0] * 3]
    big = make_field(synth.make_field_params(0, [300] * 3, scene="S1-soft", aabb=aabb), [300] * 3, aabb, [0.5, 8.0])
    p5 = np.stack([synth.look_pose(0.05 * v, 0.0, (0.0, 0.0, 0.0)) for v in range(5)])
    i8 = [800.0, 800.0, 400, 400]
    a_rgb, a_dep = render_views(big, p5, i8, 800, 800, frames_in_flight=1)
    b_rgb, b_dep = render_views(big, p5, i8, 800, 800, frames_in_flight=2)
    assert torch.equal(a_rgb, b_rgb) and torch.equal(a_dep, b_dep)


def test_parameter_upload_is_ordered_across_the_pipe_streams(tiny_params):
    """the lazy device upload (relayout + pack kernels on the caller's stream) of a field whose parameters just changed —
    the state evaluation() finds after training steps — must be complete before a view on ANOTHER stream reads the copies. Small frames
    (no host-side wait anywhere in the call), fresh parameters before every trajectory, pipelined against serial, repeated."""
    from text2nerf_amd import render_views, release_workspaces, workspace_reserved
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    poses = [synth.look_pose(0.15 * v - 0.3, 0.05, (0.1, 0.0, -1.2)) for v in range(4)]
    intr = [64.0, 64.0, 32, 32]
    g = torch.Generator().manual_seed(3)
    for trial in range(6):
        with torch.no_grad():      # an "optimiser step": every tensor changes in place (versions bump, the device copies are stale)
            for p in f.parameters():
                p.add_(0.01 * torch.randn(p.shape, generator=g).to(p.device))
        a_rgb, a_dep = render_views(f, poses, intr, 64, 64, frames_in_flight=2)
        torch.cuda.synchronize()
        b_rgb, b_dep = render_views(f, poses, intr, 64, 64, frames_in_flight=1)
        assert torch.equal(a_rgb, b_rgb) and torch.equal(a_dep, b_dep), f"trial {trial}: pipelined views differ from serial ones"
    # the pipes of repeated calls share their side streams: the scratch held for the device does not grow call by call
    held = workspace_reserved(dev())
    for _ in range(5):
        render_views(f, poses, intr, 64, 64, frames_in_flight=2)
    assert workspace_reserved(dev()) == held
    assert release_workspaces(dev(), keep_current=False) == held and workspace_reserved(dev()) == 0
