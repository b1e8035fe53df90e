"""Deterministic synthetic fields, rays and poses for tests and bench.py.

Nothing here reads the reference or the network: every array is drawn from
``numpy.random.Generator(PCG64(seed))`` (stream-stable across machines), so the
golden generator in this container and the parity tests on the GPU box rebuild
bit-identical inputs from a seed instead of shipping 70 MB of weights.

Scenes follow SURVEY.md §8(d):
  * ``random``  – all factors ``scale * randn`` (tiny parity models),
  * ``S1-soft`` / ``S1-sharp`` – seeded random field + rank-1 "room shell"
    (walls at |x|=6, |y|=4, |z|=7; amplitude 10.6 / 30),
  * ``S2`` – "fog": density factors ``1.2 * randn``.

State-dict keys and logical shapes are those of the reference's
``TensorVMSplit`` (models/tensoRF.py:144-160, models/tensorBase.py:88-99).
"""
from __future__ import annotations

import math

import numpy as np

MAT_MODE = ((0, 1), (0, 2), (1, 2))   # models/tensorBase.py:190
VEC_MODE = (2, 1, 0)                  # models/tensorBase.py:191


def _randn(rng, shape, scale):
    return (rng.standard_normal(size=shape, dtype=np.float32) * np.float32(scale)).astype(np.float32)


def _linear(rng, fan_out, fan_in, zero_bias=False):
    bound = 1.0 / math.sqrt(fan_in)
    w = rng.uniform(-bound, bound, size=(fan_out, fan_in)).astype(np.float32)
    b = np.zeros(fan_out, np.float32) if zero_bias else rng.uniform(-bound, bound, size=(fan_out,)).astype(np.float32)
    return w, b


def mlp_in_channels(shading_mode: str, app_dim: int, fea_pe: int, view_pe: int = 6, pos_pe: int = 6) -> int:
    if shading_mode == "MLP_Fea_noview":
        return 2 * fea_pe * app_dim + app_dim
    if shading_mode == "MLP_Fea":
        return 2 * view_pe * 3 + 2 * fea_pe * app_dim + 3 + app_dim
    if shading_mode == "MLP_PE":
        return (3 + 2 * view_pe * 3) + (3 + 2 * pos_pe * 3) + app_dim
    if shading_mode == "MLP":
        return (3 + 2 * view_pe * 3) + app_dim
    return 0


def make_field_params(seed, grid_size, density_n_comp=(16, 16, 16), app_n_comp=(48, 48, 48), app_dim=27,
                      feature_c=128, fea_pe=6, shading_mode="MLP_Fea_noview", scene="random",
                      density_scale=0.1, app_scale=0.1, aabb=((-8., -8., -8.), (8., 8., 8.)), view_pe=6, pos_pe=6):
    """Return ``{state_dict key: float32 ndarray}`` for a TensorVMSplit field."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = [int(x) for x in grid_size]
    if scene == "S2":
        density_scale = 1.2
    sd = {}
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        v = VEC_MODE[i]
        sd[f"density_plane.{i}"] = _randn(rng, (1, density_n_comp[i], g[m1], g[m0]), density_scale)
        sd[f"density_line.{i}"] = _randn(rng, (1, density_n_comp[i], g[v], 1), density_scale)
    for i in range(3):
        m0, m1 = MAT_MODE[i]
        v = VEC_MODE[i]
        sd[f"app_plane.{i}"] = _randn(rng, (1, app_n_comp[i], g[m1], g[m0]), app_scale)
        sd[f"app_line.{i}"] = _randn(rng, (1, app_n_comp[i], g[v], 1), app_scale)
    sd["basis_mat.weight"], _ = _linear(rng, app_dim, int(sum(app_n_comp)))
    in_c = mlp_in_channels(shading_mode, app_dim, fea_pe, view_pe, pos_pe)
    if in_c:
        sd["renderModule.mlp.0.weight"], sd["renderModule.mlp.0.bias"] = _linear(rng, feature_c, in_c)
        sd["renderModule.mlp.2.weight"], sd["renderModule.mlp.2.bias"] = _linear(rng, feature_c, feature_c)
        sd["renderModule.mlp.4.weight"], sd["renderModule.mlp.4.bias"] = _linear(rng, 3, feature_c, zero_bias=True)

    if scene in ("S1-soft", "S1-sharp"):
        amp = np.float32(10.6 if scene == "S1-soft" else 30.0)
        lo, hi = np.asarray(aabb[0], np.float32), np.asarray(aabb[1], np.float32)
        axes = [np.linspace(lo[a], hi[a], g[a], dtype=np.float32) for a in range(3)]
        inside = [(np.abs(axes[0]) <= 6.15), (np.abs(axes[1]) <= 4.15), (np.abs(axes[2]) <= 7.15)]
        wall_at = {2: 7.0, 1: 4.0, 0: 6.0}
        for i in range(3):
            m0, m1 = MAT_MODE[i]
            v = VEC_MODE[i]
            sd[f"density_plane.{i}"][0, 0] = np.outer(inside[m1], inside[m0]).astype(np.float32)
            wall = (np.abs(np.abs(axes[v]) - np.float32(wall_at[v])) < 0.15).astype(np.float32)
            sd[f"density_line.{i}"][0, 0, :, 0] = amp * wall
    return sd


def ray_directions_np(H, W, fx, fy, cx, cy):
    """Camera-space unit directions (dataLoader/ray_utils.py:24-42 + scene_gen.py:45), float32."""
    xs = np.arange(W, dtype=np.float32) + np.float32(0.5)
    ys = np.arange(H, dtype=np.float32) + np.float32(0.5)
    i, j = np.meshgrid(xs, ys, indexing="xy")
    d = np.stack([(i - np.float32(cx)) / np.float32(fx), (j - np.float32(cy)) / np.float32(fy),
                  np.ones_like(i)], -1).astype(np.float32)
    n = np.sqrt((d * d).sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)
    return (d / n).astype(np.float32)


def frame_rays_np(H, W, c2w=None, stride=1):
    """[H*W/stride^2, 6] rays with SceneGen intrinsics (f=max(H,W), c=(W//2,H//2)); identity pose by default."""
    f = float(max(H, W))
    d = ray_directions_np(H, W, f, f, W // 2, H // 2)
    if stride > 1:
        d = d[::stride, ::stride]
    d = d.reshape(-1, 3)
    c2w = np.eye(4, dtype=np.float32)[:3] if c2w is None else np.asarray(c2w, np.float32)[:3]
    rd = (d @ c2w[:, :3].T).astype(np.float32)
    ro = np.broadcast_to(c2w[:, 3], rd.shape).astype(np.float32)
    return np.concatenate([ro, rd], 1).astype(np.float32)


def look_pose(yaw=0.0, pitch=0.0, center=(0., 0., 0.)):
    """Small OpenCV-convention c2w (x right, y down, z forward): yaw about y, then pitch about x."""
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], np.float64)
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]], np.float64)
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = (ry @ rx).astype(np.float32)
    m[:3, 3] = np.asarray(center, np.float32)
    return m


def local_fixed_like_poses(n=9, angle=0.2, rc=0.2):
    """9 small-baseline poses around the origin, shaped like scene_util.get_local_fixed_poses2's output
    (centre + 8 neighbours); used only as benchmark/test inputs (the pose generator itself is out of scope)."""
    poses = [look_pose()]
    k = 0
    while len(poses) < n:
        a = 2 * math.pi * k / (n - 1)
        poses.append(look_pose(yaw=angle * math.cos(a) * 0.6, pitch=angle * math.sin(a) * 0.2,
                               center=(rc * math.cos(a), rc * math.sin(a) * 0.5, 0.0)))
        k += 1
    return np.stack(poses)


def n_to_reso(n_voxels, aabb):
    """utils.py:292-296 restated: voxel edge from the box volume, floor per axis."""
    lo, hi = np.asarray(aabb[0], np.float32), np.asarray(aabb[1], np.float32)
    ext = (hi - lo).astype(np.float32)
    voxel = np.float32(np.float32(np.prod(ext, dtype=np.float32) / np.float32(n_voxels)) ** np.float32(1.0 / 3.0))
    return [int(x) for x in np.floor(ext / voxel)]


def cal_n_samples(reso, step_ratio=0.5):
    """utils.py:298-299."""
    return int(np.linalg.norm(reso) / step_ratio)


def concentrate_density(sd, lo_frac=(0.25, 0.3, 0.2), hi_frac=(0.7, 0.8, 0.65)):
    """Zero the density factors outside an index window per axis (in place on a state_dict of numpy arrays): every term
    of the density feature then vanishes outside a box, which gives updateAlphaMask / shrink tests a non-trivial bounding
    box."""
    mat_mode, vec_mode = ((0, 1), (0, 2), (1, 2)), (2, 1, 0)

    def window(n, ax):
        return int(round(lo_frac[ax] * (n - 1))), int(round(hi_frac[ax] * (n - 1)))

    for i in range(3):
        line = sd[f"density_line.{i}"]
        a, b = window(line.shape[2], vec_mode[i])
        line[:, :, :a] = 0
        line[:, :, b + 1:] = 0
        plane = sd[f"density_plane.{i}"]          # [1, C, g[m1], g[m0]]
        m0, m1 = mat_mode[i]
        a, b = window(plane.shape[2], m1)
        plane[:, :, :a] = 0
        plane[:, :, b + 1:] = 0
        a, b = window(plane.shape[3], m0)
        plane[:, :, :, :a] = 0
        plane[:, :, :, b + 1:] = 0
    return sd


def rgbd_frame(seed, H, W, n_boxes=3, holes=0):
    """Synthetic RGB-D view for the warp / depth-filter tests (numpy PCG64): a slanted background plane with `n_boxes`
    fronto-parallel boxes in front of it (depth discontinuities), smooth colour gradients plus per-object tints, fp32.
    `holes` > 0 zeroes that many random depth pixels (the filters treat depth == 0 as a discontinuity). Returns
    (rgb [H,W,3] in [0,1], depth [H,W] in ~[2,7])."""
    g = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    depth = (5.5 + 1.2 * (xx / W - 0.5) + 0.8 * (yy / H - 0.5)).astype(np.float32)
    depth += (0.01 * g.standard_normal((H, W))).astype(np.float32)
    rgb = np.stack([0.2 + 0.6 * xx / W, 0.3 + 0.5 * yy / H, 0.5 + 0.3 * np.sin(xx / 7.0) * np.cos(yy / 5.0)], -1).astype(np.float32)
    for _ in range(n_boxes):
        x0, y0 = int(g.integers(0, W - 8)), int(g.integers(0, H - 8))
        w, h = int(g.integers(6, max(7, W // 3))), int(g.integers(6, max(7, H // 3)))
        d = np.float32(g.uniform(2.2, 4.5))
        sl = (slice(y0, min(H, y0 + h)), slice(x0, min(W, x0 + w)))
        depth[sl] = d + (0.005 * g.standard_normal(depth[sl].shape)).astype(np.float32)
        rgb[sl] = (0.5 * rgb[sl] + 0.5 * g.uniform(0, 1, 3).astype(np.float32)).astype(np.float32)
    rgb = np.clip(rgb + (0.02 * g.standard_normal(rgb.shape)).astype(np.float32), 0, 1).astype(np.float32)
    for _ in range(holes):
        depth[int(g.integers(0, H)), int(g.integers(0, W))] = 0.0
    return rgb, depth


def align_inputs(seed, H):
    """Synthetic inputs of the global depth alignment (text2nerf_main.py:233-270; tests/golden/make_golden_align.py): a rendered depth
    map (float32, as the renderer returns it), a monocular estimate that is an affine function of (depth - push_depth) plus noise
    (float64, as `depth_est / 12000 + push_depth` is), and the validity map of the warped view with an unseen band."""
    g = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:H, 0:H].astype(np.float64) / H
    true = 3.0 + 1.5 * np.sin(3 * xx) * np.cos(2 * yy) + 0.8 * xx
    depth_rendered = (true + g.normal(0, 0.01, true.shape)).astype(np.float32)
    s, t = (0.8, 0.35) if seed % 2 == 0 else (1.3, -0.4)
    depth_est = (true - 2.0) / s + t + 2.0 + g.normal(0, 0.01, true.shape)
    my_map = (g.uniform(0, 1, true.shape) < 0.6).astype(np.float64)
    my_map[:, : H // 4] = 0.0
    return depth_rendered, depth_est, my_map
