"""Renderer harness with the reference's call surface (renderer.py:14-42)."""
from __future__ import annotations

import numpy as np
import torch


class SimpleSampler:
    """Permutation batcher (renderer.py:14-26); host-side, unchanged semantics."""

    def __init__(self, total, batch):
        self.total, self.batch = total, batch
        self.curr = total
        self.ids = None

    def nextids(self):
        self.curr += self.batch
        if self.curr + self.batch > self.total:
            self.ids = torch.LongTensor(np.random.permutation(self.total))
            self.curr = 0
        return self.ids[self.curr:self.curr + self.batch]


def detect_frame_width(rays, probe=8192):
    """Width W of a row-major pinhole raster in `rays` [R, >=6] (what `evaluation` passes: all rays of one image,
    renderer.py:85-89), or 0. Looks at the first `probe` rays on the host: one origin, directions that change smoothly along a row
    and jump at a row's end (consecutive direction deltas differ by more than half their size only there); the jump must repeat at
    2W - 1 when visible, the second row must start like the first, and R must be a multiple of W with at least 8 rows of at least 8
    pixels. The hint only selects the tile marcher (8x8-pixel tiles of coherent rays); a wrong guess costs speed, never correctness."""
    R = int(rays.shape[0])
    if R < 64 or rays.shape[1] < 6:
        return 0
    head = rays[: min(R, probe), :6].detach().float().cpu().numpy()
    o, d = head[:, :3], head[:, 3:6]
    if not np.all(np.abs(o - o[0]).max(axis=1) <= 1e-6 * (1.0 + np.abs(o[0]).max())):
        return 0
    delta = d[1:] - d[:-1]
    n = np.linalg.norm(delta, axis=1)
    if not np.isfinite(n).all() or n[0] <= 0:
        return 0
    jump = np.linalg.norm(delta[1:] - delta[:-1], axis=1) > 0.5 * np.maximum(n[:-1], 1e-30)
    ks = np.nonzero(jump)[0]
    if ks.size == 0:
        return 0
    W = int(ks[0]) + 2          # jump[k] compares delta[k+1] (the wrap d[k+2] - d[k+1]) with delta[k]: the row ends at index k + 1
    if W < 8 or R % W or R // W < 8:
        return 0
    if 2 * W <= head.shape[0]:  # the second row: starts like the first, ends with the same jump
        if np.linalg.norm(delta[W] - delta[0]) > 0.5 * n[0] or not jump[2 * W - 2]:
            return 0
    return W


def OctreeRender_trilinear_fast(rays, tensorf, chunk=4096, N_samples=-1, ndc_ray=False, white_bg=True, is_train=False,
                                device="cuda"):
    """Same signature and 5-tuple ``(rgb [R,3], None, depth [R], weights [R,N], z_vals [R,N])`` as renderer.py:28-42.

    Eval: the whole batch goes to the HIP renderer in one call (it sub-launches internally; ``chunk`` only bounds the
    reference's Python loop and is not needed here). Train: the reference draws one jitter vector per chunk from the CPU
    generator, so the chunk loop is kept to consume the RNG stream identically (the driver's batch is one chunk)."""
    if not is_train:
        # the driver's evaluation hands over all rays of one image: without an explicit hint the raster width is detected, so that
        # the frame takes the tile marcher (tensorf.auto_frame_width = False turns that off)
        keep_w = tensorf.frame_width
        if not keep_w and not ndc_ray and getattr(tensorf, "auto_frame_width", True):
            tensorf.frame_width = detect_frame_width(rays)
        try:
            rgb, depth, z, w = tensorf(rays, is_train=False, white_bg=white_bg, ndc_ray=ndc_ray, N_samples=N_samples)
        finally:
            tensorf.frame_width = keep_w
        return rgb, None, depth, w, z
    outs = [[], [], [], []]
    n = rays.shape[0]
    for k in range(n // chunk + int(n % chunk > 0)):
        o = tensorf(rays[k * chunk:(k + 1) * chunk], is_train=True, white_bg=white_bg, ndc_ray=ndc_ray,
                    N_samples=N_samples)
        for lst, t in zip(outs, o):
            lst.append(t)
    rgb, depth, z, w = [torch.cat(x) if x[0] is not None else None for x in outs]
    return rgb, None, depth, w, z


@torch.no_grad()
def render_views(tensorf, poses, intrinsic, H, W, N_samples=-1, white_bg=True):
    """Device-side counterpart of the per-view loop of ``evaluation`` / ``evaluation_path`` (renderer.py:85-93,160-170)
    without its file I/O: for every camera-to-world pose generate the [H*W,6] rays on the GPU
    (dataLoader/scene_gen.py:44-45,92-94), render the whole frame in one call and return ``rgb [V,H,W,3]`` (clamped) and
    ``depth [V,H,W]``. Nothing crosses PCIe per view. ``intrinsic`` = [fx, fy, cx, cy]."""
    from .ray_utils import generate_rays
    dev = tensorf.basis_mat.weight.device
    keep, keep_w = tensorf.materialize_weights, tensorf.frame_width
    tensorf.materialize_weights = False            # evaluation discards weights / z_vals (renderer.py:89)
    tensorf.frame_width = W                        # whole row-major frames: the 8x8-tile marcher applies
    rgbs, depths = [], []
    try:
        for c2w in poses:
            rays = generate_rays(H, W, intrinsic, c2w, device=dev)
            rgb, depth, _, _ = tensorf(rays, is_train=False, white_bg=white_bg, N_samples=N_samples)
            rgbs.append(rgb.clamp(0.0, 1.0).reshape(H, W, 3))
            depths.append(depth.reshape(H, W))
    finally:
        tensorf.materialize_weights, tensorf.frame_width = keep, keep_w
    return torch.stack(rgbs), torch.stack(depths)



@torch.no_grad()
def postprocess_frame(rgb, depth, near_far, push_depth=None, gt_rgb=None):
    """Device-side form of the per-view post-processing of ``evaluation`` (renderer.py:91-113) / ``evaluation_path``
    (:168-176) with ``visualize_depth_numpy`` (utils.py:241-257): returns ``(rgb8 [..,3] uint8, depth8 [..,3] uint8 JET in
    OpenCV's BGR order, psnr or None)`` as device tensors — one elementwise HIP kernel instead of a D2H copy + numpy per
    view. ``push_depth`` given: the ``evaluation`` form ``max(depth - push_depth + 0.8, 0)``; ``None``: ``evaluation_path``."""
    from . import _lib
    import ctypes as C
    lib = _lib.load()
    dev = rgb.device
    if dev.type != "cuda":
        raise _lib.T2NError("postprocess_frame runs on the MI355X only (no CPU fallback)")
    shape = tuple(depth.shape)
    rgb_c = rgb.reshape(-1, 3).contiguous().float()
    dep_c = depth.reshape(-1).contiguous().float()
    n = dep_c.numel()
    rgb8 = torch.empty(n, 3, dtype=torch.uint8, device=dev)
    dep8 = torch.empty(n, 3, dtype=torch.uint8, device=dev)
    sq = torch.zeros(1, dtype=torch.float64, device=dev) if gt_rgb is not None else None
    gt = gt_rgb.reshape(-1, 3).to(dev).contiguous().float() if gt_rgb is not None else None
    with torch.cuda.device(dev):
        _lib.check(lib.t2n_frame_postprocess(_lib.ptr(rgb_c), _lib.ptr(dep_c), n, float(push_depth or 0.0), 0.8,
                                             1 if push_depth is not None else 0, float(near_far[0]), float(near_far[1]),
                                             _lib.ptr(rgb8), _lib.ptr(dep8), _lib.ptr(gt), _lib.ptr(sq),
                                             _lib.current_stream_ptr(dev)), "t2n_frame_postprocess")
    psnr = None
    if sq is not None:
        loss = float(sq.item()) / (3 * n)
        psnr = -10.0 * float(np.log(loss)) / float(np.log(10.0))
    return rgb8.reshape(shape + (3,)), dep8.reshape(shape + (3,)), psnr


@torch.no_grad()
def evaluation_frames(tensorf, poses, intrinsic, H, W, near_far, N_samples=-1, white_bg=True, push_depth=2.0, gt_rgbs=None):
    """``evaluation`` (renderer.py:45-140) without its file I/O, entirely on the device: rays from ``(c2w, intrinsics)``,
    one render call per view, post-processing kernel; returns ``(rgb8 [V,H,W,3], depth8 [V,H,W,3], PSNRs)``."""
    rgbs, depths = render_views(tensorf, poses, intrinsic, H, W, N_samples=N_samples, white_bg=white_bg)
    out_rgb, out_dep, psnrs = [], [], []
    for v in range(rgbs.shape[0]):
        r8, d8, p = postprocess_frame(rgbs[v], depths[v], near_far, push_depth=push_depth,
                                      gt_rgb=None if gt_rgbs is None else gt_rgbs[v])
        out_rgb.append(r8)
        out_dep.append(d8)
        if p is not None:
            psnrs.append(p)
    return torch.stack(out_rgb), torch.stack(out_dep), psnrs
