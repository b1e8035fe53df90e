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


def OctreeRender_trilinear_fast(rays, tensorf, chunk=4096, N_samples=-1, ndc_ray=False, white_bg=True, is_train=False,
                                device="cuda"):
    """Same signature and 5-tuple ``(rgb [R,3], None, depth [R], weights [R,N], z_vals [R,N])`` as renderer.py:28-42.

    Eval: the whole batch goes to the HIP renderer in one call (it sub-launches internally; ``chunk`` only bounds the
    reference's Python loop and is not needed here). Train: the reference draws one jitter vector per chunk from the CPU
    generator, so the chunk loop is kept to consume the RNG stream identically (the driver's batch is one chunk)."""
    if not is_train:
        rgb, depth, z, w = tensorf(rays, is_train=False, white_bg=white_bg, ndc_ray=ndc_ray, N_samples=N_samples)
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
    keep = tensorf.materialize_weights
    tensorf.materialize_weights = False            # evaluation discards weights / z_vals (renderer.py:89)
    rgbs, depths = [], []
    try:
        for c2w in poses:
            rays = generate_rays(H, W, intrinsic, c2w, device=dev)
            rgb, depth, _, _ = tensorf(rays, is_train=False, white_bg=white_bg, N_samples=N_samples)
            rgbs.append(rgb.clamp(0.0, 1.0).reshape(H, W, 3))
            depths.append(depth.reshape(H, W))
    finally:
        tensorf.materialize_weights = keep
    return torch.stack(rgbs), torch.stack(depths)
