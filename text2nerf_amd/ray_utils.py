"""Ray generation with the reference's function names (dataLoader/ray_utils.py:24-42,66-87), computed by HIP kernels."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _c2w_host(c2w):
    m = torch.as_tensor(c2w, dtype=torch.float32).detach().cpu()[:3, :4].contiguous()
    return (C.c_float * 12)(*m.reshape(-1).tolist())


def get_ray_directions(H, W, focal, center=None, device="cuda", normalize=False):
    """(H, W, 3) camera-space directions for pixel centres; ``normalize=True`` fuses scene_gen.py:45's division."""
    lib = _lib.load()
    cent = center if center is not None else [W / 2, H / 2]
    out = torch.empty(H, W, 3, device=device, dtype=torch.float32)
    with torch.cuda.device(out.device):
        _lib.check(lib.t2n_ray_directions(H, W, float(focal[0]), float(focal[1]), float(cent[0]), float(cent[1]),
                                          1 if normalize else 0, _lib.ptr(out), _lib.current_stream_ptr(out.device)),
                   "t2n_ray_directions")
    return out


def get_rays(directions, c2w):
    """rays_o, rays_d ([H*W,3] each) in world coordinates; directions are rotated, not re-normalised."""
    lib = _lib.load()
    d = directions.reshape(-1, 3).contiguous().float()
    if d.device.type != "cuda":
        d = d.cuda()
    n = d.shape[0]
    ro, rd = torch.empty_like(d), torch.empty_like(d)
    with torch.cuda.device(d.device):
        _lib.check(lib.t2n_get_rays(_lib.ptr(d), n, _c2w_host(c2w), _lib.ptr(ro), _lib.ptr(rd), None,
                                    _lib.current_stream_ptr(d.device)), "t2n_get_rays")
    return ro, rd


def generate_rays(H, W, intrinsic, c2w, device="cuda"):
    """Fused SceneGen recipe (scene_gen.py:44-45,92-94): [H*W,6] rows (o, d) straight from (intrinsics, pose)."""
    lib = _lib.load()
    fx, fy, cx, cy = [float(v) for v in intrinsic]
    out = torch.empty(H * W, 6, device=device, dtype=torch.float32)
    with torch.cuda.device(out.device):
        _lib.check(lib.t2n_generate_rays(H, W, fx, fy, cx, cy, _c2w_host(c2w), _lib.ptr(out),
                                         _lib.current_stream_ptr(out.device)), "t2n_generate_rays")
    return out
