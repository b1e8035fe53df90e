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


def get_ray_directions_blender(H, W, focal, center=None, device="cuda"):
    """dataLoader/ray_utils.py:45-63: the same pixel-centre grid in the OpenGL / Blender camera frame (y up, looking down -z): the
    OpenCV directions of the HIP kernel with the signs of y and z flipped (an exact operation)."""
    d = get_ray_directions(H, W, focal, center=center, device=device)
    return d * torch.tensor([1.0, -1.0, -1.0], device=d.device)


def _on_gpu(name, *ts):
    for t in ts:
        if t.device.type != "cuda":
            raise _lib.T2NError(f"{name}: the mirror computes on the MI355X only (tensor on {t.device})")


def depth2dist(z_vals, cos_angle):
    """dataLoader/ray_utils.py:9-15: sample spacings along a ray from its depths (the last one 1e10), times the ray's cosine."""
    _on_gpu("depth2dist", z_vals, cos_angle)
    gap = z_vals[..., 1:] - z_vals[..., :-1]
    far = torch.full_like(z_vals[..., :1], 1e10)
    return torch.cat([gap, far], -1) * cos_angle.unsqueeze(-1)


def ndc2dist(ndc_pts, cos_angle):
    """dataLoader/ray_utils.py:18-21: Euclidean spacings of consecutive NDC points [R, N, 3]; the last one is 1e10 cos."""
    _on_gpu("ndc2dist", ndc_pts, cos_angle)
    gap = (ndc_pts[:, 1:] - ndc_pts[:, :-1]).norm(dim=-1)
    return torch.cat([gap, 1e10 * cos_angle.unsqueeze(-1)], -1)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """dataLoader/ray_utils.py:129-171 (hierarchical sampling; no caller in the driver): inverse-CDF samples of the piecewise-constant
    density `weights` [B, M] over `bins` [B, M + 1]. det: evenly spaced quantiles; pytest: numpy's seed-0 draws, as there."""
    import numpy as np
    _on_gpu("sample_pdf", bins, weights)
    dev = weights.device
    w = weights + 1e-5
    cdf = torch.cumsum(w / w.sum(-1, keepdim=True), -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    shape = list(cdf.shape[:-1]) + [N_samples]
    if pytest:
        np.random.seed(0)
        u = torch.Tensor(np.broadcast_to(np.linspace(0.0, 1.0, N_samples), shape).copy() if det else np.random.rand(*shape)).to(dev)
    elif det:
        u = torch.linspace(0.0, 1.0, steps=N_samples, device=dev).expand(shape)
    else:
        u = torch.rand(shape, device=dev)
    u = u.contiguous()
    hi = torch.searchsorted(cdf.detach(), u, right=True)
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b_lo, b_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    span = c_hi - c_lo
    span = torch.where(span < 1e-5, torch.ones_like(span), span)
    return b_lo + (u - c_lo) / span * (b_hi - b_lo)


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


def _marcher(rays, N_samples, lindisp, perturb, bbox_3D, want_xyz=True):
    lib = _lib.load()
    # the reference draws the perturbation on the rays' own device generator (:217) — CPU rays consume the CPU stream
    pr = perturb * torch.rand((rays.shape[0], N_samples), device=rays.device) if perturb > 0 else None
    r = rays.contiguous().float()
    if r.device.type != "cuda":
        r = r.cuda()
    dev = r.device
    n = r.shape[0]
    steps = torch.linspace(0, 1, N_samples, device=dev)
    pr = pr.to(dev).float().contiguous() if pr is not None else None
    bb = None
    if bbox_3D is not None:
        b = torch.as_tensor(bbox_3D, dtype=torch.float32).detach().cpu().reshape(2, 3)
        bb = (C.c_float * 6)(*b.reshape(-1).tolist())
    xyz = torch.empty(n, N_samples, 3, device=dev) if want_xyz else None
    z = torch.empty(n, N_samples, device=dev)
    nf = torch.empty(n, 2, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.t2n_ray_marcher(_lib.ptr(r), n, r.shape[1], int(N_samples), 1 if lindisp else 0, bb, _lib.ptr(steps),
                                       _lib.ptr(pr), _lib.ptr(xyz), _lib.ptr(z), _lib.ptr(nf), _lib.current_stream_ptr(dev)),
                   "t2n_ray_marcher")
    return r, xyz, z, nf


def dda(rays_o, rays_d, bbox_3D):
    """dataLoader/ray_utils.py:174-181: slab intersection of rays with the box; returns (t_min [N,1], t_max [N,1])."""
    rays = torch.cat([rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)], 1)
    _, _, _, nf = _marcher(rays, 1, False, 0, bbox_3D, want_xyz=False)
    return nf[:, 0:1], nf[:, 1:2]


def ray_marcher(rays, N_samples=64, lindisp=False, perturb=0, bbox_3D=None):
    """dataLoader/ray_utils.py:184-228: N_samples evenly spaced between the box entry and exit of every ray (or between the
    ray's own near/far columns 6,7 without a box). Returns (xyz [N,S,3], rays_o, rays_d, z_vals [N,S])."""
    r, xyz, z, _ = _marcher(rays, N_samples, lindisp, perturb, bbox_3D)
    return xyz, r[:, 0:3], r[:, 3:6], z


def _ndc(H, W, focal, near, rays_o, rays_d, blender):
    lib = _lib.load()
    o = rays_o.reshape(-1, 3).contiguous().float()
    d = rays_d.reshape(-1, 3).contiguous().float()
    if o.device.type != "cuda":
        o, d = o.cuda(), d.cuda()
    oo, dd = torch.empty_like(o), torch.empty_like(d)
    with torch.cuda.device(o.device):
        _lib.check(lib.t2n_ndc_rays(int(H), int(W), float(focal), float(near), 1 if blender else 0, _lib.ptr(o), _lib.ptr(d), o.shape[0],
                                    _lib.ptr(oo), _lib.ptr(dd), _lib.current_stream_ptr(o.device)), "t2n_ndc_rays")
    return oo.reshape(rays_o.shape), dd.reshape(rays_d.shape)


def ndc_rays_blender(H, W, focal, near, rays_o, rays_d):
    """dataLoader/ray_utils.py:88-105 (the form `evaluation_path` applies when ndc_ray is set, renderer.py:162-163)."""
    return _ndc(H, W, focal, near, rays_o, rays_d, True)


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """dataLoader/ray_utils.py:107-124."""
    return _ndc(H, W, focal, near, rays_o, rays_d, False)
