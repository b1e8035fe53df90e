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
