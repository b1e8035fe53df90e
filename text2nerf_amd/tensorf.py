"""Host-side mirror of the reference's radiance-field interface for the VM-split path.

``TensorVMSplit(aabb, gridSize, device, **kargs)`` keeps the reference call surface (models/tensoRF.py:139-239,
models/tensorBase.py:163-507): same constructor keywords, ``forward`` signature and return tuple, parameter names /
logical shapes (``state_dict`` interchange with reference ``.th`` checkpoints), ``get_optparam_groups``, ``TV_loss_*``,
``filtering_rays``, ``save`` / ``load`` / ``get_kwargs``.  All arithmetic of the ray-marching path runs in
libt2n_hip.so (hand-written gfx950 kernels) through the C-ABI in include/t2n.h; torch is used for device memory,
streams and autograd plumbing only.  There is no CPU execution path: calling the renderer without the HIP library or
a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import time
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import FLAG_ADD_BG, FLAG_COHERENT, FLAG_DEVICE_ROWS, FLAG_KEEP_CTX, FLAG_NDC, FLAG_TRAIN, T2NError

MAT_MODE = [[0, 1], [0, 2], [1, 2]]
VEC_MODE = [2, 1, 0]

# train_step seeds the TV terms on a side stream (two more library calls and two stream hand-overs per step) only for batches whose step is
# GPU-bound; below, the TV pass stays inside the optimiser step
_SEED_MIN_RAYS = 8192
_WORKSPACE = {}


def _ws_key(device):
    device = torch.device(device)
    if device.type != "cuda":
        return (str(device), 0)
    return (str(device), int(torch.cuda.current_stream(device).cuda_stream))


_WS_IDLE = {}            # (device, stream) -> consecutive requests that used less than a quarter of the buffer
_WS_SHRINK_AFTER = 64    # ... after which the buffer is given back (a training run whose appearance lists tripled once and shrank again)


def workspace(device, nbytes: int) -> torch.Tensor:
    """Scratch buffer handed to the C-ABI calls, one per (device, stream): calls queued on one stream execute in order and may
    share it (every call is done with it when it returns control to the stream); fields driven on DIFFERENT streams get different
    buffers, so a render on one stream never aliases the backward scratch of another. It grows when a call needs more and is given
    back when _WS_SHRINK_AFTER consecutive requests used less than a quarter of it. Either way the old buffer is dropped BEFORE the
    new one is allocated and torch's cached blocks go back to the driver: a buffer that grew in steps would otherwise leave every
    earlier size reserved in the caching allocator (54 GiB reserved for 9.5 GiB in use, profiles/round4_train_soak.txt)."""
    key = _ws_key(device)
    buf = _WORKSPACE.get(key)
    nbytes = int(nbytes)
    if buf is not None and buf.numel() >= nbytes:
        if nbytes * 4 < buf.numel() and buf.numel() > (256 << 20):
            _WS_IDLE[key] = _WS_IDLE.get(key, 0) + 1
            if _WS_IDLE[key] < _WS_SHRINK_AFTER:
                return buf
        else:
            _WS_IDLE[key] = 0
            return buf
    _WS_IDLE[key] = 0
    had = buf is not None
    _WORKSPACE.pop(key, None)
    del buf
    if had and torch.device(device).type == "cuda":
        torch.cuda.empty_cache()      # (rare: sizes move geometrically)
    _WORKSPACE[key] = buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return buf


def _ladder(n: int, step: float = 1.25, floor: int = 1024) -> int:
    """n rounded up to the next rung of a geometric ladder: sizes derived from running counts change rarely, and a caching allocator
    sees the same few block sizes again and again."""
    v = floor
    while v < n:
        v = int(v * step) + 1
    return v


def release_workspaces(device=None, keep_current=True) -> int:
    """Drop the scratch buffers of `device` (all devices when None) — all of them, or all but the current stream's. Returns the bytes
    released to torch's caching allocator. For long runs that drove a field from many streams; a pipelined evaluation keeps its
    side streams (renderer._FramePipe) and needs no call."""
    freed = 0
    for key in list(_WORKSPACE):
        d, st = key
        if device is not None and d != str(torch.device(device)):
            continue
        if keep_current and key == _ws_key(d):
            continue
        freed += _WORKSPACE.pop(key).numel()
    return freed


def workspace_reserved(device) -> int:
    """Bytes of scratch currently held for `device` (all streams)."""
    d = str(torch.device(device))
    return sum(b.numel() for (k, _), b in _WORKSPACE.items() if k == d)


class _PinnedRing:
    """Host -> device copies that do not stall the stream. A pageable-source hipMemcpy waits for ALL queued GPU work before it
    starts (and blocks the host), which in a training loop idles the GPU for the rest of the host-side batch preparation
    (~0.5 ms per iteration measured: rays, jitter and targets arrive on the CPU exactly like in the reference,
    renderer.py:33 / text2nerf_main.py:550-553). Staging through a small ring of pinned buffers makes the copies truly
    asynchronous; a slot is reused only after the event recorded behind its last copy has completed."""

    SLOTS = 8

    def __init__(self):
        self.bufs = [None] * self.SLOTS
        self.events = [None] * self.SLOTS
        self.next = 0

    def copy(self, t: torch.Tensor, device) -> torch.Tensor:
        if t.device.type != "cpu" or t.numel() == 0:
            return t.to(device)
        if t.is_pinned():
            return t.to(device, non_blocking=True)
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        i = self.next
        self.next = (i + 1) % self.SLOTS
        if self.events[i] is not None:
            self.events[i].synchronize()
        if self.bufs[i] is None or self.bufs[i].numel() < nbytes:
            self.bufs[i] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
        stage = self.bufs[i][:nbytes].view(t.dtype).view(t.shape)
        stage.copy_(t)
        out = stage.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self.events[i] = ev
        return out


    def copy_many(self, ts, device):
        """Several CPU tensors through ONE staging slot and ONE host -> device copy (a training step moves rays, jitter and two
        target tensors: four copies, four event records otherwise). Returns device tensors shaped like the inputs (256-B aligned
        slices of one buffer)."""
        offs, total = [], 0
        for t in ts:
            offs.append(total)
            total += (t.numel() * t.element_size() + 255) // 256 * 256
        i = self.next
        self.next = (i + 1) % self.SLOTS
        if self.events[i] is not None:
            self.events[i].synchronize()
        if self.bufs[i] is None or self.bufs[i].numel() < total:
            self.bufs[i] = torch.empty(max(total, 1 << 20), dtype=torch.uint8).pin_memory()
        stage = self.bufs[i][:total]
        for t, o in zip(ts, offs):
            stage[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape).copy_(t)
        dev_buf = stage.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self.events[i] = ev
        return [dev_buf[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape) for t, o in zip(ts, offs)]


_RING = {}


def to_device_async_many(ts, device):
    """``[t.to(device) for t in ts]`` with the CPU tensors among them staged together (see _PinnedRing.copy_many)."""
    device = torch.device(device)
    cpu = [k for k, t in enumerate(ts) if isinstance(t, torch.Tensor) and t.device.type == "cpu" and t.numel() and not t.is_pinned()]
    if device.type != "cuda" or len(cpu) < 2:
        return [to_device_async(t, device) for t in ts]
    ring = _RING.get(str(device))
    if ring is None:
        ring = _RING[str(device)] = _PinnedRing()
    out = [None if k in cpu else to_device_async(t, device) for k, t in enumerate(ts)]
    with torch.cuda.device(device):
        moved = ring.copy_many([ts[k].contiguous() for k in cpu], device)
    for k, m in zip(cpu, moved):
        out[k] = m
    return out


def to_device_async(t: torch.Tensor, device) -> torch.Tensor:
    """``t.to(device)`` for CPU tensors through a pinned staging ring (see _PinnedRing); device tensors pass through."""
    device = torch.device(device)
    if device.type != "cuda" or not isinstance(t, torch.Tensor) or t.device.type != "cpu":
        return t.to(device)
    key = str(device)
    ring = _RING.get(key)
    if ring is None:
        ring = _RING[key] = _PinnedRing()
    with torch.cuda.device(device):
        return ring.copy(t, device)


def workspace_budget(device=None) -> int:
    """Largest scratch buffer a render call may take (a call needing more is split into sub-launches by the library).
    Default: an eighth of the device's memory, at most 32 GiB — on a 288 GB MI355X a whole 800x800 x 518-sample frame
    runs as ONE launch even with worst-case lists (16 GiB; image-ordered frames ask for budgeted lists instead: ~6 GiB for the first
    frame, ~4.7 GiB from then on, see t2n_render_workspace_bytes_hint); T2N_WORKSPACE_GIB overrides."""
    env = os.environ.get("T2N_WORKSPACE_GIB")
    if env is not None:
        return int(float(env) * (1 << 30))
    total = torch.cuda.get_device_properties(device).total_memory if device is not None and torch.cuda.is_available() else 64 << 30
    return int(min(32 << 30, total // 8))


def raw2alpha(sigma: torch.Tensor, dist: torch.Tensor):
    """models/tensorBase.py:19-26 on the GPU: ``(alpha, weights, T[:, -1:])``."""
    lib = _lib.load()
    sigma = sigma.contiguous().float()
    dist = dist.contiguous().float()
    R, N = sigma.shape
    alpha = torch.empty_like(sigma)
    w = torch.empty_like(sigma)
    bg = torch.empty(R, 1, device=sigma.device, dtype=torch.float32)
    _lib.check(lib.t2n_raw2alpha(_lib.ptr(sigma), _lib.ptr(dist), R, N, _lib.ptr(alpha), _lib.ptr(w), _lib.ptr(bg),
                                 _lib.current_stream_ptr(sigma.device)), "t2n_raw2alpha")
    return alpha, w, bg


def positional_encoding(positions, freqs):
    """models/tensorBase.py:11-17: ``[sin(p 2^k) ..., cos(p 2^k) ...]`` with the sines of all (channel, octave) pairs first (channel-major,
    octave-minor), then the cosines. The render path never calls this: the head kernels encode in registers (csrc/t2n_mlp_ss.hip,
    t2n_shade.hip); it is here for callers of the reference's module-level name."""
    if positions.device.type != "cuda":
        raise T2NError("positional_encoding: the mirror computes on the MI355X only")
    octaves = torch.pow(2.0, torch.arange(freqs, device=positions.device, dtype=torch.float32))
    arg = (positions.unsqueeze(-1) * octaves).flatten(-2)
    return torch.cat([arg.sin(), arg.cos()], dim=-1)


def SHRender(xyz_sampled, viewdirs, features):
    """models/tensorBase.py:29-33: degree-2 spherical-harmonics colour, ``relu(sum_b Y_b(dir) f[c, b] + 0.5)``; the basis comes from the
    HIP kernel behind ``eval_sh_bases``. (Inside a render the SH head runs in the shade kernel, csrc/t2n_heads.hip.)"""
    from .sh import eval_sh_bases
    basis = eval_sh_bases(2, viewdirs)
    coeff = features.reshape(-1, 3, basis.shape[-1])
    return torch.relu((coeff * basis[:, None]).sum(-1) + 0.5)


def RGBRender(xyz_sampled, viewdirs, features):
    """models/tensorBase.py:36-39: the features are the colour."""
    return features


class AlphaGridMask(nn.Module):
    """Occupancy volume with the reference's attributes (models/tensorBase.py:41-59). ``sample_alpha`` runs the trilinear
    lookup on the GPU through the owning field (t2n_alpha_at); the render kernels apply the mask themselves."""

    def __init__(self, device, aabb, alpha_volume):
        super().__init__()
        self.device = device
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32).to(device)
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invgridSize = 1.0 / self.aabbSize * 2
        self.alpha_volume = alpha_volume.view(1, 1, *alpha_volume.shape[-3:]).float().to(device)
        self.gridSize = torch.LongTensor([alpha_volume.shape[-1], alpha_volume.shape[-2], alpha_volume.shape[-3]]).to(device)
        object.__setattr__(self, "_field", None)

    def normalize_coord(self, xyz_sampled):
        return (xyz_sampled - self.aabb[0]) * self.invgridSize - 1

    def sample_alpha(self, xyz_sampled):
        if self._field is None:
            raise T2NError("AlphaGridMask.sample_alpha needs the mask to be attached to a TensorVMSplit (tensorf.alphaMask = mask)")
        return self._field._alpha_at(xyz_sampled)


class MLPRender_Fea_noview(nn.Module):
    """Parameter container with the reference's key names (models/tensorBase.py:88-99). The arithmetic of
    ``forward`` (:101-109) lives in the fused HIP shade kernel; it is reached through TensorVMSplit."""

    def __init__(self, inChanel, feape=6, featureC=128):
        super().__init__()
        self.in_mlpC = 2 * feape * inChanel + inChanel
        self.feape = feape
        l0, l1, l2 = nn.Linear(self.in_mlpC, featureC), nn.Linear(featureC, featureC), nn.Linear(featureC, 3)
        self.mlp = nn.Sequential(l0, nn.ReLU(inplace=True), l1, nn.ReLU(inplace=True), l2)
        nn.init.constant_(self.mlp[-1].bias, 0)

    def forward(self, pts, viewdirs, features):
        raise T2NError("MLPRender_Fea_noview runs fused inside TensorVMSplit's HIP shade kernel; call "
                       "TensorVMSplit.shade(xyz_norm) / forward(rays) instead")


class _ViewHead(nn.Module):
    """Parameter containers of the view-dependent heads with the reference's key names; their arithmetic runs on the general
    head path of the HIP library (csrc/t2n_heads.hip), reached through TensorVMSplit.forward."""

    def _build(self, in_mlpC, featureC):
        self.in_mlpC = in_mlpC
        l0, l1, l2 = nn.Linear(in_mlpC, featureC), nn.Linear(featureC, featureC), nn.Linear(featureC, 3)
        self.mlp = nn.Sequential(l0, nn.ReLU(inplace=True), l1, nn.ReLU(inplace=True), l2)
        nn.init.constant_(self.mlp[-1].bias, 0)

    def forward(self, pts, viewdirs, features):
        raise T2NError(f"{type(self).__name__} runs inside TensorVMSplit's HIP render call; use TensorVMSplit.forward(rays)")


class MLPRender_Fea(_ViewHead):
    """models/tensorBase.py:62-86: [features, viewdirs, PE(features, feape), PE(viewdirs, viewpe)] -> 3-layer MLP -> sigmoid."""

    def __init__(self, inChanel, viewpe=6, feape=6, featureC=128):
        super().__init__()
        self.viewpe, self.feape = viewpe, feape
        self._build(2 * viewpe * 3 + 2 * feape * inChanel + 3 + inChanel, featureC)


class MLPRender_PE(_ViewHead):
    """models/tensorBase.py:111-135: [features, viewdirs, PE(pts, pospe), PE(viewdirs, viewpe)]."""

    def __init__(self, inChanel, viewpe=6, pospe=6, featureC=128):
        super().__init__()
        self.viewpe, self.pospe = viewpe, pospe
        self._build((3 + 2 * viewpe * 3) + (3 + 2 * pospe * 3) + inChanel, featureC)


class MLPRender(_ViewHead):
    """models/tensorBase.py:137-159: [features, viewdirs, PE(viewdirs, viewpe)]."""

    def __init__(self, inChanel, viewpe=6, featureC=128):
        super().__init__()
        self.viewpe = viewpe
        self._build((3 + 2 * viewpe * 3) + inChanel, featureC)


class TensorBase(nn.Module):
    """models/tensorBase.py:163-507: what every field of the reference shares — constructor keywords, step size, ray sampling, the
    render call, checkpoints, occupancy-mask maintenance — on the HIP kernels. The factor containers and what depends on their
    layout (init_svd_volume, get_optparam_groups, the TV terms, compute_densityfeature / compute_appfeature, up-sampling, shrink)
    come from the subclass, as in the reference; the kernels read VM-split factor tensors (TensorVMSplit, and TensorVM / TensorCP
    through views / an exact embedding)."""

    def __init__(self, aabb, gridSize, device, density_n_comp=8, appearance_n_comp=24, app_dim=27,
                 shadingMode="MLP_PE", alphaMask=None, near_far=[2.0, 6.0], density_shift=-10, alphaMask_thres=0.001,
                 distance_scale=25, rayMarch_weight_thres=0.0001, pos_pe=6, view_pe=6, fea_pe=6, featureC=128,
                 step_ratio=2.0, fea2denseAct="softplus"):
        super().__init__()
        if isinstance(density_n_comp, int):
            density_n_comp = [density_n_comp] * 3
        if isinstance(appearance_n_comp, int):
            appearance_n_comp = [appearance_n_comp] * 3
        self.density_n_comp = list(density_n_comp)
        self.app_n_comp = list(appearance_n_comp)
        self.app_dim = app_dim
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32).to(device)
        self.alphaMask = alphaMask
        self.device = device
        self.density_shift = density_shift
        self.alphaMask_thres = alphaMask_thres
        self.distance_scale = distance_scale
        self.rayMarch_weight_thres = rayMarch_weight_thres
        self.fea2denseAct = fea2denseAct
        self.near_far = near_far
        self.step_ratio = step_ratio
        self.matMode, self.vecMode, self.comp_w = MAT_MODE, VEC_MODE, [1, 1, 1]
        self.shadingMode, self.pos_pe, self.view_pe, self.fea_pe, self.featureC = shadingMode, pos_pe, view_pe, fea_pe, featureC
        self.keep_activation_rows = True  # training: the forward keeps the MLP activations of the appearance samples for the backward
        self.defer_factor_grads = False   # set by optim.TVAdam(field=self): plane / line gradients stay on the device (channel-last)
        self.materialize_weights = True   # the reference always returns weights/z_vals; set False to skip 8*N B/ray
        self.z_gate = 2.0                 # models/tensorBase.py:460
        self.frame_width = 0              # set to the image width when eval rays are whole row-major frames: enables the
                                          # 8x8-tile marcher (shared dot-product tables); 0 = unknown -> per-ray marcher
        self.mlp_exact_fp32 = False       # False: f16 two-way-split MFMA products; True: exact fp32 MFMA (forward and backward)
        # 'fp32' (default) or 'bf16' (BASELINE configs[4]): the forward gathers read bf16 copies of the 12 factor tensors;
        # the render equals the fp32 render of the bf16-rounded tensors bit for bit, the parameters stay fp32 masters
        self.factor_storage = "fp32"
        # Early ray termination of eval renders that return neither weights nor z_vals (evaluation(), render_views, ...): OPT-IN. The
        # reference never terminates (models/tensorBase.py:19-26,494-505: every in-box sample is evaluated), so the default 0.0 is its
        # sample-for-sample arithmetic (exact evaluated-sample counts). A threshold eps > 0 stops a ray whose transmittance fell below it:
        # a bounded deviation (< eps on acc and colour, < eps * z range on depth). Values above rayMarch_weight_thres are clamped to it
        # (a skipped sample could otherwise have been an appearance sample). OctreeRender_trilinear_fast (weights returned) and
        # training are never terminated.
        self.early_termination = 0.0
        self._handle = None
        self._uploaded_key = None
        self._gbuf = None
        self._gbuf_dirty = False
        self._gbuf_stale = False
        self.last_stats = None

        if shadingMode not in _lib.SHADE_IDS:
            raise T2NError(f"shadingMode {shadingMode!r} is not implemented by the HIP renderer "
                           f"(supported: {sorted(_lib.SHADE_IDS)})")
        self.update_stepSize(gridSize)
        self.init_svd_volume(gridSize[0], device)
        self.init_render_func(shadingMode, pos_pe, view_pe, fea_pe, featureC, device)
        self._is_general()        # shapes beyond the general path's capacity fail here, loudly

    def init_render_func(self, shadingMode, pos_pe, view_pe, fea_pe, featureC, device):
        """models/tensorBase.py:200-217: the head's parameter container (its arithmetic runs in the shade kernels)."""
        if shadingMode == "MLP_Fea_noview":
            self.renderModule = MLPRender_Fea_noview(self.app_dim, fea_pe, featureC).to(device)
        elif shadingMode == "MLP_PE":        # models/tensorBase.py:201-208
            raise T2NError("shadingMode 'MLP_PE': the reference's MLPRender_PE sizes its first layer for 3 inputs it never "
                           "concatenates (models/tensorBase.py:115 vs :126-131) and fails with a shape error for every "
                           "pos_pe / view_pe; there is no behaviour to reproduce")
        elif shadingMode == "MLP_Fea":
            self.renderModule = MLPRender_Fea(self.app_dim, view_pe, fea_pe, featureC).to(device)
        elif shadingMode == "MLP":
            self.renderModule = MLPRender(self.app_dim, view_pe, featureC).to(device)
        elif shadingMode == "SH":
            self.renderModule = SHRender
        elif shadingMode == "RGB":
            assert self.app_dim == 3
            self.renderModule = RGBRender
        else:
            raise T2NError(f"Unrecognized shading module {shadingMode!r}")

    # ---- sampling as a stage of its own (models/tensorBase.py:293-323); forward() samples inside the march kernels ----------------
    def _sample(self, rays_o, rays_d, N, jitter, ndc):
        lib = _lib.load()
        h = self.sync_params()
        dev = self.basis_mat.weight.device if hasattr(self, "basis_mat") else self._all_params()[0].device
        o = rays_o.detach().reshape(-1, 3).contiguous().float().to(dev)
        d = rays_d.detach().reshape(-1, 3).contiguous().float().to(dev)
        n = o.shape[0]
        pts = torch.empty(n, N, 3, device=dev, dtype=torch.float32)
        z = None if ndc else torch.empty(n, N, device=dev, dtype=torch.float32)
        valid = torch.empty(n, N, device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            _lib.check(lib.t2n_sample_ray(h, _lib.ptr(o), _lib.ptr(d), n, int(N), _lib.ptr(jitter), 1 if ndc else 0, _lib.ptr(pts),
                                          _lib.ptr(z), _lib.ptr(valid), _lib.current_stream_ptr(dev)), "t2n_sample_ray")
        return pts, z, valid.bool()

    def sample_ray(self, rays_o, rays_d, is_train=True, N_samples=-1):
        """models/tensorBase.py:304-323: ``(rays_pts [R,N,3], interpx [R,N], ~mask_outbbox [R,N])``. Train mode draws the per-ray offset
        from the CPU default generator, like the reference (its ``rng`` lives on the CPU, :313-317)."""
        N = int(N_samples) if N_samples > 0 else self.nSamples
        jitter = None
        if is_train:
            jitter = to_device_async(torch.rand(rays_d.reshape(-1, 3).shape[0], 1), self._all_params()[0].device).reshape(-1).contiguous()
        pts, z, valid = self._sample(rays_o, rays_d, N, jitter, False)
        lead = tuple(rays_o.shape[:-1])
        return pts.reshape(lead + (N, 3)), z.reshape(lead + (N,)), valid.reshape(lead + (N,))

    def sample_ray_ndc(self, rays_o, rays_d, is_train=True, N_samples=-1):
        """models/tensorBase.py:293-302: one depth row ``linspace(near, far, N)`` for all rays (+ one shared jitter row drawn on the rays'
        device in train mode); returns ``(rays_pts [R,N,3], interpx [1,N], ~mask_outbbox [R,N])``."""
        N = int(N_samples) if N_samples > 0 else self.nSamples
        near, far = self.near_far
        dev = self._all_params()[0].device
        interpx = torch.linspace(near, far, N).unsqueeze(0).to(rays_o)
        if is_train:
            interpx += torch.rand_like(interpx).to(rays_o) * ((far - near) / N)
        table = interpx.to(dev).reshape(-1).contiguous().float()
        pts, _, valid = self._sample(rays_o, rays_d, N, table, True)
        lead = tuple(rays_o.shape[:-1])
        return pts.reshape(lead + (N, 3)), interpx, valid.reshape(lead + (N,))

    def update_stepSize(self, gridSize):
        """models/tensorBase.py:220-231, same fp32 tensor arithmetic — evaluated on the HOST so that the scalars that
        decide which samples exist (step size, sample count) do not depend on the device's reduction order
        (a device-side mean multiplies by 1/3 where the CPU divides by 3: one ulp of step size)."""
        aabb = self.aabb.detach().float().cpu()
        size = aabb[1] - aabb[0]
        grid = torch.LongTensor([int(x) for x in gridSize])
        units = size / (grid - 1)
        step = torch.mean(units) * self.step_ratio
        diag = torch.sqrt(torch.sum(torch.square(size)))
        self.aabbSize = size.to(self.device)
        self.invaabbSize = (2.0 / size).to(self.device)
        self.gridSize = grid.to(self.device)
        self.units = units.to(self.device)
        self.stepSize = step.to(self.device)
        self.aabbDiag = diag.to(self.device)
        self.nSamples = int((diag / step).item()) + 1
        if self._handle is not None:
            lib = _lib.load()
            d = self._desc()
            _lib.check(lib.t2n_field_set_desc(self._handle, C.byref(d)), "t2n_field_set_desc")

    # ---- checkpoint surface (models/tensorBase.py:251-290) --------------------------------------------------------------
    def get_kwargs(self):
        return {"aabb": self.aabb, "gridSize": self.gridSize.tolist(), "density_n_comp": self.density_n_comp,
                "appearance_n_comp": self.app_n_comp, "app_dim": self.app_dim, "density_shift": self.density_shift,
                "alphaMask_thres": self.alphaMask_thres, "distance_scale": self.distance_scale,
                "rayMarch_weight_thres": self.rayMarch_weight_thres, "fea2denseAct": self.fea2denseAct,
                "near_far": self.near_far, "step_ratio": self.step_ratio, "shadingMode": self.shadingMode,
                "pos_pe": self.pos_pe, "view_pe": self.view_pe, "fea_pe": self.fea_pe, "featureC": self.featureC}

    def state_dict(self, *a, **k):
        sd = super().state_dict(*a, **k)
        return type(sd)((key, v) for key, v in sd.items() if not key.startswith("alphaMask."))   # the reference's mask holds no parameters

    def save(self, path):
        """models/tensorBase.py:275-283: kwargs + state_dict (+ bit-packed alpha mask)."""
        ckpt = {"kwargs": self.get_kwargs(), "state_dict": self.state_dict()}
        if self.alphaMask is not None:
            vol = self.alphaMask.alpha_volume.bool().cpu().numpy()
            ckpt.update({"alphaMask.shape": vol.shape})
            ckpt.update({"alphaMask.mask": np.packbits(vol.reshape(-1))})
            ckpt.update({"alphaMask.aabb": self.alphaMask.aabb.cpu()})
        torch.save(ckpt, path)

    def load(self, ckpt):
        """models/tensorBase.py:285-290."""
        if "alphaMask.aabb" in ckpt.keys():
            length = int(np.prod(ckpt["alphaMask.shape"]))
            vol = torch.from_numpy(np.unpackbits(ckpt["alphaMask.mask"])[:length].reshape(ckpt["alphaMask.shape"]))
            self.alphaMask = AlphaGridMask(self.device, ckpt["alphaMask.aabb"].to(self.device), vol.float().to(self.device))
        self.load_state_dict(ckpt["state_dict"], strict=False)

    # ---- C-ABI plumbing ------------------------------------------------------------------------------------------------
    def _desc(self) -> _lib.FieldDesc:
        d = _lib.FieldDesc()
        a = self.aabb.detach().float().cpu()
        inv = self.invaabbSize.detach().float().cpu()
        for k in range(3):
            d.aabb_min[k], d.aabb_max[k], d.inv_aabb_size[k] = float(a[0, k]), float(a[1, k]), float(inv[k])
            d.grid[k] = int(self.gridSize[k])
        kd, ka, kdim, kpe, kfc = self._kernel_shape()
        d.density_n_comp, d.app_n_comp, d.app_dim = kd, ka, kdim
        d.shading = _lib.SHADE_IDS[self.shadingMode]
        d.fea_pe, d.feature_c = kpe, kfc
        d.act = _lib.ACT_IDS[self.fea2denseAct]
        d.density_shift, d.distance_scale = float(self.density_shift), float(self.distance_scale)
        d.weight_thres = float(self.rayMarch_weight_thres)
        d.step_size = float(self.stepSize)
        d.near, d.far = float(self.near_far[0]), float(self.near_far[1])
        d.z_gate = float(self.z_gate)
        d.view_pe, d.pos_pe = int(self.view_pe), int(self.pos_pe)
        return d

    # ---- field / head shapes other than the kernels' own (the reference is generic in n_lamb_sigma, n_lamb_sh,
    # data_dim_color, fea_pe, featureC: models/tensoRF.py:144-160, models/tensorBase.py:88-100, e_opt.py:83-107) -----------------------
    # The kernels are built for 16 density / 48 appearance components per plane and the 27 / 6 / 128 MLP_Fea_noview head. Smaller
    # shapes — fewer components (also different counts per plane: upstream TensoRF's [16,4,4] / [48,12,12]), app_dim < 27,
    # fea_pe < 6, featureC < 128 — run on the SAME kernels through an exact algebraic embedding: missing components are zero
    # channels (they add 0 * 0 to the sum over components), a missing feature is a zero row of basis_mat whose encoding columns have
    # zero weights (sin 0 = 0, cos 0 = 1 meet a zero weight), missing octaves are zero weight columns, missing hidden units have zero
    # weights and biases (relu(0) = 0). The embedded tensors are built with differentiable torch ops (cached per parameter version), so
    # autograd carries the kernels' gradients back to the real parameters: a slow path (one padded copy per parameter change), not an
    # error. Larger shapes have no embedding: they run on the general-shape path (_is_general, csrc/t2n_generic.hip).
    KERNEL_DEN, KERNEL_APP, KERNEL_DIM, KERNEL_PE, KERNEL_FC = 16, 48, 27, 6, 128

    GENERAL_DIM, GENERAL_FC, GENERAL_PE = 64, 256, 16     # limits of the general-shape path (csrc/t2n_generic.hip)

    def _is_general(self):
        """Shapes BEYOND the tuned kernels' (more components per plane, a wider head): rendered and differentiated by the general-shape
        kernels (csrc/t2n_generic.hip: loops over the counts, reference layouts read in place) — slow, but not an error. Fixed at
        construction."""
        flag = self.__dict__.get("_general_flag")
        if flag is None:
            wide_head = False
            if isinstance(self.renderModule, nn.Module):
                wide_head = self.featureC > self.KERNEL_FC
                if self.shadingMode == "MLP_Fea_noview":
                    wide_head = wide_head or self.app_dim > self.KERNEL_DIM or self.fea_pe > self.KERNEL_PE
                else:      # view-dependent heads: the tuned general-head path holds app_dim <= 32 and 512 inputs
                    n_in = self.renderModule.mlp[0].weight.shape[1]
                    wide_head = wide_head or self.app_dim > 32 or n_in > 512
            flag = max(self.density_n_comp) > self.KERNEL_DEN or max(self.app_n_comp) > self.KERNEL_APP or wide_head
            if flag:
                if type(self) is not TensorVMSplit:
                    raise T2NError(f"{type(self).__name__} with n_comp {self.density_n_comp} / {self.app_n_comp}: the general-shape path serves "
                                   "TensorVMSplit only")
                if self.app_dim > self.GENERAL_DIM or (isinstance(self.renderModule, nn.Module) and self.featureC > self.GENERAL_FC) or \
                        max(self.fea_pe, self.view_pe) > self.GENERAL_PE or self.fea_pe < 0:
                    raise T2NError(f"app_dim {self.app_dim} / featureC {self.featureC} / fea_pe {self.fea_pe} / view_pe {self.view_pe}: beyond the "
                                   f"general-shape path ({self.GENERAL_DIM} / {self.GENERAL_FC} / {self.GENERAL_PE})")
            self.__dict__["_general_flag"] = flag
        return flag

    def _kernel_shape(self):
        """(density comps, appearance comps, app_dim, fea_pe, featureC) the native field is created with."""
        if self._is_general():
            raise T2NError("this field's shape runs on the general-shape path (forward / backward through the render call only: no native "
                           "field handle, hence no stage entry points, occupancy-mask operators or train_step)")
        if self.shadingMode == "MLP_Fea_noview":
            if self.fea_pe < 0:
                raise T2NError(f"fea_pe {self.fea_pe}")
            return self.KERNEL_DEN, self.KERNEL_APP, self.KERNEL_DIM, self.KERNEL_PE, self.KERNEL_FC
        return self.KERNEL_DEN, self.KERNEL_APP, self.app_dim, self.fea_pe, self.KERNEL_FC if isinstance(self.renderModule, nn.Module) else self.featureC

    def _needs_embed(self):
        flag = self.__dict__.get("_embed_flag")
        if flag is None and self._is_general():
            flag = self.__dict__["_embed_flag"] = False
        if flag is None:     # (component counts and head sizes are fixed at construction)
            kd, ka, kdim, kpe, kfc = self._kernel_shape()
            flag = (any(c != kd for c in self.density_n_comp) or any(c != ka for c in self.app_n_comp) or kdim != self.app_dim
                    or kpe != self.fea_pe or (isinstance(self.renderModule, nn.Module) and kfc != self.featureC))
            self.__dict__["_embed_flag"] = flag
        return flag

    def _real_params(self):
        """The 13 (+6) parameter tensors in kernel order. Read straight from the containers' dicts: nn.ParameterList / nn.Sequential
        indexing costs ~2 us per element, and this list is built several times per training step on a host-bound loop."""
        mods = self._modules
        ps = [*mods["density_plane"]._parameters.values(), *mods["density_line"]._parameters.values(),
              *mods["app_plane"]._parameters.values(), *mods["app_line"]._parameters.values(), mods["basis_mat"]._parameters["weight"]]
        rm = mods.get("renderModule")
        if rm is not None:
            m = rm._modules["mlp"]._modules
            ps += [m["0"]._parameters["weight"], m["0"]._parameters["bias"], m["2"]._parameters["weight"], m["2"]._parameters["bias"],
                   m["4"]._parameters["weight"], m["4"]._parameters["bias"]]
        return ps

    def _embedded_params(self):
        """The 19 kernel-order tensors in the kernels' shapes, as differentiable functions of the real parameters."""
        leaves = self._real_params()
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in leaves)
        key = tuple((p.data_ptr(), p._version) for p in leaves)
        cache = getattr(self, "_embed_cache", None)
        if cache is not None and cache[0] == key and (cache[1] or not grad):
            return cache[2]     # (backward runs with grad mode off: it must see the forward's tensors, not rebuild them)
        kd, ka, kdim, kpe, kfc = self._kernel_shape()
        dev = self.basis_mat.weight.device
        F = torch.nn.functional
        with torch.set_grad_enabled(grad):
            def pad_c(t, c):      # [1, C, H, W] -> [1, c, H, W]
                return t if t.shape[1] == c else F.pad(t, (0, 0, 0, 0, 0, c - t.shape[1])).contiguous()
            out = [pad_c(t, kd) for t in leaves[0:6]] + [pad_c(t, ka) for t in leaves[6:12]]
            # basis_mat [app_dim, sum C_k] -> [kdim, 3 ka]: the columns of plane k's components move to 48 k ..
            B = leaves[12]
            cols, off = [], 0
            for k, c in enumerate(self.app_n_comp):
                cols.append(torch.arange(c, device=dev) + k * ka)
                off += c
            cols = torch.cat(cols)
            Bk = B.new_zeros((kdim, 3 * ka))
            Bk[: self.app_dim, cols] = B
            out.append(Bk)
            if isinstance(self.renderModule, nn.Module):
                w0, b0, w1, b1, w2, b2 = leaves[13:19]
                fc = self.featureC
                if self.shadingMode == "MLP_Fea_noview":
                    # reference columns (models/tensorBase.py:11-17,101-104): [features | sin: feature-major, octave-minor | cos: same]
                    a, pe = self.app_dim, self.fea_pe
                    f = torch.arange(a, device=dev)
                    src = [f]
                    dst = [f]
                    if pe > 0:
                        fo = (f[:, None] * pe + torch.arange(pe, device=dev)[None]).reshape(-1)          # f * pe + o
                        fo_k = (f[:, None] * kpe + torch.arange(pe, device=dev)[None]).reshape(-1)       # f * 6 + o
                        src += [a + fo, a + a * pe + fo]
                        dst += [kdim + fo_k, kdim + kdim * kpe + fo_k]
                    src, dst = torch.cat(src), torch.cat(dst)
                    W0 = w0.new_zeros((kfc, kdim * (1 + 2 * kpe)))
                    W0[:fc, dst] = w0[:, src]
                else:   # view-dependent heads: the input row keeps the reference's own column order; only the hidden width is padded
                    W0 = w0.new_zeros((kfc, w0.shape[1]))
                    W0[:fc] = w0
                W1 = w1.new_zeros((kfc, kfc))
                W1[:fc, :fc] = w1
                W2 = w2.new_zeros((3, kfc))
                W2[:, :fc] = w2
                out += [W0, F.pad(b0, (0, kfc - fc)), W1, F.pad(b1, (0, kfc - fc)), W2, b2]
        self._embed_cache = (key, grad, out)
        return out

    def _all_params(self):
        return self._embedded_params() if self._needs_embed() else self._real_params()

    # autograd hooks: the tensors autograd tracks (the module's leaf parameters) and, for a list of same-shaped buffers (the
    # gradients), the 19 kernel-order views into them. Identity for the VM-split layout; TensorVM overrides both.
    def _autograd_params(self):
        return self._all_params()

    def _kernel_views(self, tensors):
        return list(tensors)

    def _param_struct(self, tensors, cls=_lib.FieldParams):
        s = cls()
        t = [None if x is None else x.data_ptr() for x in tensors]   # None -> NULL (e.g. deferred factor gradients)
        for k in range(3):
            s.density_plane[k] = t[k]
            s.density_line[k] = t[3 + k]
            s.app_plane[k] = t[6 + k]
            s.app_line[k] = t[9 + k]
        s.basis_weight = t[12]
        if len(t) > 13:
            s.mlp_w0, s.mlp_b0, s.mlp_w1, s.mlp_b1, s.mlp_w2, s.mlp_b2 = t[13:19]
        return s

    def supports_deferred_factor_grads(self):
        """True when the 12 factor nn.Parameters ARE the kernel's tensors (TensorVMSplit; not the stacked TensorVM / the
        embedded TensorCP) and the device keeps fp32 master copies: then ``defer_factor_grads`` lets the backward leave the
        plane / line gradients in the library's channel-last buffers for ``optim.TVAdam(field=...)`` to consume in place."""
        return type(self)._kernel_views is TensorVMSplit._kernel_views and type(self)._autograd_params is TensorVMSplit._autograd_params \
            and self.factor_storage == "fp32" and not self._needs_embed() and not self._is_general()

    def factor_grad_buffer(self, _raw=False):
        """The channel-last gradients of the 12 plane / line tensors as ONE flat fp32 tensor owned here and handed to the native
        field (t2n_field_set_grad_buffer): deferred backward calls ACCUMULATE into it — a batch split into chunks, gradient
        accumulation, a data-parallel all-reduce in place (parallel.allreduce_gradients(field=...)) — and TVAdam(field=...) consumes
        and zeroes it."""
        if getattr(self, "_gbuf", None) is not None:
            if getattr(self, "_gbuf_stale", False) and not _raw:   # consumed by the optimiser: readers see it zeroed
                self._drop_preseed()
                self._gbuf.zero_()
                self._gbuf_stale = False
                self._gbuf_dirty = False
            return self._gbuf
        h = self.sync_params()
        if getattr(self, "_gbuf", None) is None:
            lib = _lib.load()
            dev = self.basis_mat.weight.device
            n = int(lib.t2n_field_grad_buffer_bytes(h))
            self._gbuf = torch.zeros(n // 4, device=dev, dtype=torch.float32)
            _lib.check(lib.t2n_field_set_grad_buffer(h, _lib.ptr(self._gbuf), n), "t2n_field_set_grad_buffer")
            self._gbuf_dirty = False
            self._gbuf_stale = False
        return self._gbuf

    def zero_factor_grads(self, lazy=False):
        """The factor gradient buffer back to zero. `lazy` (TVAdam after it consumed the gradients): only marked — the fill runs in front
        of the next accumulating backward, or never, when train_step seeds the buffer with the TV gradient instead."""
        have = getattr(self, "_gbuf", None) is not None
        if lazy and have and self._gbuf_dirty:
            self._gbuf_stale = True           # (stays dirty: a later eager call still fills)
        else:
            self._drop_preseed()
            if have and self._gbuf_dirty:
                self._gbuf.zero_()
            self._gbuf_stale = False
            self._gbuf_dirty = False
        self._gbuf_reduced = False
        self._deferred_grad_key = None

    def _drop_preseed(self):
        """A TV seed written ahead for a train_step that is not coming: whoever fills the buffer next is ordered behind that kernel."""
        pre = self.__dict__.get("_gbuf_preseed")
        if pre is not None:
            torch.cuda.current_stream(self.basis_mat.weight.device).wait_event(pre[3])
            self.__dict__["_gbuf_preseed"] = None

    def _tv_weights(self, tv):
        tv_d = tv_a = 0.0
        for planes, weight in tv:
            if planes is self.density_plane:
                tv_d = float(weight) * 1e-2
            elif planes is self.app_plane:
                tv_a = float(weight) * 1e-2
            else:
                raise T2NError("train_step: tv entries must be tensorf.density_plane / tensorf.app_plane")
        return tv_d, tv_a

    def _factor_key(self):
        return tuple((p.data_ptr(), p._version) for p in self._all_params()[:12])

    def seed_factor_grads_with_tv(self, tv, ahead=False):
        """Initialise the factor gradient buffer to the TV gradient of the current parameters (t2n_field_tv_seed) on a side stream, beside
        whatever the current stream does next: returns the event the backward has to wait for. `tv`: [(tensorf.density_plane, weight),
        (tensorf.app_plane, weight)] as for TVAdam.step. `ahead` (train_step, right behind its optimiser step): the seed of the NEXT
        step, written while the iteration's tail of small launches runs instead of beside the next forward's march (which it slows by a
        quarter); to everybody but a train_step with the same terms on unchanged parameters the buffer then counts as consumed
        (zero-filled before use)."""
        lib = _lib.load()
        tv_d, tv_a = self._tv_weights(tv)
        dev = self.basis_mat.weight.device
        pre = self.__dict__.get("_gbuf_preseed")
        if not ahead and pre is not None and getattr(self, "_gbuf_stale", False) and pre[:2] == (tv_d, tv_a) and pre[2] == self._factor_key():
            self.__dict__["_gbuf_preseed"] = None          # written ahead by the previous step: take it
            self._gbuf_stale = False
            self._gbuf_dirty = True
            self._gbuf_reduced = False
            return pre[3]
        h = self.sync_params()
        self.factor_grad_buffer(_raw=True)     # (every element is overwritten: a pending zero fill is dropped)
        cur = torch.cuda.current_stream(dev)
        side = self.__dict__.get("_seed_stream")
        if side is None:
            side = self.__dict__["_seed_stream"] = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)            # the previous step's Adam has read the buffer and written the parameters
        with torch.cuda.device(dev):
            _lib.check(lib.t2n_field_tv_seed(h, tv_d, tv_a, side.cuda_stream), "t2n_field_tv_seed")
        ev = side.record_event()
        self._gbuf_dirty = True
        self._gbuf_reduced = False
        if ahead:
            self._gbuf_stale = True
            self.__dict__["_gbuf_preseed"] = (tv_d, tv_a, self._factor_key(), ev)
        else:
            self._gbuf_stale = False
            self.__dict__["_gbuf_preseed"] = None
        return ev

    def sync_params(self, force=False, frame_width=None):
        """Create the native field on first use and re-upload when any parameter changed (in-place optimiser steps
        and load_state_dict bump tensor versions). `frame_width`: this call's raster width (default: self.frame_width)."""
        lib = _lib.load()
        ps = self._all_params()
        dev = ps[0].device
        if dev.type != "cuda":
            raise T2NError(f"TensorVMSplit parameters live on {dev}; the renderer runs on an MI355X (cuda/HIP) only")
        for p in ps:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise T2NError("parameters must be contiguous float32")
        if self._handle is None:
            h = C.c_void_p()
            d = self._desc()
            _lib.check(lib.t2n_field_create(C.byref(d), C.byref(h)), "t2n_field_create")
            self._handle = h
            self._precision_set = None
            self._term_set = None
            self._storage_set = None
            self._frame_w_set = None
            self._head_rows_set = 32
            self._alpha_key = "unset"
        queued = False     # device work queued by this call (mask / parameter uploads)
        mask = self.alphaMask
        mkey = None if mask is None else (mask.alpha_volume.data_ptr(), mask.alpha_volume._version, tuple(mask.alpha_volume.shape))
        if getattr(self, "_alpha_key", "unset") != mkey:
            queued = True
            self._wait_readers(dev)
            with torch.cuda.device(dev):
                if mask is None:
                    _lib.check(lib.t2n_field_set_alpha_mask(self._handle, None, 0, 0, 0, None, None, _lib.current_stream_ptr(dev)),
                               "t2n_field_set_alpha_mask")
                else:
                    vol = mask.alpha_volume.reshape(mask.alpha_volume.shape[-3:]).contiguous().float().to(dev)
                    amin = (C.c_float * 3)(*mask.aabb[0].float().cpu().tolist())
                    ainv = (C.c_float * 3)(*mask.invgridSize.float().cpu().tolist())
                    D, H, W = vol.shape
                    _lib.check(lib.t2n_field_set_alpha_mask(self._handle, _lib.ptr(vol), D, H, W, amin, ainv,
                                                            _lib.current_stream_ptr(dev)), "t2n_field_set_alpha_mask")
                    object.__setattr__(mask, "_field", self)   # plain attribute: a Module attribute would register a cycle
            self._alpha_key = mkey
        hr = int(getattr(self, "head_scratch_rows_per_ray", 32))
        if getattr(self, "_head_rows_set", 32) != hr:
            _lib.check(lib.t2n_field_set_head_scratch_rows(self._handle, hr), "t2n_field_set_head_scratch_rows")
            self._head_rows_set = hr
        fw = int(self.frame_width if frame_width is None else frame_width)
        if getattr(self, "_frame_w_set", None) != fw:
            _lib.check(lib.t2n_field_set_frame_width(self._handle, fw), "t2n_field_set_frame_width")
            self._frame_w_set = fw
        if self._precision_set != bool(self.mlp_exact_fp32):
            _lib.check(lib.t2n_field_set_mlp_precision(self._handle, 1 if self.mlp_exact_fp32 else 0),
                       "t2n_field_set_mlp_precision")
            self._precision_set = bool(self.mlp_exact_fp32)
        # (rayMarch_weight_thres = 0 — `weight > thres`, models/tensorBase.py:477 — is a legitimate setting: 0 then means off)
        et = max(0.0, min(float(self.early_termination or 0.0), float(self.rayMarch_weight_thres)))
        if getattr(self, "_term_set", None) != et:
            _lib.check(lib.t2n_field_set_early_termination(self._handle, et), "t2n_field_set_early_termination")
            self._term_set = et
        if self.factor_storage not in ("fp32", "bf16"):
            raise T2NError(f"factor_storage {self.factor_storage!r}: expected 'fp32' or 'bf16'")
        if getattr(self, "_storage_set", None) != self.factor_storage:
            _lib.check(lib.t2n_field_set_factor_storage(self._handle, 1 if self.factor_storage == "bf16" else 0),
                       "t2n_field_set_factor_storage")
            self._storage_set = self.factor_storage
            force = True
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if force or key != self._uploaded_key:
            self._wait_readers(dev)
            with torch.cuda.device(dev):
                st = self._param_struct([p.detach() for p in ps])
                if not force and self._uploaded_key is not None and key[:12] == getattr(self, "_device_factor_key", None):
                    # the factor copies on the device are current (optim.TVAdam stepped them in place): head only
                    _lib.check(lib.t2n_field_upload_head(self._handle, C.byref(st), _lib.current_stream_ptr(dev)),
                               "t2n_field_upload_head")
                else:
                    _lib.check(lib.t2n_field_upload(self._handle, C.byref(st), _lib.current_stream_ptr(dev)),
                               "t2n_field_upload")
            self._uploaded_key = key
            self._device_factor_key = key[:12]
            queued = True
        # The uploads above are asynchronous kernels on the CALLER's stream. Renders of this field on other streams (frames in flight
        # on alternating streams: renderer._FramePipe) must not read half-written factor / head / mask copies: an event behind the
        # upload, waited for once by every other stream that comes through here.
        cur = torch.cuda.current_stream(dev)
        if queued:
            ev = torch.cuda.Event()
            ev.record(cur)
            self._upload_event, self._upload_seen, self._upload_stream = ev, {int(cur.cuda_stream)}, int(cur.cuda_stream)
        elif getattr(self, "_upload_event", None) is not None and int(cur.cuda_stream) not in self._upload_seen:
            cur.wait_event(self._upload_event)
            self._upload_seen.add(int(cur.cuda_stream))
        return self._handle

    def _wait_readers(self, dev):
        """ADVICE r4 (write-after-read): an upload rewrites the factor / head / mask copies in place; frames of this field still in flight
        on OTHER streams (renderer._FramePipe) read them. Every render on a stream other than the last upload's leaves an event
        (_note_reader); the uploading stream waits for those before it overwrites anything. (One _FramePipe per device at a time: the
        pipes share their side streams and scratch.)"""
        readers = self.__dict__.get("_reader_events")
        if readers:
            cur = torch.cuda.current_stream(dev)
            for sid, ev in readers.items():
                if sid != int(cur.cuda_stream):
                    cur.wait_event(ev)
            readers.clear()

    def _note_reader(self, dev):
        seen = getattr(self, "_upload_seen", None)
        cur = torch.cuda.current_stream(dev)
        sid = int(cur.cuda_stream)
        if seen is None or sid == getattr(self, "_upload_stream", sid):
            return                      # same stream as the uploads: ordered by the stream itself
        ev = torch.cuda.Event()
        ev.record(cur)
        self.__dict__.setdefault("_reader_events", {})[sid] = ev

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _lib.load().t2n_field_destroy(self._handle)
                self._handle = None
                self._device_rows_ovf_seen = self._device_rows_issued = 0   # (the new handle's record starts from zero)
        except Exception:
            pass

    def timing(self, on=True, kernels=None):
        """Bracket kernel launches with HIP events (t2n_timing_enable). ``kernels``: names from ``_lib.KERNEL_NAMES`` to bracket only those
        (an event pair costs the stream a ~10 us bubble per bracketed group: inside a timed region ask for the kernel you need)."""
        mode = 1 if on else 0
        if on and kernels is not None:
            mode = sum(1 << (_lib.KERNEL_NAMES.index(k) + 1) for k in kernels)
        _lib.check(_lib.load().t2n_timing_enable(self.sync_params(), mode), "t2n_timing_enable")

    def read_timing(self, reset=True):
        ms = (C.c_double * _lib.T2N_K_COUNT)()
        n = (C.c_int64 * _lib.T2N_K_COUNT)()
        _lib.check(_lib.load().t2n_timing_read(self.sync_params(), ms, n, 1 if reset else 0), "t2n_timing_read")
        return {_lib.KERNEL_NAMES[k]: (ms[k], n[k]) for k in range(_lib.T2N_K_COUNT) if n[k]}

    # ---- stage methods (reference names) ---------------------------------------------------------------------------------
    def normalize_coord(self, xyz_sampled):
        return (xyz_sampled - self.aabb[0]) * self.invaabbSize - 1

    def feature2density(self, density_features):
        if self.fea2denseAct == "softplus":
            return torch.nn.functional.softplus(density_features + self.density_shift)
        return torch.relu(density_features)

    def _density_at(self, xyz, want_sigma):
        lib = _lib.load()
        h = self.sync_params()
        xyz = xyz.detach().reshape(-1, 3).contiguous().float().to(self.basis_mat.weight.device)
        n = xyz.shape[0]
        out = torch.empty(n, device=xyz.device, dtype=torch.float32)
        with torch.cuda.device(xyz.device):
            _lib.check(lib.t2n_density_at(h, _lib.ptr(xyz), n, None if want_sigma else _lib.ptr(out),
                                          _lib.ptr(out) if want_sigma else None, _lib.current_stream_ptr(xyz.device)),
                       "t2n_density_at")
        return out

    def _alpha_at(self, xyz_world):
        lib = _lib.load()
        h = self.sync_params()
        dev = self.basis_mat.weight.device
        xyz = xyz_world.detach().reshape(-1, 3).contiguous().float().to(dev)
        out = torch.empty(xyz.shape[0], device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.t2n_alpha_at(h, _lib.ptr(xyz), xyz.shape[0], _lib.ptr(out), _lib.current_stream_ptr(dev)), "t2n_alpha_at")
        return out

    def compute_sigma(self, xyz_norm):
        """compute_densityfeature + feature2density fused in the kernel."""
        return self._density_at(xyz_norm, want_sigma=True)

    def shade(self, xyz_norm, viewdirs=None, want_features=True, want_rgb=True):
        """compute_appfeature (+ renderModule) at normalised points: returns (app_features [n,app_dim], rgb [n,3])."""
        lib = _lib.load()
        h = self.sync_params()
        dev = self.basis_mat.weight.device
        xyz = xyz_norm.detach().reshape(-1, 3).contiguous().float().to(dev)
        n = xyz.shape[0]
        vd = None if viewdirs is None else viewdirs.detach().reshape(-1, 3).contiguous().float().to(dev)
        feat = torch.empty(n, self.app_dim, device=dev, dtype=torch.float32) if want_features else None
        rgb = torch.empty(n, 3, device=dev, dtype=torch.float32) if want_rgb else None
        ws = workspace(dev, 256)
        with torch.cuda.device(dev):
            _lib.check(lib.t2n_shade_at(h, _lib.ptr(xyz), _lib.ptr(vd), n, _lib.ptr(feat), _lib.ptr(rgb), _lib.ptr(ws),
                                        ws.numel(), _lib.current_stream_ptr(dev)), "t2n_shade_at")
        return feat, rgb

    @torch.no_grad()
    def filtering_rays(self, all_rays, all_rgbs, all_depth=None, N_samples=256, chunk=10240 * 5, bbox_only=False):
        """models/tensorBase.py:372-404: slab test (bbox_only=True, the driver's call, text2nerf_main.py:477) or "any of the
        ray's N_samples samples lies in an occupied cell of the alphaMask"."""
        if not bbox_only and self.alphaMask is None:
            raise T2NError("filtering_rays(bbox_only=False) needs an alphaMask (updateAlphaMask() or a checkpoint's)")
        lib = _lib.load()
        h = self.sync_params()
        dev = self.basis_mat.weight.device
        tt = time.time()
        flat = all_rays.reshape(-1, all_rays.shape[-1])
        masks = []
        for idx in torch.split(torch.arange(flat.shape[0]), chunk):
            r = flat[idx].to(dev).contiguous().float()
            m = torch.empty(r.shape[0], dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                if bbox_only:
                    _lib.check(lib.t2n_filter_rays_bbox(h, _lib.ptr(r), r.shape[0], r.shape[1], _lib.ptr(m),
                                                        _lib.current_stream_ptr(dev)), "t2n_filter_rays_bbox")
                else:
                    _lib.check(lib.t2n_filter_rays_alpha(h, _lib.ptr(r), r.shape[0], r.shape[1], int(N_samples), _lib.ptr(m),
                                                         _lib.current_stream_ptr(dev)), "t2n_filter_rays_alpha")
            masks.append(m.bool().cpu())
        mask = torch.cat(masks).view(all_rgbs.shape[:-1])
        print(f"Ray filtering done! takes {time.time() - tt} s. ray mask ratio: {torch.sum(mask) / flat.shape[0]}")
        if all_depth is not None:
            return all_rays[mask], all_rgbs[mask], all_depth[mask]
        return all_rays[mask], all_rgbs[mask]

    # ---- coarse-to-fine / occupancy maintenance ----------------------------------------------------------
    def _drop_handle(self):
        """Grid shape or aabb changed: the native field is rebuilt on the next use."""
        if self.__dict__.get("_gbuf_preseed") is not None:
            self._drop_preseed()      # a seed kernel in flight reads the old field and writes the old buffer
        if self._handle is not None:
            _lib.load().t2n_field_destroy(self._handle)
            self._handle = None
            self._device_rows_ovf_seen = self._device_rows_issued = 0
        self._uploaded_key = None
        self._gbuf = None            # channel-last factor gradients: sized by the grid
        self._gbuf_dirty = False
        self._gbuf_stale = False
        self.__dict__["_gbuf_preseed"] = None

    @torch.no_grad()
    def compute_alpha(self, xyz_locs, length=1):
        """models/tensorBase.py:412-434: 1 - exp(-sigma * length) at world-space points (sigma = 0 outside the alphaMask)."""
        lib = _lib.load()
        h = self.sync_params()
        dev = self.basis_mat.weight.device
        xyz = xyz_locs.detach().reshape(-1, 3).contiguous().float().to(dev)
        out = torch.empty(xyz.shape[0], device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.t2n_compute_alpha(h, _lib.ptr(xyz), xyz.shape[0], float(length), _lib.ptr(out),
                                             _lib.current_stream_ptr(dev)), "t2n_compute_alpha")
        return out.view(xyz_locs.shape[:-1])

    @torch.no_grad()
    def getDenseAlpha(self, gridSize=None):
        """models/tensorBase.py:328-344: alpha at the nodes of a dense grid over the aabb, one step long; returns
        (alpha [gx,gy,gz], dense_xyz [gx,gy,gz,3]). Node positions are generated inside the kernel from the same
        torch.linspace(0, 1, g) values the reference lerps with."""
        lib = _lib.load()
        h = self.sync_params()
        dev = self.basis_mat.weight.device
        g = [int(x) for x in (self.gridSize if gridSize is None else gridSize)]
        lins = [torch.linspace(0, 1, n).to(dev) for n in g]
        alpha = torch.empty(g, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.t2n_dense_alpha(h, _lib.ptr(lins[0]), _lib.ptr(lins[1]), _lib.ptr(lins[2]), g[0], g[1], g[2],
                                           float(self.stepSize), _lib.ptr(alpha), _lib.current_stream_ptr(dev)),
                       "t2n_dense_alpha")
        samples = torch.stack(torch.meshgrid(*lins, indexing="ij"), -1)
        dense_xyz = self.aabb[0] * (1 - samples) + self.aabb[1] * samples
        return alpha, dense_xyz

    @torch.no_grad()
    def updateAlphaMask(self, gridSize=(200, 200, 200)):
        """models/tensorBase.py:346-370: rebuild the occupancy mask from the current density (3x3x3 dilation, threshold
        alphaMask_thres) and return the bounding box of the occupied voxels."""
        lib = _lib.load()
        dev = self.basis_mat.weight.device
        g = [int(x) for x in gridSize]
        alpha, _ = self.getDenseAlpha(g)
        vol = torch.empty((g[2], g[1], g[0]), device=dev, dtype=torch.float32)
        box = torch.empty(6, device=dev, dtype=torch.int32)
        with torch.cuda.device(dev):
            _lib.check(lib.t2n_alpha_volume(_lib.ptr(alpha), g[0], g[1], g[2], float(self.alphaMask_thres), _lib.ptr(vol),
                                            _lib.ptr(box), _lib.current_stream_ptr(dev)), "t2n_alpha_volume")
        self.alphaMask = AlphaGridMask(self.device, self.aabb, vol)
        b = box.cpu().tolist()
        if b[3] < 0:
            raise T2NError("updateAlphaMask: no voxel reaches alphaMask_thres (the reference fails on the empty amin too)")
        lins = [torch.linspace(0, 1, n) for n in g]
        a = self.aabb.detach().float().cpu()
        lo = torch.stack([a[0, k] * (1 - lins[k][b[k]]) + a[1, k] * lins[k][b[k]] for k in range(3)])
        hi = torch.stack([a[0, k] * (1 - lins[k][b[3 + k]]) + a[1, k] * lins[k][b[3 + k]] for k in range(3)])
        new_aabb = torch.stack((lo, hi)).to(self.aabb.device)
        total = int(g[0] * g[1] * g[2])
        print(f"bbox: {lo, hi} alpha rest %%%f" % (float(vol.sum()) / total * 100))
        return new_aabb

    # ---- the render call ----------------------------------------------------------------------------------------------------
    def forward(self, rays_chunk, white_bg=True, is_train=False, ndc_ray=False, N_samples=-1, frame_width=None):
        """models/tensorBase.py:436-507: returns (rgb_map [R,3], depth_map [R], z_vals [R,N], weight [R,N]). `frame_width` (not in
        the reference): the raster width of this call's rays when they are a whole row-major image (default: self.frame_width)."""
        dev = self.basis_mat.weight.device
        jitter = None
        if is_train and not ndc_ray and rays_chunk.shape[0]:
            # the reference draws on the CPU default generator even for GPU runs (models/tensorBase.py:313-317); host rays and the
            # draw travel as one staged copy
            rays, jitter = to_device_async_many([rays_chunk, torch.rand(rays_chunk.shape[0], 1)], dev)
            jitter = jitter.reshape(-1).contiguous()
        else:
            rays = to_device_async(rays_chunk, dev)
        if rays.dtype != torch.float32 or not rays.is_contiguous():
            rays = rays.contiguous().float()
        R = rays.shape[0]
        N = int(N_samples) if N_samples > 0 else self.nSamples
        if R == 0:
            e = torch.empty(0, N, device=dev) if self.materialize_weights or is_train else None
            return torch.empty(0, 3, device=dev), torch.empty(0, device=dev), e, e
        add_bg = bool(white_bg)
        if ndc_ray:
            # sample_ray_ndc (models/tensorBase.py:293-299): one depth table for all rays, linspace(near, far, N) on the rays'
            # device (the reference's chunk is on the GPU by then), plus — in train mode — ONE shared jitter row drawn
            # with rand_like on that device's generator. The table rides in the `jitter` slot of the C-ABI (T2N_FLAG_NDC).
            near, far = self.near_far
            interpx = torch.linspace(near, far, N).unsqueeze(0).to(rays)
            if is_train:
                interpx += torch.rand_like(interpx).to(rays) * ((far - near) / N)
            jitter = interpx.reshape(-1).contiguous()
        if is_train and not white_bg:
            add_bg = bool(torch.rand((1,)) < 0.5)   # models/tensorBase.py:497
        flags = (FLAG_TRAIN if is_train else 0) | (FLAG_ADD_BG if add_bg else 0) | (FLAG_NDC if ndc_ray else 0)
        fw = int(self.frame_width if frame_width is None else frame_width)
        if not is_train and not ndc_ray and fw and R % fw == 0:
            flags |= FLAG_COHERENT
        if self._is_general():
            if ndc_ray or self.alphaMask is not None:
                raise T2NError("the general-shape path has no NDC sampling and no AlphaGridMask")
            flags &= ~FLAG_COHERENT
            ps = self._real_params()
            if torch.is_grad_enabled() and any(p.requires_grad for p in ps):
                return _GeneralFn.apply(self, rays, N, flags, jitter, *ps)
            rgb, depth, z, w, _ = self._general_forward(rays, N, flags, jitter)
            return (rgb, depth, z, w) if self.materialize_weights or is_train else (rgb, depth, None, None)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._autograd_params())
        if needs_grad and self._needs_embed():
            # every differentiable forward gets its own embedding graph: a second forward + backward at an unchanged parameter version
            # (gradient accumulation) must not walk the first one's freed graph
            self._embed_cache = None
        if needs_grad:
            rgb, depth, z, w = _RenderFn.apply(self, rays, N, flags, jitter, *self._autograd_params())
        else:
            rgb, depth, z, w = self._render_raw(rays, N, flags, jitter, self.materialize_weights, frame_width=fw)
        if ndc_ray and z is not None:
            z = z[:1]          # sample_ray_ndc returns ONE [1,N] depth row (models/tensorBase.py:296-302)
        return rgb, depth, z, w

    def _render_raw(self, rays, N, flags, jitter, want_wz, keep_ctx=False, frame_width=None, reuse_ctx=False):
        lib = _lib.load()
        h = self.sync_params(frame_width=frame_width)
        dev = rays.device
        R = rays.shape[0]
        rgb = torch.empty(R, 3, device=dev, dtype=torch.float32)
        depth = torch.empty(R, device=dev, dtype=torch.float32)
        w = torch.empty(R, N, device=dev, dtype=torch.float32) if want_wz else None
        z = torch.empty(R, N, device=dev, dtype=torch.float32) if want_wz else None
        stats = torch.empty(_lib.T2N_STAT_COUNT, device=dev, dtype=torch.int64)
        if keep_ctx:
            # the kept context must survive until backward: a private buffer, not the shared scratch
            # plus room for the activation rows of the appearance samples (1728 B each), sized from the previous backward's
            # row count: the forward then keeps them and the backward skips its appearance recompute (too small a guess
            # only costs that recompute)
            rows_hint = int(getattr(self, "_ctx_rows_hint", 0))
            rows_hint = _ladder(rows_hint) // 32 * 32 if rows_hint else 0     # (a few distinct sizes over a run: the allocator reuses its blocks)
            need = int(lib.t2n_render_workspace_bytes_ctx(R, N)) + (256 + rows_hint * 1728 if rows_hint else 0) \
                + int(lib.t2n_render_head_scratch_bytes(h, R))       # (general view-dependent heads: activation scratch behind the context)
            self._ctx_rows_cap = rows_hint
            if reuse_ctx:
                # train_step: the step's backward is queued before the next step's forward, so ONE retained buffer serves every step
                # (stream order); it grows like workspace() does — old buffer dropped first, cached blocks returned to the driver
                ws = getattr(self, "_ctx_buf", None)
                if ws is None or ws.numel() < need or ws.device != dev:
                    had = ws is not None
                    self._ctx_buf = ws = None
                    if had:
                        torch.cuda.empty_cache()
                    self._ctx_buf = ws = torch.empty(need, dtype=torch.uint8, device=dev)
                ws = ws[:need]
            else:
                ws = torch.empty(need, dtype=torch.uint8, device=dev)
            flags |= FLAG_KEEP_CTX
        else:
            need = int(lib.t2n_render_workspace_bytes(max(R, 1), N)) + int(lib.t2n_render_head_scratch_bytes(h, max(R, 1)))
            if flags & FLAG_COHERENT:
                # image-ordered frames: lists budgeted from what the previous frame needed instead of the worst case (the library
                # checks the march kernels' counters and redoes an overflowed frame with worst-case lists)
                need = min(need, int(lib.t2n_render_workspace_bytes_hint(h, max(R, 1), N)))
            ws = workspace(dev, min(need, max(workspace_budget(dev), int(lib.t2n_render_workspace_bytes(1024, N)) + int(lib.t2n_render_head_scratch_bytes(h, 1024)))))
            if getattr(self, "workspace_bytes_override", None):    # tests: hand the call exactly this much of the buffer
                ws = workspace(dev, int(self.workspace_bytes_override))[:int(self.workspace_bytes_override)]
        with torch.cuda.device(dev):
            _lib.check(lib.t2n_render_forward(h, _lib.ptr(rays), R, rays.shape[1], N, flags, _lib.ptr(jitter),
                                              _lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(w), _lib.ptr(z),
                                              _lib.ptr(stats) if getattr(self, 'collect_stats', True) else None,
                                              _lib.ptr(ws), ws.numel(), _lib.current_stream_ptr(dev)),
                       "t2n_render_forward")
        self.last_stats = stats
        self._note_reader(dev)
        if keep_ctx:
            return rgb, depth, z, w, ws
        return rgb, depth, z, w

    def _backward_raw(self, rays, jitter, N, flags, ws, d_rgb, d_depth, d_w, head_grads=None, device_rows=False):
        """t2n_render_backward for a KEEP_CTX forward (workspace `ws`): returns the 19 gradient tensors in autograd order (None for the
        12 factor tensors in deferred mode: those accumulate in factor_grad_buffer()). `head_grads`: optional preallocated, zeroed
        gradient tensors of the 7 head tensors (train_step reuses one flat buffer)."""
        lib = _lib.load()
        dev = rays.device
        params = self._autograd_params()
        # deferred mode: the 12 plane / line gradients stay channel-last in the field's gradient buffer (no layout pass, no
        # zero-filled 69.6 MB of gradient tensors); optim.TVAdam(field=...) steps from there
        defer = bool(getattr(self, "defer_factor_grads", False)) and self.supports_deferred_factor_grads()
        if defer:
            self.factor_grad_buffer()     # caller-owned: this backward accumulates into it (chunked batches add up; a zero fill the
            self._gbuf_dirty = True       # optimiser left pending runs here)
        if head_grads is not None and defer:
            grads = [None] * 12 + list(head_grads)
        else:
            # ONE zero-filled allocation for the gradient tensors (19 zeros_like calls are 19 allocations and 19 fill launches on a loop
            # whose host side is the bottleneck); fresh per backward: autograd may adopt the views as the parameters' .grad
            want = [p for i, p in enumerate(params) if not (defer and i < 12)]
            flat = torch.zeros(sum(p.numel() for p in want), device=dev, dtype=torch.float32)
            views, off = [], 0
            for p in want:
                views.append(flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
            grads = [None] * (len(params) - len(want)) + views if defer else views
        gs = self._param_struct(self._kernel_views(grads), _lib.FieldGrads)
        R = rays.shape[0]
        d_rgb = torch.zeros(R, 3, device=dev) if d_rgb is None else d_rgb.contiguous().float()
        d_depth = torch.zeros(R, device=dev) if d_depth is None else d_depth.contiguous().float()
        d_w = None if d_w is None else d_w.contiguous().float()
        st = _lib.current_stream_ptr(dev)
        if device_rows:
            # nothing is read back: the capacity is the hint the forward kept its activation rows with; the kernels clip to the
            # row count on the device (T2N_FLAG_DEVICE_ROWS). train_step polls the count afterwards, without waiting.
            with torch.cuda.device(dev):
                need = int(lib.t2n_backward_workspace_bytes(self._handle, int(self._ctx_rows_cap), R, N))
                have = _WORKSPACE.get(_ws_key(dev))
                bws = workspace(dev, need) if have is not None and have.numel() >= need else workspace(dev, int(need * 1.25))
                _lib.check(lib.t2n_render_backward(self._handle, _lib.ptr(rays), R, rays.shape[1], N, flags | FLAG_KEEP_CTX | FLAG_DEVICE_ROWS,
                                                   _lib.ptr(jitter), _lib.ptr(d_rgb), _lib.ptr(d_depth), _lib.ptr(d_w),
                                                   C.byref(gs), _lib.ptr(ws), ws.numel(), _lib.ptr(bws), bws.numel(),
                                                   st), "t2n_render_backward")
            self._deferred_grad_key = self._uploaded_key if defer else None
            return grads
        with torch.cuda.device(dev):
            rows = C.c_int64(0)
            # the counts reached pinned host memory behind the forward's march kernel: this waits for that copy's event, not
            # for the stream (the loss kernels queued since keep running)
            _lib.check(lib.t2n_render_ctx_rows(_lib.ptr(ws), R, N, st, C.byref(rows)), "t2n_render_ctx_rows")
            # shared grow-only scratch (geometric growth): the row count changes every iteration, and a fresh
            # multi-GB torch.empty per backward would hit hipMalloc each time
            self._ctx_rows_hint = (int(rows.value * 1.25) + 95) // 32 * 32 if self.keep_activation_rows else 0
            need = int(lib.t2n_backward_workspace_bytes(self._handle, rows.value, R, N))
            have = _WORKSPACE.get(_ws_key(dev))
            bws = workspace(dev, need) if have is not None and have.numel() >= need else workspace(dev, int(need * 1.25))
            _lib.check(lib.t2n_render_backward(self._handle, _lib.ptr(rays), R, rays.shape[1], N, flags | FLAG_KEEP_CTX,
                                               _lib.ptr(jitter), _lib.ptr(d_rgb), _lib.ptr(d_depth), _lib.ptr(d_w),
                                               C.byref(gs), _lib.ptr(ws), ws.numel(), _lib.ptr(bws), bws.numel(),
                                               st), "t2n_render_backward")
        self._deferred_grad_key = self._uploaded_key if defer else None   # which parameters the device-side gradients belong to
        return grads

    # ---- general-shape path (csrc/t2n_generic.hip) --------------------------------------------------------------------------------
    def _general_desc(self):
        d = _lib.GenericDesc()
        a = self.aabb.detach().float().cpu()
        inv = self.invaabbSize.detach().float().cpu()
        for k in range(3):
            d.aabb_min[k], d.aabb_max[k], d.inv_aabb_size[k] = float(a[0, k]), float(a[1, k]), float(inv[k])
            d.grid[k] = int(self.gridSize[k])
            d.density_n_comp[k], d.app_n_comp[k] = int(self.density_n_comp[k]), int(self.app_n_comp[k])
        d.app_dim, d.shading = int(self.app_dim), _lib.SHADE_IDS[self.shadingMode]
        d.fea_pe, d.view_pe, d.feature_c = int(self.fea_pe), int(self.view_pe), int(self.featureC)
        d.act = _lib.ACT_IDS[self.fea2denseAct]
        d.density_shift, d.distance_scale = float(self.density_shift), float(self.distance_scale)
        d.weight_thres, d.step_size = float(self.rayMarch_weight_thres), float(self.stepSize)
        d.near, d.far, d.z_gate = float(self.near_far[0]), float(self.near_far[1]), float(self.z_gate)
        return d

    def _general_forward(self, rays, N, flags, jitter):
        """t2n_generic_forward: (rgb, depth, z_vals, weights, workspace). The workspace holds the context the backward needs."""
        lib = _lib.load()
        ps = self._real_params()
        dev = rays.device
        if dev.type != "cuda" or any(p.device != dev for p in ps):
            raise T2NError("the renderer runs on an MI355X (cuda/HIP) only: rays and parameters must live there")
        R = rays.shape[0]
        if R * N > 0x7fffffff:
            raise T2NError("general-shape path: more than 2^31 samples in one call (render in chunks: OctreeRender_trilinear_fast does)")
        rgb, depth = torch.empty(R, 3, device=dev), torch.empty(R, device=dev)
        w, z = torch.empty(R, N, device=dev), torch.empty(R, N, device=dev)
        stats = torch.empty(_lib.T2N_STAT_COUNT, device=dev, dtype=torch.int64)
        d = self._general_desc()
        ws = torch.empty(int(lib.t2n_generic_workspace_bytes_desc(C.byref(d), R, N)), dtype=torch.uint8, device=dev)
        st = self._param_struct([p.detach() for p in ps])
        with torch.cuda.device(dev):
            _lib.check(lib.t2n_generic_forward(C.byref(d), C.byref(st), _lib.ptr(rays), R, rays.shape[1], N, flags, _lib.ptr(jitter),
                                               _lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(w), _lib.ptr(z), _lib.ptr(stats), _lib.ptr(ws),
                                               ws.numel(), _lib.current_stream_ptr(dev)), "t2n_generic_forward")
        self.last_stats = stats
        return rgb, depth, z, w, ws

    def _general_backward(self, rays, jitter, N, flags, ws, w, z, d_rgb, d_depth, d_w):
        lib = _lib.load()
        ps = self._real_params()
        dev = rays.device
        R = rays.shape[0]
        grads = [torch.zeros_like(p) for p in ps]
        d_rgb = torch.zeros(R, 3, device=dev) if d_rgb is None else d_rgb.contiguous().float()
        d_depth = torch.zeros(R, device=dev) if d_depth is None else d_depth.contiguous().float()
        d_w = None if d_w is None else d_w.contiguous().float()
        d = self._general_desc()
        st = self._param_struct([p.detach() for p in ps])
        gs = self._param_struct(grads, _lib.FieldGrads)
        with torch.cuda.device(dev):
            _lib.check(lib.t2n_generic_backward(C.byref(d), C.byref(st), _lib.ptr(rays), R, rays.shape[1], N, flags, _lib.ptr(jitter),
                                                _lib.ptr(w), _lib.ptr(z), _lib.ptr(d_rgb), _lib.ptr(d_depth), _lib.ptr(d_w), C.byref(gs),
                                                _lib.ptr(ws), ws.numel(), _lib.current_stream_ptr(dev)), "t2n_generic_backward")
        return grads

    def train_step(self, rays, rgb_target, depth_target, optimizer, N_samples=-1, white_bg=True, w_depth=0.005, w_trans=1e3, delta=0.1,
                   tv=(), all_reduce=None, all_reduce_averages=True, speculative=False, fused=None, graph=None):
        """One optimisation step of text2nerf_main.py:547-590 without the autograd graph: render (train mode, CPU-generator jitter
        like models/tensorBase.py:313-317) -> the driver's loss as ONE kernel that emits d_rgb / d_depth / d_weights
        (t2n_train_loss) -> t2n_render_backward -> optimizer.step(). `optimizer`: optim.TVAdam(field=self) (TV terms via `tv`, as
        in TVAdam.step); `all_reduce`: optional callable run between backward and step (data-parallel:
        lambda: parallel.allreduce_gradients(params, field=self)); it must AVERAGE over the ranks (the TV terms, identical on every
        rank, may already sit in the gradient buffer when it runs: a mean leaves them as they are, a sum would multiply them by the
        world size) — pass all_reduce_averages=False for any other reduction and the TV terms are added after it. Returns the device tensor [mse, depth loss, transmittance loss,
        total] of this batch (no host synchronisation). Same arithmetic as the autograd path: tests/test_train_step.py.
        `speculative=True`: the backward does not wait for the forward's appearance row count either (T2N_FLAG_DEVICE_ROWS): its
        row capacity is 1.25x the largest need of the last eight steps, the kernels clip to the actual count on the device, and the
        host learns every step's need from a record in pinned memory, without waiting (late at worst, never lost). A step whose count
        exceeds the capacity drops the appearance gradients of the rows past it — counted in `self.device_rows_overflows`. The first
        step (and the step after a recorded overflow) takes the counted route.
        `fused` (default: whenever it applies — this field shape, fused MLP_Fea_noview head in split-f16 mode, no alpha mask, optimizer =
        optim.TVAdam(field=self), an averaging all_reduce): the whole step is ONE C call (t2n_train_step, text2nerf_amd/trainer.py) that
        reads nothing on the host; a step whose appearance rows exceed its capacity applies NO update and is submitted again (never a
        truncated gradient). Host batches take the PIPELINED form of the call: the step's early part (batch copy, march, plan, appearance
        binning) runs beside the previous step's tail. `graph=True` (single process): the call captured once per input buffer into a
        hipGraph and replayed — one submission per step, but ROCm's graph executor serialises what the eager call overlaps (1.02 against
        0.86 ms per 16 384-ray step measured), so it is opt-in. The returned loss tensor is a fixed buffer the next step overwrites."""
        lib = _lib.load()
        params = self._autograd_params()
        can_fuse = self._can_fuse_train_step(optimizer) and (all_reduce is None or all_reduce_averages) and not speculative
        if fused and not can_fuse:
            raise T2NError("train_step(fused=True): needs the tuned field shape with the MLP_Fea_noview head (split-f16 arithmetic), fp32 factor "
                           "storage, no alpha mask, optim.TVAdam(field=tensorf) and an averaging all_reduce")
        if can_fuse and fused is not False:
            from .trainer import FusedStep
            fs = self.__dict__.get("_fused_step")
            if fs is None or fs.opt is not optimizer:
                fs = self.__dict__["_fused_step"] = FusedStep(self, optimizer)
            N = int(N_samples) if N_samples > 0 else self.nSamples
            flags = FLAG_ADD_BG if (white_bg or bool(torch.rand((1,)) < 0.5)) else 0
            use_graph = bool(graph) and all_reduce is None
            return fs.step(rays, rgb_target, depth_target, N, flags, w_depth, w_trans, delta, tv, use_graph, all_reduce)
        if any(not p.is_leaf for p in params):
            raise T2NError("train_step needs the kernels' own field shape (the parameters ARE the kernel tensors); embedded shapes, "
                           "TensorVM and TensorCP train through the autograd form (OctreeRender_trilinear_fast + loss.backward())")
        dev = self.basis_mat.weight.device
        R = rays.shape[0]
        N = int(N_samples) if N_samples > 0 else self.nSamples
        # rays, the jitter draw (CPU generator, like the reference) and the two targets travel as ONE staged host -> device copy
        rays, jitter, rgb_t, dep_t = to_device_async_many([rays, torch.rand(R, 1), rgb_target, depth_target], dev)
        if rays.dtype != torch.float32 or not rays.is_contiguous():
            rays = rays.contiguous().float()
        jitter = jitter.reshape(-1).contiguous()
        flags = FLAG_TRAIN | (FLAG_ADD_BG if (white_bg or bool(torch.rand((1,)) < 0.5)) else 0)
        rgb_t = rgb_t.contiguous().float()
        dep_t = dep_t.contiguous().float()
        # TV terms: the gradient buffer starts as the TV gradient (one write-only pass on a side stream beside the forward) instead of
        # being zero-filled, accumulated into and TV-incremented after the backward
        seed_ev = seed_terms = None
        if tv and R >= _SEED_MIN_RAYS and getattr(optimizer, "field", None) is self and getattr(self, "defer_factor_grads", False) \
                and self.supports_deferred_factor_grads() and (all_reduce is None or all_reduce_averages):
            seed_ev = self.seed_factor_grads_with_tv(tv)
            seed_terms, tv = list(tv), ()
        device_rows = False
        if speculative:
            device_rows = self._poll_device_rows() and bool(self.keep_activation_rows) and int(getattr(self, "_ctx_rows_hint", 0)) > 0 \
                and self.shadingMode == "MLP_Fea_noview" and not self.mlp_exact_fp32
        head = params[12:]
        if getattr(self, "_head_flat", None) is None or self._head_flat.numel() != sum(p.numel() for p in head):
            self._head_flat = torch.zeros(sum(p.numel() for p in head), device=dev)
        with torch.no_grad():
            rgb, depth, z, w, ws = self._render_raw(rays, N, flags, jitter, True, keep_ctx=True, reuse_ctx=True)
            d_rgb, d_depth, d_w = torch.empty_like(rgb), torch.empty_like(depth), torch.empty_like(w)
            losses = torch.empty(4, device=dev)
            lws = torch.empty(int(lib.t2n_train_loss_workspace_bytes(R)), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.t2n_train_loss(_lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(w), _lib.ptr(z), _lib.ptr(rgb_t), _lib.ptr(dep_t), R, N,
                                              float(w_depth), float(w_trans), float(delta), _lib.ptr(d_rgb), _lib.ptr(d_depth), _lib.ptr(d_w),
                                              _lib.ptr(losses), _lib.ptr(lws), lws.numel(), _lib.current_stream_ptr(dev)), "t2n_train_loss")
            # head gradients: views of one flat buffer (one memset), installed as .grad of the 7 head tensors
            self._head_flat.zero_()
            views, off = [], 0
            for p in head:
                views.append(self._head_flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            if seed_ev is not None:
                torch.cuda.current_stream(dev).wait_event(seed_ev)
            try:
                grads = self._backward_raw(rays, jitter, N, flags, ws, d_rgb, d_depth, d_w, head_grads=views, device_rows=device_rows)
            except T2NError as e:
                # the C side refuses the device-side plan where its scatters do not apply (checked before anything is queued):
                # the counted route, as documented
                if not device_rows or "DEVICE_ROWS" not in str(e):
                    raise
                device_rows = False
                grads = self._backward_raw(rays, jitter, N, flags, ws, d_rgb, d_depth, d_w, head_grads=views, device_rows=False)
            if device_rows:
                self.device_rows_steps = getattr(self, "device_rows_steps", 0) + 1
                self._device_rows_issued = getattr(self, "_device_rows_issued", 0) + 1
            for p, g in zip(params, grads):
                p.grad = g
            if all_reduce is not None:
                all_reduce()
            optimizer.step(tv=tv) if tv else optimizer.step()
            if seed_terms is not None:
                self.seed_factor_grads_with_tv(seed_terms, ahead=True)
            if speculative:
                # nothing in the step waits for the GPU any more: the host is held to two steps of run-ahead (a queue that grows
                # without bound stalls in the HIP runtime for milliseconds at a time: 7-ms pauses every ~11 steps measured)
                evs = self.__dict__.setdefault("_spec_events", [])
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                evs.append(ev)
                if len(evs) > 2:
                    evs.pop(0).synchronize()
        return losses

    def _can_fuse_train_step(self, optimizer):
        return (getattr(optimizer, "field", None) is self and self.supports_deferred_factor_grads() and self.shadingMode == "MLP_Fea_noview"
                and not self.mlp_exact_fp32 and self.alphaMask is None and int(self.app_dim) == 27 and int(self.fea_pe) == 6
                and int(self.featureC) == 128 and list(self.density_n_comp) == [16] * 3 and list(self.app_n_comp) == [48] * 3
                and (int(self.nSamples) <= 1024))

    def _poll_device_rows(self):
        """What the speculative steps have recorded so far (k_bwd_plan writes every step's row NEED and an overflow count into pinned
        host memory: t2n_field_device_rows_record — a plain host read, never waits, and no record is ever lost, it is late at worst).
        The capacity hint becomes 1.25 x the largest need of the last eight records. Returns False when an overflow has been recorded
        since the last poll: the next step takes the counted route."""
        if not getattr(self, "device_rows_steps", 0) or self._handle is None:
            return True
        rec = (C.c_uint32 * 10)()
        _lib.check(_lib.load().t2n_field_device_rows_record(self._handle, rec), "t2n_field_device_rows_record")
        if rec[9] < getattr(self, "_device_rows_issued", 0):
            self.device_rows_unanswered = getattr(self, "device_rows_unanswered", 0) + 1     # (records still on their way)
        # margin over the recorded need: 1.25, widened by a quarter with every recorded overflow (a row count growing faster than the
        # record travels), narrowed again over 64 clean polls
        margin = getattr(self, "_device_rows_margin", 1.25)
        seen = getattr(self, "_device_rows_ovf_seen", 0)
        overflowed = rec[8] > seen
        if overflowed:
            self.device_rows_overflows = getattr(self, "device_rows_overflows", 0) + int(rec[8] - seen)
            self._device_rows_ovf_seen = int(rec[8])
            margin = min(2.0, margin * 1.25)
            self._device_rows_clean = 0
        else:
            self._device_rows_clean = getattr(self, "_device_rows_clean", 0) + 1
            if self._device_rows_clean >= 64:
                margin, self._device_rows_clean = max(1.25, margin / 1.1), 0
        self._device_rows_margin = margin
        need = max(rec[0:8])
        if need:
            self._ctx_rows_hint = (int(need * margin) + 95) // 32 * 32
        return not overflowed

    def stats(self):
        """Counters of the last render call (one device->host copy): evaluated / appearance samples, overflow."""
        s = self.last_stats.cpu().tolist()
        if s[_lib.STAT_OVERFLOW]:
            raise T2NError("appearance list overflow — workspace sizing bug")
        return {"evaluated": s[_lib.STAT_EVALUATED], "appearance": s[_lib.STAT_APPEARANCE], "f16_redo": s[_lib.STAT_F16_REDO],
                "list_retry": s[_lib.STAT_LIST_RETRY]}


class TensorVMSplit(TensorBase):
    """models/tensoRF.py:139-304: three plane / line factor pairs per quantity, stored as separate tensors."""

    def __init__(self, aabb, gridSize, device, **kargs):
        super().__init__(aabb, gridSize, device, **kargs)

    # ---- containers (models/tensoRF.py:144-160) -------------------------------------------------------------------
    def init_svd_volume(self, res, device):
        self.density_plane, self.density_line = self.init_one_svd(self.density_n_comp, self.gridSize, 0.1, device)
        self.app_plane, self.app_line = self.init_one_svd(self.app_n_comp, self.gridSize, 0.1, device)
        self.basis_mat = nn.Linear(sum(self.app_n_comp), self.app_dim, bias=False).to(device)

    def init_one_svd(self, n_component, gridSize, scale, device):
        planes, lines = [], []
        g = [int(x) for x in gridSize]
        for i in range(3):
            m0, m1 = MAT_MODE[i]
            planes.append(nn.Parameter(scale * torch.randn((1, n_component[i], g[m1], g[m0]))))
            lines.append(nn.Parameter(scale * torch.randn((1, n_component[i], g[VEC_MODE[i]], 1))))
        return nn.ParameterList(planes).to(device), nn.ParameterList(lines).to(device)

    # ---- optimiser / regulariser surface ----------------------------------------------------------------------------
    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):
        """models/tensoRF.py:164-174: same groups, same order."""
        groups = [{"params": self.density_line, "lr": lr_init_spatialxyz},
                  {"params": self.density_plane, "lr": lr_init_spatialxyz},
                  {"params": self.app_line, "lr": lr_init_spatialxyz},
                  {"params": self.app_plane, "lr": lr_init_spatialxyz},
                  {"params": self.basis_mat.parameters(), "lr": lr_init_network}]
        if isinstance(self.renderModule, nn.Module):
            groups += [{"params": self.renderModule.parameters(), "lr": lr_init_network}]
        return groups

    def TV_loss_density(self, reg):
        """models/tensoRF.py:193-197. A TVLoss-like `reg` on device planes runs as two HIP kernels per plane (losses.tv_planes)."""
        from .losses import tv_planes
        fused = tv_planes(reg, list(self.density_plane), 1e-2)
        if fused is not None:
            return fused
        total = 0
        for p in self.density_plane:
            total = total + reg(p) * 1e-2
        return total

    def TV_loss_app(self, reg):
        """models/tensoRF.py:199-203."""
        from .losses import tv_planes
        fused = tv_planes(reg, list(self.app_plane), 1e-2)
        if fused is not None:
            return fused
        total = 0
        for p in self.app_plane:
            total = total + reg(p) * 1e-2
        return total

    def density_L1(self):
        total = 0
        for p, l in zip(self.density_plane, self.density_line):
            total = total + torch.mean(torch.abs(p)) + torch.mean(torch.abs(l))
        return total

    def vectorDiffs(self, vector_comps):
        total = 0
        for v in vector_comps:
            n_comp, n_size = v.shape[1:-1]
            m = v.view(n_comp, n_size)
            dotp = m @ m.transpose(-1, -2)
            total = total + torch.mean(torch.abs(dotp.view(-1)[1:].view(n_comp - 1, n_comp + 1)[..., :-1]))
        return total

    def vector_comp_diffs(self):
        return self.vectorDiffs(self.density_line) + self.vectorDiffs(self.app_line)

    def compute_densityfeature(self, xyz_sampled):
        """models/tensoRF.py:205-220 — xyz already normalised to [-1,1]."""
        return self._density_at(xyz_sampled, want_sigma=False)

    def compute_appfeature(self, xyz_sampled):
        """models/tensoRF.py:223-239."""
        return self.shade(xyz_sampled, want_rgb=False)[0]

    @torch.no_grad()
    def up_sampling_VM(self, plane_coef, line_coef, res_target):
        """models/tensoRF.py:258-272: bilinear (align_corners=True) resize of the three planes and lines."""
        lib = _lib.load()
        res = [int(x) for x in res_target]

        def resize(t, Hout, Wout):
            t = t.detach().contiguous().float()
            _, Cn, Hin, Win = t.shape
            out = torch.empty((1, Cn, Hout, Wout), device=t.device, dtype=torch.float32)
            with torch.cuda.device(t.device):
                _lib.check(lib.t2n_upsample_bilinear(_lib.ptr(t), Cn, Hin, Win, _lib.ptr(out), Hout, Wout,
                                                     _lib.current_stream_ptr(t.device)), "t2n_upsample_bilinear")
            return out

        for i in range(3):
            m0, m1 = MAT_MODE[i]
            plane_coef[i] = nn.Parameter(resize(plane_coef[i].data, res[m1], res[m0]))
            line_coef[i] = nn.Parameter(resize(line_coef[i].data, res[VEC_MODE[i]], 1))
        return plane_coef, line_coef

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        """models/tensoRF.py:274-280."""
        self.app_plane, self.app_line = self.up_sampling_VM(self.app_plane, self.app_line, res_target)
        self.density_plane, self.density_line = self.up_sampling_VM(self.density_plane, self.density_line, res_target)
        self._drop_handle()
        self.update_stepSize(res_target)
        print(f"upsamping to {res_target}")

    @torch.no_grad()
    def _voxel_window(self, new_aabb):
        """Voxel range [lo, hi) per axis that covers ``new_aabb`` and the box that range spans — what both ``shrink`` forms need.
        Same fp32 arithmetic as models/tensoRF.py:283-287,297-303 (the lower index is rounded twice there, the upper one once, then
        made exclusive and clipped to the grid; the box is re-derived from the indices by linear interpolation between the old corners)
        unless the alpha mask lives on this very grid, in which case the requested box is kept (:296)."""
        box = torch.as_tensor(new_aabb, dtype=torch.float32).to(self.aabb.device)
        origin, far = self.aabb[0], self.aabb[1]
        lo = torch.round(torch.round((box[0] - origin) / self.units)).long()
        hi = torch.minimum(torch.round((box[1] - origin) / self.units).long() + 1, self.gridSize)
        mask_on_this_grid = self.alphaMask is not None and bool(torch.all(self.alphaMask.gridSize.to(self.gridSize.device) == self.gridSize))
        if not mask_on_this_grid:
            steps = self.gridSize - 1
            f_lo, f_hi = lo / steps, (hi - 1) / steps
            box = torch.stack([(1 - f_lo) * origin + f_lo * far, (1 - f_hi) * origin + f_hi * far])
        return lo, hi, box

    def _adopt_window(self, lo, hi, box):
        self.aabb = box
        self._drop_handle()
        self.update_stepSize(tuple(int(n) for n in (hi - lo)))

    def shrink(self, new_aabb):
        """models/tensoRF.py:282-320: crop every factor to the voxel range covering new_aabb (host index arithmetic + tensor slicing;
        nothing to compute on the device)."""
        lo, hi, box = self._voxel_window(new_aabb)

        def window(t, *axes):   # keep [lo, hi) along the grid axes `axes` of a [1, C, ...] factor (last listed axis = innermost)
            idx = [slice(None), slice(None)] + [slice(int(lo[a]), int(hi[a])) for a in axes]
            return nn.Parameter(t.data[tuple(idx)].contiguous())

        for i in range(3):
            m0, m1 = MAT_MODE[i]
            for planes, lines in ((self.density_plane, self.density_line), (self.app_plane, self.app_line)):
                planes[i] = window(planes[i], m1, m0)
                lines[i] = nn.Parameter(lines[i].data[:, :, int(lo[VEC_MODE[i]]):int(hi[VEC_MODE[i]]), :].contiguous())
        self._adopt_window(lo, hi, box)


class TensorVM(TensorVMSplit):
    """models/tensoRF.py:4-136: the stacked-coefficient VM field — ``plane_coef [3, A + D, res, res]`` and
    ``line_coef [3, A + D, res, 1]`` with the appearance components first (``[:, :A]``) and the density components last
    (``[:, -D:]``) — on the SAME kernels as TensorVMSplit: each of the 12 factor tensors the C-ABI wants is a contiguous
    slice of the two stacked tensors, so the native field reads the parameters in place and the backward writes straight
    into slices of two dense gradient tensors. ``density_n_comp`` / ``appearance_n_comp`` are scalars as in upstream
    TensoRF (the reference class cannot be built from this driver's list-valued options) and must be 16 / 48 here; the
    grid is cubic (``res = gridSize[0]``, models/tensoRF.py:9-14)."""

    def __init__(self, aabb, gridSize, device, **kargs):
        d, a = int(kargs.pop("density_n_comp", 16)), int(kargs.pop("appearance_n_comp", 48))
        if len(set(int(g) for g in gridSize)) != 1:
            raise T2NError("TensorVM stores res x res planes for all three pairs: the grid must be cubic")
        super().__init__(aabb, gridSize, device, density_n_comp=[d] * 3, appearance_n_comp=[a] * 3, **kargs)
        self._d, self._a = d, a

    def compute_features(self, xyz_sampled):
        """models/tensoRF.py:24-44: ``(sigma_feature [n], app_features [n, app_dim])`` at normalised points, both through the gather kernels."""
        return self.compute_densityfeature(xyz_sampled), self.compute_appfeature(xyz_sampled)

    def init_svd_volume(self, res, device):
        res = int(res)
        c = self.app_n_comp[0] + self.density_n_comp[0]
        self.plane_coef = nn.Parameter(0.1 * torch.randn((3, c, res, res), device=device))
        self.line_coef = nn.Parameter(0.1 * torch.randn((3, c, res, 1), device=device))
        self.basis_mat = nn.Linear(self.app_n_comp[0] * 3, self.app_dim, bias=False).to(device)

    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):
        groups = [{"params": self.line_coef, "lr": lr_init_spatialxyz}, {"params": self.plane_coef, "lr": lr_init_spatialxyz},
                  {"params": self.basis_mat.parameters(), "lr": lr_init_network}]
        if isinstance(self.renderModule, nn.Module):
            groups += [{"params": self.renderModule.parameters(), "lr": lr_init_network}]
        return groups

    def _mlp_params(self):
        if not isinstance(self.renderModule, nn.Module):
            return []
        m = self.renderModule.mlp
        return [m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias]

    def _autograd_params(self):
        return [self.plane_coef, self.line_coef, self.basis_mat.weight] + self._mlp_params()

    def _kernel_views(self, tensors):
        """[plane, line, basis, mlp...] (parameters or gradients) -> the 19 kernel-order tensors (contiguous slices)."""
        plane, line = tensors[0], tensors[1]
        a = self.app_n_comp[0]
        den_p = [plane[k, a:].unsqueeze(0) for k in range(3)]
        den_l = [line[k, a:].unsqueeze(0) for k in range(3)]
        app_p = [plane[k, :a].unsqueeze(0) for k in range(3)]
        app_l = [line[k, :a].unsqueeze(0) for k in range(3)]
        return den_p + den_l + app_p + app_l + list(tensors[2:])

    def _all_params(self):
        return self._kernel_views([p.detach() if not torch.is_grad_enabled() else p for p in self._autograd_params()])

    def get_kwargs(self):
        kw = super().get_kwargs()
        kw.update({"density_n_comp": self.density_n_comp[0], "appearance_n_comp": self.app_n_comp[0]})
        return kw

    def vector_comp_diffs(self):
        a, d = self.app_n_comp[0], self.density_n_comp[0]
        return self.vectorDiffs(self.line_coef[:, -d:]) + self.vectorDiffs(self.line_coef[:, :a])

    def vectorDiffs(self, vector_comps):
        total = 0
        for idx in range(len(vector_comps)):
            n_comp, n_size = vector_comps[idx].shape[:-1]
            m = vector_comps[idx].view(n_comp, n_size)
            dotp = torch.matmul(m, m.transpose(-1, -2))
            total = total + torch.mean(torch.abs(dotp.view(-1)[1:].view(n_comp - 1, n_comp + 1)[..., :-1]))
        return total

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        """models/tensoRF.py:126-136: bilinear (align_corners=True) resize of the stacked tensors — the planes by the scale factor
        res_target[0] / res (output size floor(res * scale), as F.interpolate computes it), the lines to res_target[0] — through the
        ATen-exact resize kernel. The reference then calls an undefined ``compute_stepSize`` (an AttributeError upstream); what it
        evidently means, ``update_stepSize``, is what runs here."""
        lib = _lib.load()
        res = int(self.line_coef.shape[2])
        scale = res_target[0] / res                      # python floats, like the reference
        out_p = int(math.floor(float(res) * scale))      # F.interpolate(scale_factor=...): floor(input_size * scale_factor)
        out_l = int(res_target[0])

        def resize(t, Hout, Wout):
            t = t.detach().contiguous().float()
            n, c, Hin, Win = t.shape
            out = torch.empty((n, c, Hout, Wout), device=t.device, dtype=torch.float32)
            with torch.cuda.device(t.device):
                _lib.check(lib.t2n_upsample_bilinear(_lib.ptr(t), n * c, Hin, Win, _lib.ptr(out), Hout, Wout,
                                                     _lib.current_stream_ptr(t.device)), "t2n_upsample_bilinear")
            return out

        if self.plane_coef.device.type != "cuda":
            raise T2NError("TensorVM.upsample_volume_grid runs on the MI355X only (no CPU fallback)")
        if out_p != out_l:
            raise T2NError(f"TensorVM.upsample_volume_grid: planes would become {out_p}^2 but lines {out_l} (res_target must be "
                           "reachable by the scale factor, as in the reference)")
        self.plane_coef = nn.Parameter(resize(self.plane_coef.data, out_p, out_p))
        self.line_coef = nn.Parameter(resize(self.line_coef.data, out_l, 1))
        self._drop_handle()
        self.update_stepSize([out_l] * 3)
        print(f"upsamping to {res_target}")

    def _unsupported(self, *a, **k):
        raise T2NError("TensorVM has no shrink / TV / L1 terms in the reference either (models/tensoRF.py:4-136 defines none)")

    shrink = TV_loss_density = TV_loss_app = density_L1 = _unsupported


class TensorCP(TensorVMSplit):
    """models/tensoRF.py:306-434: the CP (line-only) field, sigma = sum_c Lz[c](z) Ly[c](y) Lx[c](x), rendered by the VM-split
    kernels through an exact algebraic embedding: bilinear interpolation of the rank-one table Ly (x) Lx IS the product of
    the two linear interpolations, so the field equals a VM field with plane_0 = Ly (x) Lx, line_0 = Lz and pairs 1, 2
    zero (likewise for appearance; basis_mat [app_dim, A] becomes [basis | 0 | 0]). The virtual VM tensors are built with
    differentiable torch ops, so autograd carries the kernels' plane / line gradients back to the six line parameters.
    "Supported cheaply" (SURVEY.md 8a footnote): two thirds of the gathers read zeros; n_comp must be 16 / 48."""

    def __init__(self, aabb, gridSize, device, **kargs):
        super().__init__(aabb, gridSize, device, **kargs)

    def init_svd_volume(self, res, device):
        if self.density_n_comp[0] != 16 or self.app_n_comp[0] != 48:
            raise T2NError("TensorCP on the HIP kernels needs density_n_comp[0] == 16 and appearance_n_comp[0] == 48")
        self.density_line = self.init_one_svd(self.density_n_comp[0], self.gridSize, 0.2, device)
        self.app_line = self.init_one_svd(self.app_n_comp[0], self.gridSize, 0.2, device)
        self.basis_mat = nn.Linear(self.app_n_comp[0], self.app_dim, bias=False).to(device)
        self._virt, self._virt_key, self._virt_grad = None, None, False

    def init_one_svd(self, n_component, gridSize, scale, device):
        g = [int(x) for x in gridSize]
        return nn.ParameterList([nn.Parameter(scale * torch.randn((1, n_component, g[VEC_MODE[i]], 1))) for i in range(3)]).to(device)

    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):
        groups = [{"params": self.density_line, "lr": lr_init_spatialxyz}, {"params": self.app_line, "lr": lr_init_spatialxyz},
                  {"params": self.basis_mat.parameters(), "lr": lr_init_network}]
        if isinstance(self.renderModule, nn.Module):
            groups += [{"params": self.renderModule.parameters(), "lr": lr_init_network}]
        return groups

    def _leaves(self):
        ps = list(self.density_line) + list(self.app_line) + [self.basis_mat.weight]
        if isinstance(self.renderModule, nn.Module):
            m = self.renderModule.mlp
            ps += [m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias]
        return ps

    def _embed(self, lines):
        """lines[i] is along VEC_MODE[i] = (z, y, x): pair 0 = plane over (x, y) [1,C,gy,gx] x line over z."""
        lz, ly, lx = lines[0], lines[1], lines[2]
        plane0 = ly[0, :, :, 0][:, :, None] * lx[0, :, :, 0][:, None, :]          # [C, gy, gx]
        return plane0.unsqueeze(0).contiguous(), lz

    def _virtual(self):
        leaves = self._leaves()
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in leaves)
        key = tuple((p.data_ptr(), p._version) for p in leaves)
        if self._virt is not None and key == self._virt_key and (self._virt_grad or not grad):
            return self._virt     # (backward runs with grad mode off: it must see the forward's tensors, not rebuild them)
        g = [int(x) for x in self.gridSize]
        dev = self.basis_mat.weight.device

        def zeros_pairs(C):
            planes = [None] + [torch.zeros(1, C, g[MAT_MODE[k][1]], g[MAT_MODE[k][0]], device=dev) for k in (1, 2)]
            lines = [None] + [torch.zeros(1, C, g[VEC_MODE[k]], 1, device=dev) for k in (1, 2)]
            return planes, lines
        with torch.set_grad_enabled(grad):
            dp, dl = zeros_pairs(self.density_n_comp[0])
            ap, al = zeros_pairs(self.app_n_comp[0])
            dp[0], dl[0] = self._embed(self.density_line)
            ap[0], al[0] = self._embed(self.app_line)
            A = self.app_n_comp[0]
            basis = torch.cat([self.basis_mat.weight, torch.zeros(self.app_dim, 2 * A, device=dev)], 1).contiguous()
            virt = dp + dl + ap + al + [basis] + self._leaves()[7:]
        self._virt, self._virt_key, self._virt_grad = virt, key, grad
        return virt

    def _all_params(self):
        return self._virtual()

    def _autograd_params(self):
        return self._virtual()

    def state_dict(self, *a, **k):
        return super().state_dict(*a, **k)

    def density_L1(self):
        return sum(torch.mean(torch.abs(l)) for l in self.density_line)

    def TV_loss_density(self, reg):
        return sum(reg(l) * 1e-3 for l in self.density_line)

    def TV_loss_app(self, reg):
        return sum(reg(l) * 1e-3 for l in self.app_line)

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        """models/tensoRF.py:381-385 (lines only), through the same ATen-exact resize kernel as the VM variant."""
        dummy_p = [torch.zeros(1, 1, 2, 2, device=self.basis_mat.weight.device) for _ in range(3)]
        for name in ("density_line", "app_line"):
            lines = getattr(self, name)
            _, new = self.up_sampling_VM([p.clone() for p in dummy_p], lines, [max(int(r), 2) for r in res_target])
            setattr(self, name, new)
        self._virt = None
        self._drop_handle()
        self.update_stepSize(res_target)

    @torch.no_grad()
    def shrink(self, new_aabb):
        """models/tensoRF.py:387-416: crop the six lines to the voxel range covering new_aabb (host index arithmetic + slicing). The
        reference dereferences ``self.alphaMask`` unconditionally; without a mask the aabb is corrected to the voxel range here, as
        it is when the mask's grid differs from the field's."""
        lo, hi, box = self._voxel_window(new_aabb)
        for i in range(3):
            a = VEC_MODE[i]
            for lines in (self.density_line, self.app_line):
                lines[i] = nn.Parameter(lines[i].data[:, :, int(lo[a]):int(hi[a]), :].contiguous())
        self._virt = None
        self._adopt_window(lo, hi, box)


class _GeneralFn(torch.autograd.Function):
    """Autograd bridge of the general-shape path: t2n_generic_forward / t2n_generic_backward; the parameters are the module's own
    leaf tensors (reference layouts), the gradients come back in those layouts."""

    @staticmethod
    def forward(ctx, field, rays, N, flags, jitter, *params):
        rgb, depth, z, w, ws = field._general_forward(rays, N, flags, jitter)
        ctx.field, ctx.N, ctx.flags, ctx.ws = field, N, flags, ws
        ctx.key = tuple((p.data_ptr(), p._version) for p in params)
        ctx.save_for_backward(rays, jitter, w, z)
        ctx.mark_non_differentiable(z)
        return rgb, depth, z, w

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_z, d_w):
        field = ctx.field
        rays, jitter, w, z = ctx.saved_tensors
        if tuple((p.data_ptr(), p._version) for p in field._real_params()) != ctx.key:
            raise T2NError("parameters changed between forward and backward")
        grads = field._general_backward(rays, jitter, ctx.N, ctx.flags, ctx.ws, w, z, d_rgb, d_depth, d_w)
        ctx.ws = None
        return (None, None, None, None, None) + tuple(grads)


class _RenderFn(torch.autograd.Function):
    """Autograd bridge: forward = t2n_render_forward (context kept in a private workspace), backward =
    t2n_render_backward. Gradients come back in the reference parameter layouts, so torch.optim / TVLoss see ordinary
    dense .grad tensors."""

    @staticmethod
    def forward(ctx, field, rays, N, flags, jitter, *params):
        rgb, depth, z, w, ws = field._render_raw(rays, N, flags, jitter, True, keep_ctx=True)
        ctx.field, ctx.N, ctx.flags, ctx.ws = field, N, flags | FLAG_KEEP_CTX, ws
        ctx.key = field._uploaded_key
        ctx.save_for_backward(rays, jitter)
        ctx.mark_non_differentiable(z)
        return rgb, depth, z, w

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_z, d_w):
        field = ctx.field
        rays, jitter = ctx.saved_tensors
        if field._uploaded_key != ctx.key:
            raise T2NError("parameters changed between forward and backward")
        grads = field._backward_raw(rays, jitter, ctx.N, ctx.flags, ctx.ws, d_rgb, d_depth, d_w)
        ctx.ws = None
        return (None, None, None, None, None) + tuple(grads)
