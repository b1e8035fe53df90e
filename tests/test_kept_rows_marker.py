"""ADVICE r1 (t2n_backward.hip): the backward may use the forward-kept MLP activation rows only when the forward stated that it kept them
with the capacity the backward derives from ITS workspace size. A C-ABI caller that passes the backward a larger buffer than the
forward saw (a pooled allocation) must get the recompute path and correct gradients, not uninitialised activations."""
import ctypes as C

import numpy as np
import pytest
import torch

from text2nerf_amd import _lib, synth
from text2nerf_amd.tensorf import FLAG_KEEP_CTX, FLAG_TRAIN, FLAG_ADD_BG
from tests.conftest import TINY
from tests.test_hip_parity import dev, make_field

pytestmark = pytest.mark.gpu


def grads_with(f, rays, jitter, N, fwd_extra, bwd_extra):
    """raw C-ABI forward + backward: the forward gets its exact context size + fwd_extra bytes, the backward is told the buffer has
    bwd_extra more than that"""
    lib = _lib.load()
    d = dev()
    h = f.sync_params()
    R = rays.shape[0]
    flags = FLAG_TRAIN | FLAG_ADD_BG | FLAG_KEEP_CTX
    base = int(lib.t2n_render_workspace_bytes_ctx(R, N))
    ws = torch.full((base + fwd_extra + bwd_extra,), 0xFF, dtype=torch.uint8, device=d)      # poison: NaN patterns if ever read
    rgb, depth = torch.empty(R, 3, device=d), torch.empty(R, device=d)
    w, z = torch.empty(R, N, device=d), torch.empty(R, N, device=d)
    st = _lib.current_stream_ptr(d)
    with torch.cuda.device(d):
        _lib.check(lib.t2n_render_forward(h, _lib.ptr(rays), R, rays.shape[1], N, flags, _lib.ptr(jitter), _lib.ptr(rgb), _lib.ptr(depth),
                                          _lib.ptr(w), _lib.ptr(z), None, _lib.ptr(ws), base + fwd_extra, st), "fwd")
        rows = C.c_int64(0)
        _lib.check(lib.t2n_render_ctx_rows(_lib.ptr(ws), R, N, st, C.byref(rows)), "rows")
        params = f._autograd_params()
        grads = [torch.zeros_like(p) for p in params]
        gs = f._param_struct(grads, _lib.FieldGrads)
        bws = torch.empty(int(lib.t2n_backward_workspace_bytes(h, rows.value, R, N)), dtype=torch.uint8, device=d)
        d_rgb = torch.ones(R, 3, device=d) / R
        d_depth = torch.zeros(R, device=d)
        _lib.check(lib.t2n_render_backward(h, _lib.ptr(rays), R, rays.shape[1], N, flags, _lib.ptr(jitter), _lib.ptr(d_rgb), _lib.ptr(d_depth),
                                           None, C.byref(gs), _lib.ptr(ws), base + fwd_extra + bwd_extra, _lib.ptr(bws), bws.numel(), st), "bwd")
    torch.cuda.synchronize()
    return grads, int(rows.value)


def test_backward_ignores_unstated_kept_rows(tiny_params):
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    rays = torch.from_numpy(synth.frame_rays_np(12, 16, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0)))).to(dev())
    g = torch.Generator().manual_seed(1)
    jitter = torch.rand(rays.shape[0], generator=g).to(dev())
    N = 40
    ref, rows = grads_with(f, rays, jitter, N, 0, 0)                       # no kept rows: recompute
    assert rows >= 32
    room = 256 + (rows + 64) * 1728
    kept, _ = grads_with(f, rays, jitter, N, room, 0)                      # forward keeps, backward uses them
    lied, _ = grads_with(f, rays, jitter, N, 0, room)                      # backward sees a larger buffer than the forward did
    for a, b, c in zip(ref, kept, lied):
        tol = 2e-5 * float(a.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= tol
        assert bool(torch.isfinite(c).all()) and float((a - c).abs().max()) <= tol
    assert any(float(a.abs().max()) > 0 for a in ref[12:])
