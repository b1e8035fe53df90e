"""Can the fused train step's device part — forward (KEEP_CTX) -> loss kernel -> backward with the device-side row plan
(T2N_FLAG_DEVICE_ROWS) — be captured into a hipGraph and replayed, now that it reads nothing on the host? The optimiser step stays eager
(its step counts and learning rates are by-value kernel arguments). Compares the loss trajectory of graph-replayed steps with eager
speculative steps from the same start and times both.   argv[1]: rays per batch (default 2048)"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from text2nerf_amd import _lib, synth  # noqa: E402
from text2nerf_amd.optim import TVAdam  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = 259
dev = torch.device("cuda:0")
lib = _lib.load()
torch.set_num_threads(2)


def make():
    field = bench.build_field(dev)[0]
    opt = TVAdam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=field)
    return field, opt


poses = bench.reference_poses("local_fixed")
allrays = torch.from_numpy(np.concatenate([synth.frame_rays_np(512, 512, c2w=p) for p in poses])).to(dev)
f0, _ = make()
with torch.no_grad():
    rgb_s, dep_s, _, _ = f0(allrays[::4], white_bg=True, is_train=False, N_samples=N)
allrgb = rgb_s.repeat_interleave(4, 0)[: allrays.shape[0]].clamp(0, 1)
alldep = dep_s.repeat_interleave(4, 0)[: allrays.shape[0]]
del f0
g = np.random.Generator(np.random.PCG64(7))
perm = torch.from_numpy(g.permutation(allrays.shape[0])).to(dev)
STEPS = 60
jit_all = torch.from_numpy(g.uniform(0, 1, (STEPS + 20, B)).astype(np.float32)).to(dev)
tv_of = lambda f: [(f.density_plane, 0.1), (f.app_plane, 0.01)]   # noqa: E731


def idx_of(k):
    return perm[(k * B) % (perm.numel() - B):][:B]


# ---- eager speculative steps (the product path; jitter drawn by train_step on the CPU is replaced by a fixed table here through
# the same internals the graph uses, so both runs see the same numbers)
def device_step(field, rays, rgb_t, dep_t, jitter, views, flags):
    """forward -> loss -> backward (device rows): what the graph captures. Returns the loss vector."""
    with torch.no_grad():
        rgb, depth, z, w, ws = field._render_raw(rays, N, flags, jitter, True, keep_ctx=True, reuse_ctx=True)
        d_rgb, d_depth, d_w = torch.empty_like(rgb), torch.empty_like(depth), torch.empty_like(w)
        losses = torch.empty(4, device=dev)
        lws = torch.empty(int(lib.t2n_train_loss_workspace_bytes(B)), dtype=torch.uint8, device=dev)
        _lib.check(lib.t2n_train_loss(_lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(w), _lib.ptr(z), _lib.ptr(rgb_t), _lib.ptr(dep_t), B, N,
                                      0.005, 1e3, 0.1, _lib.ptr(d_rgb), _lib.ptr(d_depth), _lib.ptr(d_w), _lib.ptr(losses), _lib.ptr(lws),
                                      lws.numel(), _lib.current_stream_ptr(dev)), "t2n_train_loss")
        field._head_flat.zero_()
        grads = field._backward_raw(rays, jitter, N, flags, ws, d_rgb, d_depth, d_w, head_grads=views, device_rows=True)
    return losses, grads


def run(graphed):
    field, opt = make()
    params = field._autograd_params()
    flags = _lib.FLAG_TRAIN | _lib.FLAG_ADD_BG
    # a few product steps first: capacities, workspaces, pinned record, function attributes all exist afterwards
    for k in range(6):
        i = idx_of(k)
        torch.manual_seed(k)
        field.train_step(allrays[i], allrgb[i], alldep[i], opt, N_samples=N, white_bg=True, tv=tv_of(field), speculative=True)
    torch.cuda.synchronize()
    field._poll_device_rows()
    field._ctx_rows_hint = int(field._ctx_rows_hint * 1.6)      # a fixed, generous capacity for the whole run
    head = params[12:]
    views, off = [], 0
    for p in head:
        views.append(field._head_flat[off:off + p.numel()].view_as(p))
        off += p.numel()
    rays_s, rgb_s_, dep_s_, jit_s = torch.empty(B, 6, device=dev), torch.empty(B, 3, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)

    def load(k):
        i = idx_of(k)
        torch.index_select(allrays, 0, i, out=rays_s)
        torch.index_select(allrgb, 0, i, out=rgb_s_)
        torch.index_select(alldep, 0, i, out=dep_s_)
        jit_s.copy_(jit_all[k % jit_all.shape[0]])

    def pre():      # host-driven, eager: head upload after the optimiser, TV seed on the side stream
        field.sync_params()
        ev = field.seed_factor_grads_with_tv(tv_of(field))
        torch.cuda.current_stream(dev).wait_event(ev)

    def post(grads):
        field._deferred_grad_key = field._uploaded_key
        field._gbuf_dirty = True
        for p, g_ in zip(params, grads):
            p.grad = g_
        opt.step()

    graph = None
    if graphed:
        load(6)
        pre()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            losses, grads = device_step(field, rays_s, rgb_s_, dep_s_, jit_s, views, flags)     # (warm-up on the capture stream)
        torch.cuda.current_stream().wait_stream(side)
        post(grads)
        graph = torch.cuda.CUDAGraph()
        load(7)
        pre()
        with torch.cuda.graph(graph, capture_error_mode="relaxed"):
            losses, grads = device_step(field, rays_s, rgb_s_, dep_s_, jit_s, views, flags)
        # (the capture ran nothing: replay it for step 7)
        graph.replay()
        post(grads)
        k0 = 8
    else:
        k0 = 6
    traj = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(k0, STEPS):
        load(k)
        pre()
        if graph is not None:
            graph.replay()
        else:
            losses, grads = device_step(field, rays_s, rgb_s_, dep_s_, jit_s, views, flags)
        post(grads)
        traj.append(losses.clone())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (STEPS - k0) * 1e3
    rec = (C.c_uint32 * 10)()
    lib.t2n_field_device_rows_record(field._handle, rec)
    return dt, torch.stack(traj).cpu(), list(rec)


try:
    dt_e, tr_e, rec_e = run(False)
    print(f"eager  device part + eager optimiser: {dt_e:.4f} ms per step at {B} rays; record {rec_e}")
    dt_g, tr_g, rec_g = run(True)
    print(f"graph replay          + eager optimiser: {dt_g:.4f} ms per step; record {rec_g}")
    n = min(len(tr_e) - 2, len(tr_g))
    a, b = tr_e[2:2 + n, 3], tr_g[:n, 3]
    print("total loss, eager vs graph (first 6):", a[:6].tolist(), b[:6].tolist())
    print("max relative difference of the total loss over %d steps: %.3e" % (n, float(((a - b).abs() / a.abs()).max())))
except Exception as e:  # noqa: BLE001
    import traceback
    traceback.print_exc()
    print("FAILED:", repr(e)[:800])
