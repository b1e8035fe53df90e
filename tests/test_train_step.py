"""The autograd-free training step (TensorVMSplit.train_step: render -> t2n_train_loss -> t2n_render_backward -> TVAdam) against the
autograd form of the same step (text2nerf_main.py:547-590 restated with torch ops: MSE(rgb) + 0.005 MSE(depth) + 1e3 masked
transmittance, TV terms in TVAdam), the fused loss kernel against torch's loss and autograd gradients, and deferred factor gradients
that add up over several backward calls of one step (a batch rendered in chunks)."""
import numpy as np
import pytest
import torch

from text2nerf_amd import synth
from tests.conftest import TINY
from tests.test_hip_parity import dev, make_field

pytestmark = pytest.mark.gpu


def batch(seed=3, n=384):
    g = np.random.Generator(np.random.PCG64(seed))
    rays = torch.from_numpy(synth.frame_rays_np(16, 24, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0))))[:n]
    rgb_t = torch.from_numpy(g.uniform(0, 1, (rays.shape[0], 3)).astype(np.float32))
    dep_t = torch.from_numpy(g.uniform(2, 7, (rays.shape[0],)).astype(np.float32))
    return rays, rgb_t, dep_t


def torch_loss(rgb, depth, w, z, rgb_t, dep_t):
    from text2nerf_amd.losses import TransMittanceLoss_mask
    depth = torch.where(torch.isnan(depth), torch.zeros_like(depth), depth)
    mse = torch.mean((rgb - rgb_t) ** 2)
    dl = torch.mean((depth - dep_t) ** 2)
    tl = TransMittanceLoss_mask()(w, (z - dep_t[:, None] + 0.1) < 0)
    return mse, dl, tl, mse + 0.005 * dl + 1e3 * tl


def test_fused_loss_kernel_matches_torch_loss_and_autograd():
    import ctypes as C
    from text2nerf_amd import _lib
    lib = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(5)
    R, N = 301, 37
    rgb = torch.rand(R, 3, generator=g).to(d).requires_grad_(True)
    depth = (torch.rand(R, generator=g) * 6 + 1).to(d)
    depth[7] = float("nan")
    depth.requires_grad_(True)
    w = (torch.rand(R, N, generator=g) * 0.05).to(d).requires_grad_(True)
    z = (torch.rand(R, N, generator=g).sort(1)[0] * 8).to(d)
    rgb_t, dep_t = torch.rand(R, 3, generator=g).to(d), (torch.rand(R, generator=g) * 5 + 2).to(d)
    mse, dl, tl, tot = torch_loss(rgb, depth, w, z, rgb_t, dep_t)
    tot.backward()
    d_rgb, d_depth, d_w, losses = torch.empty(R, 3, device=d), torch.empty(R, device=d), torch.empty(R, N, device=d), torch.empty(4, device=d)
    ws = torch.empty(int(lib.t2n_train_loss_workspace_bytes(R)), dtype=torch.uint8, device=d)
    with torch.cuda.device(d):
        _lib.check(lib.t2n_train_loss(_lib.ptr(rgb.detach()), _lib.ptr(depth.detach()), _lib.ptr(w.detach()), _lib.ptr(z), _lib.ptr(rgb_t), _lib.ptr(dep_t),
                                      R, N, 0.005, 1e3, 0.1, _lib.ptr(d_rgb), _lib.ptr(d_depth), _lib.ptr(d_w), _lib.ptr(losses), _lib.ptr(ws),
                                      ws.numel(), _lib.current_stream_ptr(d)), "t2n_train_loss")
    ref = torch.stack([mse, dl, tl, tot]).detach()
    assert torch.allclose(losses, ref, rtol=2e-6, atol=1e-9), (losses, ref)
    for got, want in ((d_rgb, rgb.grad), (d_depth, depth.grad), (d_w, w.grad)):
        assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-12
    assert float(d_depth[7]) == 0.0        # the NaN depth got no gradient


def assert_same_trajectory(fa, fb, steps, lr_spatial=0.02, lr_network=1e-3, frac=2e-3):
    """Parameters of two fields after `steps` Adam steps from the same start. The factor gradients are sums of floating-point
    atomics (order differs run to run) and Adam divides by sqrt(v): an element whose gradient is a near-cancelling sum can move a
    visibly different fraction of a step, so the bound is stated in Adam's own unit — `frac` of one learning-rate step per
    iteration — instead of in ulps of the parameter. A wrong loss weight, a dropped TV term or a lost chunk moves elements by
    whole steps (lr), 500x this bound."""
    for (k, a), (_, b) in zip(fa.state_dict().items(), fb.state_dict().items()):
        lr = lr_spatial if ("plane" in k or "line" in k) else lr_network
        diff = float((a - b).abs().max())
        assert diff <= frac * lr * steps, (k, diff, frac * lr * steps)
        # and the bulk of the elements agrees far more tightly than the worst one
        assert float((a - b).abs().mean()) <= 2e-5 * lr * steps + 1e-9, k


def step_autograd(f, opt, rays, rgb_t, dep_t, chunk=None):
    from text2nerf_amd import OctreeRender_trilinear_fast
    d = dev()
    rgb, _, depth, w, z = OctreeRender_trilinear_fast(rays, f, chunk=chunk or rays.shape[0], N_samples=-1, white_bg=True, is_train=True, device=d)
    mse, dl, tl, tot = torch_loss(rgb, depth, w, z, rgb_t.to(d), dep_t.to(d))
    opt.zero_grad()
    tot.backward()
    return torch.stack([mse, dl, tl, tot]).detach()


@pytest.mark.parametrize("form", ["composed-seeded", "composed", "fused", "fused-graph"])
def test_train_step_equals_the_autograd_step(tiny_params, form, monkeypatch):
    """Every form of train_step against the autograd form of the reference's step (text2nerf_main.py:547-601), with the learning rates
    and the TV weights decaying from step to step like the driver's (lr_factor, :577-583,597-598).
    composed(-seeded): render -> loss kernel -> backward -> TVAdam as separate calls (`seeded`: the TV terms written into the gradient
    buffer on a side stream in front of the backward — what that form does for batches of 8 192 rays and more, forced here);
    fused: ONE C call (t2n_train_step: nothing read on the host, every per-step scalar from device memory); fused-graph: that call
    captured once per input buffer and replayed (4 captures, then hipGraphLaunch) — the changed learning rates / TV weights reach the
    replayed kernels through device memory."""
    from text2nerf_amd import tensorf as tf
    from text2nerf_amd.optim import TVAdam
    monkeypatch.setattr(tf, "_SEED_MIN_RAYS", 0 if form == "composed-seeded" else 1 << 30)
    kw = {"composed-seeded": dict(fused=False), "composed": dict(fused=False), "fused": dict(fused=True, graph=False),
          "fused-graph": dict(fused=True, graph=True)}[form]
    rays, rgb_t, dep_t = batch()
    fa = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    fb = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    oa = TVAdam(fa.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fa)
    ob = TVAdam(fb.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fb)
    steps = 7
    for it in range(steps):
        w_d, w_a = 0.1 * 0.9 ** it, 0.01 * 0.8 ** it
        for o in (oa, ob):
            for g in o.param_groups:
                g["lr"] = g["lr"] * 0.97
        torch.manual_seed(100 + it)            # the jitter comes from the CPU default generator in both forms
        la = step_autograd(fa, oa, rays, rgb_t, dep_t)
        oa.step(tv=[(fa.density_plane, w_d), (fa.app_plane, w_a)])
        torch.manual_seed(100 + it)
        lb = fb.train_step(rays, rgb_t, dep_t, ob, N_samples=-1, white_bg=True, tv=[(fb.density_plane, w_d), (fb.app_plane, w_a)], **kw).clone()
        assert torch.allclose(la, lb, rtol=1e-5, atol=1e-9), (it, la, lb)
    fs = fb.__dict__.get("_fused_step")
    if form.startswith("fused"):
        fs.sync()
        assert fs.replays == 0 and fs.issued == steps
        assert (fs.graph_launches, fs.graph_captures) == ((steps - 1, 4) if form == "fused-graph" else (0, 0))
        if form == "fused-graph":
            assert fs.graph_nodes <= 40        # the captured step: kernels of one iteration (43 launches + torch fills in round 5's composed form)
        assert all(int(ob.state[p]["step"]) == steps for p in fb.parameters())
    else:
        assert fs is None
    assert_same_trajectory(fa, fb, steps=steps)


def test_fused_step_withholds_an_update_whose_rows_do_not_fit(tiny_params):
    """never apply a truncated gradient. A fused step whose appearance rows exceed its capacity applies NOTHING — the 19
    tensors, their Adam moments and the step count stay bit-identical — the host learns it from the pinned record (without waiting),
    counts it and submits the same batch again with a capacity that holds it; the trajectory is the one of the steps without the
    incident."""
    from text2nerf_amd.optim import TVAdam
    rays, rgb_t, dep_t = batch()
    fa = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    fb = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    oa = TVAdam(fa.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fa)
    ob = TVAdam(fb.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fb)
    tv = lambda f: [(f.density_plane, 0.1), (f.app_plane, 0.01)]   # noqa: E731
    for it in range(2):
        torch.manual_seed(40 + it)
        fa.train_step(rays, rgb_t, dep_t, oa, white_bg=True, tv=tv(fa), fused=True, graph=False)
        torch.manual_seed(40 + it)
        fb.train_step(rays, rgb_t, dep_t, ob, white_bg=True, tv=tv(fb), fused=True, graph=False)
    fs = fb._fused_step
    fs.sync()
    assert max(fs.needs) > 1024
    before = {k: v.detach().clone() for k, v in fb.state_dict().items()}
    moments = [t.clone() for p in fb.parameters() for t in (ob.state[p].get("exp_avg_cl", ob.state[p].get("exp_avg")),)]
    fs.cap_once = 256                          # far below what the batch needs
    torch.manual_seed(42)
    l = fb.train_step(rays, rgb_t, dep_t, ob, white_bg=True, tv=tv(fb), fused=True, graph=False).clone()
    torch.cuda.synchronize()
    # the withheld step: nothing moved (bit for bit), the loss of the batch is still reported
    assert bool(torch.isfinite(l).all())
    for k, v in fb.state_dict().items():
        assert torch.equal(v, before[k]), k
    for t0, p in zip(moments, fb.parameters()):
        assert torch.equal(t0, ob.state[p].get("exp_avg_cl", ob.state[p].get("exp_avg")))
    import ctypes as C
    from text2nerf_amd import _lib
    rec = (C.c_uint32 * 36)()
    _lib.check(_lib.load().t2n_field_train_record(fb._handle, rec), "t2n_field_train_record")
    assert (rec[1], rec[2]) == (2, 1)          # Adam steps applied / withheld, on the device
    torch.manual_seed(42)
    fa.train_step(rays, rgb_t, dep_t, oa, white_bg=True, tv=tv(fa), fused=True, graph=False)
    # the next call reads the record, replays the withheld batch with room for it, then runs its own
    torch.manual_seed(43)
    fa.train_step(rays, rgb_t, dep_t, oa, white_bg=True, tv=tv(fa), fused=True, graph=False)
    torch.manual_seed(43)
    fb.train_step(rays, rgb_t, dep_t, ob, white_bg=True, tv=tv(fb), fused=True, graph=False)
    fs.sync()
    assert fs.replays == 1 and fb.device_rows_overflows == 1 and fs.rows_cap > 1024
    assert all(int(ob.state[p]["step"]) == 4 for p in fb.parameters())
    _lib.check(_lib.load().t2n_field_train_record(fb._handle, rec), "t2n_field_train_record")
    assert (rec[1], rec[2]) == (4, 1)
    assert_same_trajectory(fa, fb, steps=4)


def test_fused_step_in_two_phases_equals_the_single_call(tiny_params):
    """Data-parallel form: render + loss + backward (phase 1), the caller's all-reduce of the factor gradient buffer and the head gradients
    (+ the vote word behind them), the optimiser (phase 2). With an identity all-reduce: the single call's trajectory."""
    from text2nerf_amd.optim import TVAdam
    rays, rgb_t, dep_t = batch()
    fa = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    fb = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    oa = TVAdam(fa.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fa)
    ob = TVAdam(fb.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fb)
    seen = []
    def all_reduce():
        g = fb.factor_grad_buffer()
        seen.append((float(g.abs().max()), float(fb.basis_mat.weight.grad.abs().max()), float(fb._fused_step.head_grads[-1])))
    for it in range(4):
        torch.manual_seed(7 + it)
        la = fa.train_step(rays, rgb_t, dep_t, oa, white_bg=True, tv=[(fa.density_plane, 0.1), (fa.app_plane, 0.01)], fused=True, graph=False).clone()
        torch.manual_seed(7 + it)
        lb = fb.train_step(rays, rgb_t, dep_t, ob, white_bg=True, tv=[(fb.density_plane, 0.1), (fb.app_plane, 0.01)], fused=True, all_reduce=all_reduce).clone()
        assert torch.allclose(la, lb, rtol=1e-6, atol=1e-9)
    assert len(seen) == 4 and all(a > 0 and b > 0 and v == 0.0 for a, b, v in seen)   # gradients were there, nobody voted to withhold
    fb._fused_step.sync()
    assert_same_trajectory(fa, fb, steps=4)


def test_speculative_step_equals_the_counted_step(tiny_params):
    """train_step(speculative=True): the backward reads no row count on the host (T2N_FLAG_DEVICE_ROWS: capacity from the previous
    step, kernels clip on the device) — same trajectory as the counted step, and the route really was taken."""
    from text2nerf_amd.optim import TVAdam
    rays, rgb_t, dep_t = batch()
    fa = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    fb = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    oa = TVAdam(fa.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fa)
    ob = TVAdam(fb.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fb)
    for it in range(6):
        torch.manual_seed(300 + it)
        la = fa.train_step(rays, rgb_t, dep_t, oa, N_samples=-1, white_bg=True, tv=[(fa.density_plane, 0.1), (fa.app_plane, 0.01)], fused=False)
        torch.manual_seed(300 + it)
        lb = fb.train_step(rays, rgb_t, dep_t, ob, N_samples=-1, white_bg=True, tv=[(fb.density_plane, 0.1), (fb.app_plane, 0.01)],
                           speculative=True)
        assert torch.allclose(la, lb, rtol=1e-5, atol=1e-9), (it, la, lb)
    assert getattr(fb, "device_rows_steps", 0) == 5 and getattr(fb, "device_rows_overflows", 0) == 0     # (step 0 learns the capacity)
    assert getattr(fa, "device_rows_steps", 0) == 0
    assert_same_trajectory(fa, fb, steps=6)


def test_speculative_step_survives_a_capacity_overflow(tiny_params, monkeypatch):
    """A step whose appearance rows exceed the capacity: the rows past it lose their appearance gradient (nothing else), the host
    reads it from the pinned record (late at worst), counts it, and the next step takes the counted route; the capacity follows the
    recorded need."""
    from text2nerf_amd import tensorf as tf
    from text2nerf_amd.optim import TVAdam
    monkeypatch.setattr(tf, "_ladder", lambda n, *a, **k: n)       # (the ladder's floor of 1024 rows would hide the small capacity)
    rays, rgb_t, dep_t = batch()
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    o = TVAdam(f.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=f)
    step = lambda: f.train_step(rays, rgb_t, dep_t, o, N_samples=-1, white_bg=True, speculative=True)
    torch.manual_seed(1)
    step()
    full = int(f._ctx_rows_hint)
    assert full > 256
    f._ctx_rows_hint = 64                      # far below what the batch needs
    torch.manual_seed(2)
    l1 = step()
    assert f.device_rows_steps == 1
    torch.cuda.synchronize()                   # (the record of that step has arrived: the test must not depend on the host's lead)
    torch.manual_seed(3)
    l2 = step()                                # reads the record: overflow -> counted route, capacity from the recorded need
    assert f.device_rows_overflows == 1 and f.device_rows_steps == 1
    assert int(f._ctx_rows_hint) > 256
    torch.manual_seed(4)
    step()
    assert f.device_rows_steps == 2
    torch.cuda.synchronize()
    assert bool(torch.isfinite(l1).all()) and bool(torch.isfinite(l2).all())
    assert all(bool(torch.isfinite(p).all()) for p in f.parameters())


def test_speculative_takes_the_counted_route_where_the_device_plan_does_not_apply(tiny_params):
    """The device-side row plan serves the fused split-f16 head only: with the strict-fp32 head `speculative=True` is the counted step
    (no error, no device-rows call), and the C-ABI refuses the flag outright."""
    import ctypes as C
    from text2nerf_amd import _lib
    from text2nerf_amd.optim import TVAdam
    rays, rgb_t, dep_t = batch()
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    f.mlp_exact_fp32 = True
    o = TVAdam(f.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=f)
    for it in range(3):
        torch.manual_seed(it)
        l = f.train_step(rays, rgb_t, dep_t, o, N_samples=-1, white_bg=True, speculative=True)
    assert getattr(f, "device_rows_steps", 0) == 0 and bool(torch.isfinite(l).all())
    # the flag itself on that field: T2N_ERR_UNSUPPORTED, nothing launched
    d = dev()
    r = rays.to(d)
    N = f.nSamples
    jit = torch.rand(r.shape[0], device=d)
    with torch.no_grad():
        rgb, depth, z, w, ws = f._render_raw(r, N, _lib.FLAG_TRAIN | _lib.FLAG_ADD_BG, jit, True, keep_ctx=True)
    with pytest.raises(_lib.T2NError, match="DEVICE_ROWS"):
        f._backward_raw(r, jit, N, _lib.FLAG_TRAIN | _lib.FLAG_ADD_BG, ws, torch.ones_like(rgb), torch.zeros_like(depth), None, device_rows=True)


def test_deferred_factor_gradients_add_up_over_chunks(tiny_params):
    """with TVAdam(field=...) a batch larger than `chunk` makes several backward nodes; every one of them must land in
    the factor gradients (round 1 zeroed the buffer per backward call and kept only the last chunk's). Reference: the same chunked
    step with reference-layout gradients (TVAdam without field: autograd accumulates .grad across the nodes)."""
    from text2nerf_amd.optim import TVAdam
    rays, rgb_t, dep_t = batch(n=320)
    f1 = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    f2 = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    o1 = TVAdam(f1.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))                 # reference-layout gradients
    o2 = TVAdam(f2.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=f2)       # deferred, channel-last
    for it in range(2):
        torch.manual_seed(9 + it)
        step_autograd(f1, o1, rays, rgb_t, dep_t, chunk=96)           # 4 chunks -> 4 backward nodes
        o1.step(tv=[(f1.density_plane, 0.1), (f1.app_plane, 0.01)])
        torch.manual_seed(9 + it)
        step_autograd(f2, o2, rays, rgb_t, dep_t, chunk=96)
        assert float(f2.factor_grad_buffer().abs().max()) > 0
        o2.step(tv=[(f2.density_plane, 0.1), (f2.app_plane, 0.01)])
        assert float(f2.factor_grad_buffer().abs().max()) == 0.0      # consumed and zeroed by the step
    assert_same_trajectory(f1, f2, steps=2)


def test_tv_terms_through_hip_kernels_match_torch_tvloss(tiny_params):
    """TV_loss_density / TV_loss_app with a TVLoss-like regulariser run as HIP kernels (losses.tv_planes): value and gradients
    against the plain torch evaluation of the same module (utils.py:488-504, models/tensoRF.py:193-203), with a non-unit upstream
    gradient like the driver's TV weights (text2nerf_main.py:577-586)."""
    from text2nerf_amd.losses import TVLoss, tv_planes
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    reg = TVLoss(1.7)
    assert tv_planes(reg, list(f.density_plane), 1e-2) is not None         # the fused route is taken
    for p in f.parameters():
        p.grad = None
    fused = f.TV_loss_density(reg) * 0.1 + f.TV_loss_app(reg) * 0.01
    fused.backward()
    got = {k: p.grad.detach().clone() for k, p in f.named_parameters() if p.grad is not None}
    for p in f.parameters():
        p.grad = None
    plain = sum(reg(p) * 1e-2 for p in f.density_plane) * 0.1 + sum(reg(p) * 1e-2 for p in f.app_plane) * 0.01
    plain.backward()
    assert abs(float(fused) - float(plain)) <= 1e-5 * abs(float(plain)) + 1e-12
    assert set(got) == {k for k, p in f.named_parameters() if p.grad is not None}
    for k, p in f.named_parameters():
        if p.grad is None:
            continue
        scale = float(p.grad.abs().max()) + 1e-20
        assert float((got[k] - p.grad).abs().max()) <= 2e-6 * scale, k
    # a regulariser without TVLoss_weight (any callable) takes the plain route
    assert tv_planes(lambda x: x.sum(), list(f.density_plane), 1e-2) is None


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_optimizer_virtual_ranks(tiny_params, world):
    """Sharded optimiser (t2n_train_step with shard_world / shard_rank, phases 1 / 2 / 4): `world` ranks — threads of this
    process, each with its own field, optimiser state and fused step, exchanging through in-process collectives — against the single-rank
    fused step on the same batches. Checks the kernels' side of the rule: who seeds which TV gradient, which blocks a rank's Adam
    touches (moments of blocks it does not own stay zero), and that gathered blocks reach the reference-layout tensors."""
    import ctypes as C
    import threading
    from text2nerf_amd import _lib
    from text2nerf_amd.optim import TVAdam
    from text2nerf_amd.parallel import shard_layout
    from tests.helpers.virtual_ranks import VirtualExchange, VirtualGroup
    rays, rgb_t, dep_t = batch()
    steps = 3
    mk = lambda: make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    ref = mk()
    oref = TVAdam(ref.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=ref)
    for it in range(steps):
        torch.manual_seed(100 + it)
        ref.train_step(rays, rgb_t, dep_t, oref, N_samples=-1, white_bg=True, tv=[(ref.density_plane, 0.1), (ref.app_plane, 0.01)], fused=True, graph=False)
    ref.__dict__["_fused_step"].sync()
    fields = [mk() for _ in range(world)]
    opts = [TVAdam(f.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=f) for f in fields]
    # the C layout == its Python mirror
    lay = (C.c_int64 * 36)()
    _lib.check(_lib.load().t2n_field_shard_layout(fields[0].sync_params(), world, lay), "t2n_field_shard_layout")
    assert [[int(lay[3 * t + j]) for j in range(3)] for t in range(12)] == shard_layout(TINY["grid"], world)
    vg = VirtualGroup(world)
    exs = [VirtualExchange.make(f, vg, r) for r, f in enumerate(fields)]
    errors = []

    def run(r):
        try:
            torch.cuda.set_device(dev())
            f = fields[r]
            for it in range(steps):
                vg.turn.acquire()
                torch.manual_seed(100 + it)      # every rank the SAME batch and jitter: the average of the ranks' gradients is the batch gradient
                f.train_step(rays, rgb_t, dep_t, opts[r], N_samples=-1, white_bg=True, tv=[(f.density_plane, 0.1), (f.app_plane, 0.01)],
                             fused=True, graph=False, all_reduce=exs[r])
            f.__dict__["_fused_step"].sync()
        except BaseException as e:      # noqa: BLE001
            errors.append((r, repr(e)))
            vg.barrier.abort()
            if vg.turn.locked():
                try:
                    vg.turn.release()
                except RuntimeError:
                    pass

    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not errors, errors
    torch.cuda.synchronize()
    for r, f in enumerate(fields):
        assert_same_trajectory(ref, f, steps)
        # moments: zero exactly on the body blocks of the other ranks (channel-last pooled moments of the fused step)
        fs = f.__dict__["_fused_step"]
        _, ms, vs, _ = fs._moments()
        for t in (0, 1, 2, 6, 7, 8):
            off, sl, total = exs[r].layout[t]
            m = ms[t].reshape(-1)
            for rr in range(world):
                seg = m[rr * sl:(rr + 1) * sl]
                if rr == r:
                    assert float(seg.abs().max()) > 0.0
                else:
                    assert float(seg.abs().max()) == 0.0, (r, t, rr)
            if m[world * sl:].numel():
                assert float(m[world * sl:].abs().max()) > 0.0


@pytest.mark.parametrize("seed", list(range(8)))
def test_fused_step_equals_composed_step_on_random_configurations(seed, forms=(False, True), strict=False):
    """The one-call step (t2n_train_step: folded loss, device-side plan, binned scatters with their staging / prefetch forms, early density
    bins, fused TV + Adam) against the composed step (separate render / loss / backward / TVAdam calls) on random grids — anisotropic,
    9 to 60 texels — boxes, cameras, ray counts (ragged: not multiples of 4 / 32 / 64) and sample counts: three steps from the same
    parameters, the same jitter. What changes from case to case: the number of appearance tiles and density blocks and how full they are
    (segments of a few records up to several batches), partial last batches, rays without any sample in the box."""
    from text2nerf_amd.optim import TVAdam
    g = np.random.Generator(np.random.PCG64(9000 + seed))
    grid = [int(g.integers(9, 60)) for _ in range(3)]
    lo = (-g.uniform(2.0, 9.0, 3)).astype(np.float32)
    hi = g.uniform(2.0, 9.0, 3).astype(np.float32)
    aabb = [lo.tolist(), hi.tolist()]
    near_far = [float(g.uniform(0.05, 1.0)), float(g.uniform(6.0, 14.0))]
    params = synth.make_field_params(9100 + seed, grid, density_scale=float(g.uniform(0.5, 1.4)), aabb=aabb)
    centre = tuple(float(v) for v in (lo + (hi - lo) * g.uniform(0.3, 0.7, 3)))
    H, W = int(g.integers(9, 40)), int(g.integers(9, 40))
    rays = torch.from_numpy(synth.frame_rays_np(H, W, c2w=synth.look_pose(float(g.uniform(-3, 3)), float(g.uniform(-1, 1)), centre)))
    n = int(g.integers(5, rays.shape[0]))
    rays = rays[torch.from_numpy(g.permutation(rays.shape[0])[:n])].contiguous()
    rgb_t = torch.from_numpy(g.uniform(0, 1, (n, 3)).astype(np.float32))
    dep_t = torch.from_numpy(g.uniform(1, 9, (n,)).astype(np.float32))
    N = int(g.integers(16, 120))
    fa = make_field(params, grid, aabb, near_far)
    fb = make_field(params, grid, aabb, near_far)
    oa = TVAdam(fa.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fa)
    ob = TVAdam(fb.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fb)
    steps = 3
    for it in range(steps):
        tv_a = [(fa.density_plane, 0.1), (fa.app_plane, 0.01)]
        tv_b = [(fb.density_plane, 0.1), (fb.app_plane, 0.01)]
        torch.manual_seed(300 + it)
        la = fa.train_step(rays, rgb_t, dep_t, oa, N_samples=N, white_bg=True, tv=tv_a, fused=forms[0], graph=False).clone()
        torch.manual_seed(300 + it)
        lb = fb.train_step(rays, rgb_t, dep_t, ob, N_samples=N, white_bg=True, tv=tv_b, fused=forms[1], graph=False).clone()
        assert torch.allclose(la, lb, rtol=2e-5, atol=1e-9), (seed, it, la, lb)
    for f_, fu in ((fa, forms[0]), (fb, forms[1])):     # (forms: the campaign tool also runs a path against ITSELF, the control)
        if fu:
            f_.__dict__["_fused_step"].sync()
    if strict:
        assert_same_trajectory(fa, fb, steps)
        return
    # In the suite: assert_same_trajectory's bound on ALL BUT A HANDFUL of elements. Two runs of the SAME path differ by more than that bound in
    # a single element in 2-3 % of these random cases (float atomics in a near-cancelling sum, then Adam's division by sqrt(v): the composed step
    # against itself fails 4 of 200 seeds, fused against composed 4 of 200, the same seeds: profiles/round6_fused_step_campaign.txt) — a lost
    # tile, a wrong plane or a dropped term moves thousands of elements by whole learning-rate steps.
    for (k, a), (_, b) in zip(fa.state_dict().items(), fb.state_dict().items()):
        lr = 0.02 if ("plane" in k or "line" in k) else 1e-3
        d = (a - b).abs()
        over = int((d > 2e-3 * lr * steps).sum())
        assert over <= 4, (k, over, float(d.max()))
        assert float(d.mean()) <= 2e-5 * lr * steps + 1e-9, (k, float(d.mean()))
