"""Optional fused regulariser + optimiser step: ``TVAdam`` applies the total-variation gradient of the
VM planes and torch.optim.Adam's update with two streaming HIP kernels per tensor instead of ~60 eager torch ops.

Drop-in use next to the reference's loop (text2nerf_main.py:453-454,577-590)::

    optimizer = TVAdam(tensorf.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    ...
    total_loss = loss + weight_depth_loss * depth_loss + weight_trans_loss * trans_loss      # no TV terms in the graph
    optimizer.zero_grad(); total_loss.backward()
    optimizer.step(tv=[(tensorf.density_plane, TV_weight_density), (tensorf.app_plane, TV_weight_app)])

which is numerically the reference's ``total_loss += TV_loss_density(tvreg) * w_d + TV_loss_app(tvreg) * w_a`` followed
by ``Adam.step()`` (same update formula, fp32; verified against torch in tests/test_hip_parity.py).
"""
from __future__ import annotations

import torch

from . import _lib


def _bump_version(t):
    try:
        torch._C._increment_version([t])
    except Exception:      # older torch: an in-place no-op does the same
        t.add_(0)


class TVAdam(torch.optim.Optimizer):
    """``field=tensorf`` (a TensorVMSplit with fp32 factor storage, single process): the 12 plane / line tensors are stepped
    where their data already is — the backward leaves their gradients in the library's channel-last buffers
    (``tensorf.defer_factor_grads``: ``.grad`` of those parameters stays None; several backward calls before a step — a batch
    rendered in chunks, gradient accumulation — add up in ``tensorf.factor_grad_buffer()``, which ``step`` / ``zero_grad`` zero), and
    ``t2n_field_tv_adam_step`` applies TV + Adam on the device's channel-last copies, writing the new values into the
    nn.Parameters as well. Same per-element arithmetic; the Adam moments of those tensors live channel-last in
    ``state[p]["exp_avg_cl"]`` / ``["exp_avg_sq_cl"]``. Without ``field`` every tensor takes the reference-layout kernels."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, field=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.field = None
        if field is not None:
            if not field.supports_deferred_factor_grads():
                raise _lib.T2NError("TVAdam(field=...): needs a TensorVMSplit with fp32 factor storage")
            self.field = field
            field.defer_factor_grads = True

    def _step_factors_on_device(self, tv, betas, eps):
        """TV + Adam of the field's 12 factor tensors from the device-side gradients; returns the ids of the tensors handled."""
        import ctypes as C
        f = self.field
        ps = f._all_params()
        fac = ps[:12]
        if getattr(f, "_deferred_grad_key", None) is None or f._deferred_grad_key != f._uploaded_key:
            raise _lib.T2NError("TVAdam(field=...): no device-side factor gradients for the current parameters "
                                "(call backward, with tensorf.defer_factor_grads still set, before every step)")
        f.factor_grad_buffer()
        import torch.distributed as dist
        # (opt out with field.require_reduced_grads = False: ranks that train independent scenes, a caller that reduces the buffer
        # itself, or a non-default process group)
        if getattr(f, "require_reduced_grads", True) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 \
                and not getattr(f, "_gbuf_reduced", False):
            raise _lib.T2NError("TVAdam(field=...): more than one rank — all-reduce the device-side factor gradients first "
                                "(parallel.allreduce_gradients(params, field=tensorf)); stepping from local gradients would let the "
                                "ranks drift apart (field.require_reduced_grads = False turns this check off)")
        tv_d = tv_a = 0.0
        for planes, weight in tv:
            if planes is f.density_plane:
                tv_d = float(weight) * 1e-2
            elif planes is f.app_plane:
                tv_a = float(weight) * 1e-2
            else:
                raise _lib.T2NError("TVAdam(field=...): tv entries must be tensorf.density_plane / tensorf.app_plane")
        lr_of = {id(p): float(g["lr"]) for g in self.param_groups for p in g["params"]}
        lrs, ms, vs = [], [], []
        for p in fac:
            if id(p) not in lr_of:
                raise _lib.T2NError("TVAdam(field=...): every plane / line tensor of the field must be in a parameter group")
            st = self.state[p]
            if "exp_avg_cl" not in st:
                st["step"] = 0
                st["exp_avg_cl"] = torch.zeros(p.numel(), device=p.device, dtype=torch.float32)
                st["exp_avg_sq_cl"] = torch.zeros(p.numel(), device=p.device, dtype=torch.float32)
            st["step"] += 1
            lrs.append(lr_of[id(p)]); ms.append(st["exp_avg_cl"]); vs.append(st["exp_avg_sq_cl"])
        dev = fac[0].device
        # the pointer arrays only change when a moment tensor is re-created: built once, re-used every step (host-bound loops)
        key = tuple(t.data_ptr() for t in ms) + tuple(t.data_ptr() for t in vs)
        cache = getattr(self, "_cl_args", None)
        if cache is None or cache[0] != key:
            VP = C.c_void_p * 12
            cache = self._cl_args = (key, VP(*[t.data_ptr() for t in ms]), VP(*[t.data_ptr() for t in vs]), (C.c_float * 12)(), (C.c_int64 * 12)())
        _, ms_arr, vs_arr, lr_arr, step_arr = cache
        for i, p in enumerate(fac):
            lr_arr[i] = lrs[i]
            step_arr[i] = int(self.state[p]["step"])
        with torch.cuda.device(dev):
            pst = f._param_struct([p.detach() for p in ps])
            _lib.check(_lib.load().t2n_field_tv_adam_step(f._handle, C.byref(pst), ms_arr, vs_arr, lr_arr, step_arr, float(betas[0]),
                                                          float(betas[1]), eps, tv_d, tv_a, _lib.current_stream_ptr(dev)),
                       "t2n_field_tv_adam_step")
        for p in fac:
            _bump_version(p)
        f._device_factor_key = tuple((p.data_ptr(), p._version) for p in fac)   # the device copies are these values
        f.zero_factor_grads(lazy=True)   # consumed: zeroed in front of the next accumulating backward (or seeded by train_step)
        return {id(p) for p in fac}

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)
        if self.field is not None:
            self.field.zero_factor_grads()

    @torch.no_grad()
    def step(self, tv=()):
        lib = _lib.load()
        done = set()
        if self.field is not None and self.field.defer_factor_grads:
            g0 = self.param_groups[0]
            done = self._step_factors_on_device(tv, tuple(g0["betas"]), float(g0["eps"]))
            tv = ()
        # 1. TV gradient of the listed plane groups: TV_loss_* = sum_planes 1e-2 * TVLoss(plane) (models/tensoRF.py:193-203)
        for planes, weight in tv:
            if weight == 0:
                continue
            for p in planes:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                _, C, H, W = p.shape
                with torch.cuda.device(p.device):
                    _lib.check(lib.t2n_tv_grad_add(_lib.ptr(p), _lib.ptr(p.grad), C, H, W, float(weight) * 1e-2,
                                                   _lib.current_stream_ptr(p.device)), "t2n_tv_grad_add")
        # 2. Adam: every parameter tensor in one launch (same betas / eps across groups, per-tensor lr and step)
        import ctypes as C
        ps, lrs, steps = [], [], []
        betas, eps, dev = None, None, None
        for group in self.param_groups:
            if betas is None:
                betas, eps = tuple(group["betas"]), float(group["eps"])
            elif tuple(group["betas"]) != betas or float(group["eps"]) != eps:
                raise _lib.T2NError("TVAdam: all parameter groups must share betas and eps")
            for p in group["params"]:
                if p.grad is None or id(p) in done:
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise _lib.T2NError("TVAdam needs contiguous float32 parameters and gradients on the GPU")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                ps.append(p); lrs.append(float(group["lr"])); steps.append(int(st["step"]))
                dev = p.device
        if not ps:
            return
        n = len(ps)
        VP = C.c_void_p * n
        arr = lambda f: VP(*[f(p) for p in ps])      # noqa: E731
        with torch.cuda.device(dev):
            _lib.check(lib.t2n_adam_step_multi(n, arr(lambda p: p.data_ptr()), arr(lambda p: p.grad.data_ptr()),
                                               arr(lambda p: self.state[p]["exp_avg"].data_ptr()),
                                               arr(lambda p: self.state[p]["exp_avg_sq"].data_ptr()),
                                               (C.c_int64 * n)(*[p.numel() for p in ps]), (C.c_float * n)(*lrs), float(betas[0]),
                                               float(betas[1]), eps, (C.c_int64 * n)(*steps), _lib.current_stream_ptr(dev)),
                       "t2n_adam_step_multi")
        for p in ps:
            _bump_version(p)   # changed in place through a raw pointer: the field keys its re-upload on _version
