"""The two small loss terms the training step of text2nerf_main.py:563-586 applies to the renderer's outputs and
parameters (utils.py:67-80, 488-504), restated so the C3-shaped benchmark step and users of the drop-in have them next to
the renderer. Plain torch ops; the fused forms are optim.TVAdam (TV + Adam) and TensorVMSplit.train_step / t2n_train_loss
(the render loss and its gradients in one kernel)."""
import torch
import torch.nn as nn

from . import _lib


class _TVPlaneSum(torch.autograd.Function):
    """sum_p scale * TVLoss_weight * TVLoss(p) over [1,C,H,W] device planes as two HIP kernels per plane (value: t2n_tv_value,
    gradient: t2n_tv_grad_set) instead of ~15 eager elementwise / reduction kernels per plane with their 69-MB temporaries.
    Same arithmetic as TVLoss.forward (utils.py:488-504) with the sums taken in double."""

    _COEF = {}     # (device, plane shapes) -> [P, 2] normalisers 1 / (c (h-1) w), 1 / (c h (w-1)) as a device tensor (float64)

    @staticmethod
    def forward(ctx, weight, *planes):
        lib = _lib.load()
        dev = planes[0].device
        sums = torch.zeros(len(planes), 32, 2, device=dev, dtype=torch.float64)   # T2N_TV_SLOTS slot pairs per plane
        ctx.weight = float(weight)
        ctx.save_for_backward(*planes)
        with torch.cuda.device(dev):
            st = _lib.current_stream_ptr(dev)
            for i, p in enumerate(planes):
                b, c, h, w = p.shape
                _lib.check(lib.t2n_tv_value(_lib.ptr(p.detach()), c, h, w, _lib.ptr(sums[i]), st), "t2n_tv_value")
        key = (str(dev), tuple(tuple(p.shape) for p in planes))
        coef = _TVPlaneSum._COEF.get(key)
        if coef is None:
            coef = torch.tensor([[1.0 / (p.shape[1] * (p.shape[2] - 1) * p.shape[3]), 1.0 / (p.shape[1] * p.shape[2] * (p.shape[3] - 1))]
                                 for p in planes], dtype=torch.float64, device=dev)
            _TVPlaneSum._COEF[key] = coef
        # three small device ops for all planes (the host-bound training loop pays ~10 us per eager launch)
        return ((sums.sum(1) * coef).sum() * (2.0 * ctx.weight)).to(torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        planes = ctx.saved_tensors
        dev = planes[0].device
        grads = []
        up = grad_out.detach().to(torch.float32).reshape(1).contiguous()   # the upstream scalar, read on the device by the kernels
        with torch.cuda.device(dev):
            st = _lib.current_stream_ptr(dev)
            for p in planes:
                b, c, h, w = p.shape
                g = torch.empty_like(p)           # written by the kernel, scaled by the upstream scalar
                _lib.check(lib.t2n_tv_grad_set(_lib.ptr(p.detach()), _lib.ptr(g), c, h, w, ctx.weight, _lib.ptr(up), st), "t2n_tv_grad_set")
                grads.append(g)
        return (None, *grads)


def _is_plain_tvloss(reg):
    """The fused kernels hard-code TVLoss.forward (utils.py:488-504): take them only for that arithmetic — this module's TVLoss,
    the reference's own class (``utils.TVLoss``, matched by qualified name since it cannot be imported here), or a regulariser that
    opts in with ``reg.t2n_fused_tv = True``. A subclass or look-alike with a different forward (L1 TV, masked TV) is called."""
    if getattr(reg, "t2n_fused_tv", False):
        return True
    t = type(reg)
    return t is TVLoss or (t.__name__ == "TVLoss" and t.__module__.split(".")[-1] == "utils")


def tv_planes(reg, planes, scale):
    """sum_p reg(p) * scale for a plain TVLoss `reg` (see _is_plain_tvloss; weight = its float `TVLoss_weight`) on contiguous fp32
    [1,C,H,W] device planes through the HIP kernels; None when that does not apply (the caller then evaluates reg(p) itself)."""
    w = getattr(reg, "TVLoss_weight", None)
    if w is None or not planes or not _is_plain_tvloss(reg):
        return None
    for p in planes:
        if not (p.is_cuda and p.dtype == torch.float32 and p.dim() == 4 and p.shape[0] == 1 and p.shape[2] > 1 and p.shape[3] > 1
                and p.is_contiguous()):
            return None
    return _TVPlaneSum.apply(float(w) * float(scale), *planes)


class TVLoss(nn.Module):
    """Total-variation regulariser on a [B,C,H,W] plane (utils.py:488-504)."""

    def __init__(self, TVLoss_weight=1):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight

    def forward(self, x):
        b, c, h, w = x.shape
        dh = x[:, :, 1:, :] - x[:, :, :-1, :]
        dw = x[:, :, :, 1:] - x[:, :, :, :-1]
        return self.TVLoss_weight * 2 * ((dh * dh).sum() / (c * (h - 1) * w) + (dw * dw).sum() / (c * h * (w - 1))) / b


class TransMittanceLoss_mask(nn.Module):
    """MSE of the masked mean weight per ray against 0 (utils.py:67-80)."""

    def __init__(self, device=None):
        super().__init__()

    def forward(self, transmittance, mask):
        mean_trans = torch.mean(transmittance * mask, dim=1)
        return torch.mean(mean_trans ** 2)
