"""The two small loss terms the training step of text2nerf_main.py:563-586 applies to the renderer's outputs and
parameters (utils.py:67-80, 488-504), restated so the C3-shaped benchmark step and users of the drop-in have them next to
the renderer. Plain torch ops; the fused forms are optim.TVAdam (TV + Adam) and TensorVMSplit.train_step / t2n_train_loss
(the render loss and its gradients in one kernel)."""
import torch
import torch.nn as nn


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
