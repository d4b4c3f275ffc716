"""Pixel criteria of bin_model (reference models/bin_model.py:52-60) on the HIP reduction kernels, with backward:
CharbonnierLoss (reference models/loss.py:130-141), and the sum-reduced L1 / L2 losses the option 'pixel_criterion'
can select instead (`nn.L1Loss(reduction='sum')`, `nn.MSELoss(reduction='sum')` in the reference)."""
import torch
import torch.nn as nn

from .. import ops
from .. import _lib as L


class _PixelLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, kind, eps):
        ctx.save_for_backward(x, y)
        ctx.kind, ctx.eps = kind, eps
        return ops.pixel_loss(kind, x, y, eps)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gx = ops.pixel_loss_grad(ctx.kind, x, y, g, ctx.eps)
        return (gx if ctx.needs_input_grad[0] else None,
                -gx if ctx.needs_input_grad[1] else None, None, None)


class CharbonnierLoss(nn.Module):
    """mean(sqrt((x-y)^2 + eps)); eps is NOT squared, as in the reference."""

    def __init__(self, eps=1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, x, y):
        return _PixelLossFn.apply(x, y, L.LOSS_CHARBONNIER, self.eps)


class L1SumLoss(nn.Module):
    """sum |x - y|  (= nn.L1Loss(reduction='sum'), bin_model.py:55)."""

    def forward(self, x, y):
        return _PixelLossFn.apply(x, y, L.LOSS_L1_SUM, 0.0)


class L2SumLoss(nn.Module):
    """sum (x - y)^2  (= nn.MSELoss(reduction='sum'), bin_model.py:57)."""

    def forward(self, x, y):
        return _PixelLossFn.apply(x, y, L.LOSS_L2_SUM, 0.0)
