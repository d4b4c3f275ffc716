"""CharbonnierLoss (reference models/loss.py:130-141) on the HIP reduction kernels, with backward."""
import torch
import torch.nn as nn

from .. import ops


class _CharbonnierFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, eps):
        ctx.save_for_backward(x, y)
        ctx.eps = eps
        return ops.charbonnier(x, y, eps)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gx = ops.charbonnier_grad(x, y, g, ctx.eps)
        return (gx if ctx.needs_input_grad[0] else None,
                -gx if ctx.needs_input_grad[1] else None, None)


class CharbonnierLoss(nn.Module):
    """mean(sqrt((x-y)^2 + eps)); eps is NOT squared, as in the reference."""

    def __init__(self, eps=1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, x, y):
        return _CharbonnierFn.apply(x, y, self.eps)
