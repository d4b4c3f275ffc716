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


class _MultiPixelLossFn(torch.autograd.Function):
    """bin_model.get_loss's whole arithmetic as ONE autograd node (round 5): T terms of one criterion and their mean in two
    launches, every gradient in one — instead of a pair of launches and an autograd node per term plus ~100 scalar ATen kernels
    for `sum(loss_list) / len(loss_list)` and its backward, all of which sat between the forward and the backward pass with the
    chip idle.  Same reductions, same rounding order: bit-identical to the per-term path (tests/test_gpu_loss.py)."""

    @staticmethod
    def forward(ctx, kind, eps, index_pairs, *tensors):
        ts = [t.contiguous().float() for t in tensors]
        pairs = [(ts[i], ts[j]) for i, j in index_pairs]
        loss, terms = ops.multi_pixel_loss(kind, pairs, eps)
        ctx.save_for_backward(*ts)
        ctx.kind, ctx.eps, ctx.index_pairs = kind, eps, list(index_pairs)
        ctx.mark_non_differentiable(terms)
        return loss, terms

    @staticmethod
    def backward(ctx, g, _gterms):
        ts = ctx.saved_tensors
        pairs = [(ts[i], ts[j]) for i, j in ctx.index_pairs]
        where = {}
        for term, (i, j) in enumerate(ctx.index_pairs):
            where.setdefault(i, []).append((term, 1.0))
            where.setdefault(j, []).append((term, -1.0))
        need = [k for k in range(len(ts)) if ctx.needs_input_grad[3 + k] and k in where]
        grads = [None] * len(ts)
        # one launch covers every tensor that sits in one or two terms (all of bin_model's); further pairs of terms are added
        chunks = {k: [where[k][i:i + 2] for i in range(0, len(where[k]), 2)] for k in need}
        # (one launch writes at most LOSS_MAX_TERMS gradients: 13-24 pairs with both sides needing one exceed that — advisor r05)
        for i0 in range(0, len(need), L.LOSS_MAX_TERMS):
            part = need[i0:i0 + L.LOSS_MAX_TERMS]
            for k, o in zip(part, ops.multi_pixel_loss_grad(ctx.kind, pairs, g, [(ts[k], chunks[k][0]) for k in part], ctx.eps)):
                grads[k] = o
        for k in need:
            for w in chunks[k][1:]:
                grads[k] = grads[k] + ops.multi_pixel_loss_grad(ctx.kind, pairs, g, [(ts[k], w)], ctx.eps)[0]
        return (None, None, None, *grads)


def multi_term_loss(criterion, pairs):
    """(loss, [terms]) of `criterion` over the (x, y) `pairs`, loss = sum(terms) / len(terms): fused when `criterion` is one of this
    module's (they carry `.kind`), the plain per-term loop otherwise (an injected criterion; tensors of different sizes)."""
    import os
    kind = getattr(criterion, "kind", None)
    if os.environ.get("BIN_AMD_FUSED_LOSS", "1") == "0":          # diagnostics / A-B only: the per-term path it replaced
        kind = None
    # fused only when it means what the per-term path means: every pair of ONE shape (equal numel with different shapes, or two
    # devices, raised there and must not be summed silently here — advisor r05)
    x0 = pairs[0][0]
    fusable = (kind is not None and len(pairs) <= L.LOSS_MAX_TERMS
               and all(t.is_cuda and t.device == x0.device and t.shape == x0.shape for p in pairs for t in p))
    if not fusable:
        terms = [criterion(x, y) for x, y in pairs]
        return sum(terms) / len(terms), terms
    uniq, index = [], {}
    idx_pairs = []
    for x, y in pairs:
        ij = []
        for t in (x, y):
            if id(t) not in index:
                index[id(t)] = len(uniq)
                uniq.append(t)
            ij.append(index[id(t)])
        idx_pairs.append(tuple(ij))
    loss, terms = _MultiPixelLossFn.apply(kind, getattr(criterion, "eps", 0.0), tuple(idx_pairs), *uniq)
    return loss, list(terms.unbind(0))


class CharbonnierLoss(nn.Module):
    """mean(sqrt((x-y)^2 + eps)); eps is NOT squared, as in the reference."""

    kind = L.LOSS_CHARBONNIER

    def __init__(self, eps=1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, x, y):
        return _PixelLossFn.apply(x, y, L.LOSS_CHARBONNIER, self.eps)


class L1SumLoss(nn.Module):
    """sum |x - y|  (= nn.L1Loss(reduction='sum'), bin_model.py:55)."""
    kind, eps = L.LOSS_L1_SUM, 0.0

    def forward(self, x, y):
        return _PixelLossFn.apply(x, y, L.LOSS_L1_SUM, 0.0)


class L2SumLoss(nn.Module):
    """sum (x - y)^2  (= nn.MSELoss(reduction='sum'), bin_model.py:57)."""
    kind, eps = L.LOSS_L2_SUM, 0.0

    def forward(self, x, y):
        return _PixelLossFn.apply(x, y, L.LOSS_L2_SUM, 0.0)
