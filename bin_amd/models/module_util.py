"""Mirror of the reference's `models/module_util.py:7-52` (`initialize_weights`, `make_layer`, `ResidualBlock_noBN`).

The reference never instantiates these (bin_stage4 builds its own RDB stacks), but `north_star` names the
file, so a user who imports them finds them here with the same names, constructor arguments and `state_dict` keys
(`conv1.weight`, `conv1.bias`, `conv2.weight`, `conv2.bias`).  The two 3x3 convolutions run on the HIP kernels of the conv stacks
(`binhip_conv2d_fwd` with the fused ReLU / residual epilogues; under autograd the differentiable per-op path of
`bin_amd.autograd._ConvFn`).  CUDA tensors only — there is no CPU path.  `flow_warp` (module_util.py:55-81, optical-flow
warping: no caller anywhere in the reference, not a convolution) is not part of the hot path and is not provided.
"""
import torch
import torch.nn as nn
import torch.nn.init as init


def initialize_weights(net_l, scale=1):
    """Kaiming-normal (fan_in) weights times `scale`, zero biases; BatchNorm to (1, 0) — module_util.py:7-25."""
    for net in (net_l if isinstance(net_l, list) else [net_l]):
        for m in net.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                init.kaiming_normal_(m.weight, a=0, mode="fan_in")
                with torch.no_grad():
                    m.weight.mul_(scale)
                    if m.bias is not None:
                        m.bias.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                init.constant_(m.weight, 1)
                init.constant_(m.bias, 0.0)


def make_layer(block, n_layers):
    """`n_layers` fresh instances of `block()` in a Sequential — module_util.py:28-32."""
    return nn.Sequential(*[block() for _ in range(n_layers)])


class ResidualBlock_noBN(nn.Module):
    """x + conv2(relu(conv1(x))), two 3x3 convolutions of `nf` channels — module_util.py:35-52.

    `precision`: "f16x3" (fp32 class, default) or "f16", as for the RDN modules."""

    def __init__(self, nf=64, precision="f16x3"):
        super().__init__()
        self.conv1 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)      # parameter containers: the reference's state_dict keys
        self.conv2 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.precision = precision
        initialize_weights([self.conv1, self.conv2], 0.1)
        self._cache = None

    def _weights(self, nterms):
        from .. import ops
        key = (nterms, self.conv1.weight._version, self.conv2.weight._version, self.conv1.bias._version,
               self.conv2.bias._version, self.conv1.weight.data_ptr(), self.conv2.weight.data_ptr())
        if self._cache is None or self._cache[0] != key:
            self._cache = (key, ops.ConvWeights(self.conv1.weight.detach(), self.conv1.bias.detach(), nterms=nterms),
                           ops.ConvWeights(self.conv2.weight.detach(), self.conv2.bias.detach(), nterms=nterms))
        return self._cache[1], self._cache[2]

    def forward(self, x):
        from .. import ops
        if not x.is_cuda:
            raise RuntimeError("bin_amd: ResidualBlock_noBN runs on the HIP kernels only (CUDA tensors); there is no CPU path")
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            from ..autograd import _ConvFn
            out = torch.relu(_ConvFn.apply(x.float(), self.conv1.weight, self.conv1.bias))
            return x + _ConvFn.apply(out, self.conv2.weight, self.conv2.bias)
        nterms = 3 if self.precision == "f16x3" else 1
        cw1, cw2 = self._weights(nterms)
        xp = ops.nchw_to_planes(x.float().contiguous(), nterms)
        y = ops.conv2d(ops.conv2d(xp, cw1, relu=True), cw2, residual=xp)       # ReLU and the identity add are conv epilogues
        return ops.planes_to_nchw(y, self.conv1.out_channels)
