"""Training path: torch.autograd.Function wrappers whose backward is hand-written HIP
(counterpart of the autograd graph the reference builds through ATen when bin_model.py:140 calls
`l_pix.backward()`).

* `_RdnFn`   — one whole RDN sub-network: forward = binhip_rdn_forward into a PRIVATE workspace that doubles
               as the saved activations (concat-free block buffers); backward = binhip_rdn_backward
               (dgrad on the forward conv kernel with transposed/flipped weights, MFMA wgrad with LDS
               transpose reads, fused ReLU masks / skip adds / PixelShuffle inverse).
* `_ConvLstmFn` — ConvLSTMCell (RDN.py:50-95) forward/backward kernels.
Training precision defaults to "f16x3" (fp32-class gradients, ~5e-6 relative vs torch autograd of the oracle).
Training in "f16" (single fp16 product in forward AND backward) is NOT a supported mode: its forward's ~1e-3 activation error
flips ReLU masks, and individual parameter gradients come out 1-25 % off (tests/test_gpu_train.py only checks it to
2.5e-1).  Since round 4 it is GATED: a differentiable call of an RDN whose precision resolves to "f16" raises unless the
module carries `allow_f16_training = True` (diagnostics / that test) — or none of its parameters requires a gradient (round 5:
input-gradient-only calls warn once instead).  The supported speed/accuracy trade is
`backward_precision = "f16"` (a per-network attribute, bench.py's "mixed") behind the f16x3 forward: exact loss and masks,
~2e-3 relative gradient error.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from .ops import _ptr, _stream, on_device, status_word
from .rdn_plan import c_shape, layer_names, rdn_forward, workspace


# Per-module switches (attributes of the RDN sub-network objects, models/archs/RDN.py::_RDNBase — nothing here is a mutable
# module global, so two models in one process, or two host threads, never share them):
#   module._direct_grads  (the wrappers turn it on around backward() through net.direct_param_grads()): the RDN's
#       weight gradients are written / accumulated by the kernels DIRECTLY into the parameters' .grad buffers
#       (BINHIP_BWD_ACCUMULATE) and autograd gets None for them: the four weight sets are shared by 4/3/2/1 calls, so the
#       default path costs ~1.7 k elementwise adds per step in autograd's AccumulateGrad.  Off by default:
#       torch.autograd.grad() / gradient hooks see the parameter grads only through the regular path.
#   module.backward_precision  ("f16": with an f16x3 (fp32-class) forward, run the RDN backward single-product on the hi
#       planes of the saved activations (BINHIP_BWD_SAVED_X3) — loss and ReLU masks stay exact, gradients carry ~1e-3
#       relative rounding noise, the step is ~1.4x faster.  From network_G.backward_precision /
#       BIN_AMD_BACKWARD_PRECISION; None = same as forward).

# Weight gradients on a side stream, overlapping the backward-data chain (BinRdnBwdPlan.aux_stream; one side stream per
# device, the library orders and joins it with events inside each call).  Per module: `module.wgrad_side_stream`
# (default below: BIN_AMD_WGRAD_STREAM=0 turns it off; bench.py turns it off for its exclusive kernel-timing pass).
def default_wgrad_side_stream():
    return os.environ.get("BIN_AMD_WGRAD_STREAM", "1") != "0"


_aux_streams = {}


def _aux_stream(device):
    s = _aux_streams.get(device.index)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _aux_streams[device.index] = s
    return s


def train_fused_upnet():
    """BIN_AMD_FUSED_UPNET_TRAIN=0: training keeps UPNet's two layers (forward and backward as in rounds 1-5)."""
    return os.environ.get("BIN_AMD_FUSED_UPNET_TRAIN", "1") != "0"


def default_backward_precision():
    return os.environ.get("BIN_AMD_BACKWARD_PRECISION") or None


def train_precision(module):
    from .models.archs.RDN import PRECISIONS
    p = module.precision or os.environ.get("BIN_AMD_TRAIN_PRECISION", "f16x3")
    if PRECISIONS[p] == 1 and not getattr(module, "allow_f16_training", False):
        if not any(q.requires_grad for q in module.parameters()):
            # input-gradient-only use (saliency maps, adversarial examples) with frozen parameters: nothing is being TRAINED; the
            # input gradients carry the single-product mode's few-percent noise — say so once per module, do not refuse (advisor r04)
            if not getattr(module, "_warned_f16_input_grads", False):
                import warnings
                warnings.warn("bin_amd: differentiating through an RDN in precision 'f16' with frozen parameters: input gradients "
                              "are computed with single fp16 products (a few per cent of noise); use 'f16x3' for fp32-class gradients",
                              stacklevel=3)
                module._warned_f16_input_grads = True
            return PRECISIONS[p]
        raise RuntimeError(
            "bin_amd: training with precision 'f16' (one fp16 product in forward and backward) is not a supported mode — its "
            "parameter gradients are only verified to 25 %.  Train in 'f16x3' (the default; set network_G.precision: f16x3 or "
            "leave it unset), optionally with network_G.backward_precision: f16 for the faster mixed mode; 'f16' is the "
            "inference mode (wrap inference in torch.no_grad()).  To train in it regardless (it converged like f16x3 on the synthetic "
            "task of profiles/r06_training_modes.md) set network_G.allow_f16_training: true, or module.allow_f16_training = True.")
    return PRECISIONS[p]


class _RdnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, nterms, n_frames, *args):
        frames = [a.contiguous().float() for a in args[:n_frames]]
        weights = module.kernel_weights(nterms)
        n, _, h, w = frames[0].shape
        lib = L.lib()
        nbytes = lib.binhip_rdn_workspace_bytes(n, h, w, n_frames, nterms, C.byref(c_shape(weights.shape)))
        if nbytes == 0:
            raise RuntimeError(f"bin_amd: unsupported RDN shape N={n} H={h} W={w}")
        saved = torch.empty(nbytes, dtype=torch.uint8, device=frames[0].device)
        # the fused UPNet in training (round 6): forward = one 5x5 convolution + border ring, backward = BINHIP_BWD_FUSED_UPNET + the
        # chain rule from the operator's gradient to UPNet.0 / UPNet.2 (rdn_plan.fused_upnet_weights under autograd)
        fused = (bool(module.plan_flags & L.PLAN_FUSED_UPNET) and train_fused_upnet() and
                 weights.ensure_fused_upnet(train=True) is not None and weights.fused_graph is not None)
        flags = module.plan_flags | L.PLAN_KEEP_ACTS | (L.PLAN_FUSED_UPNET_TRAIN if fused else 0)
        out = rdn_forward(weights, frames, ws=saved, flags=flags, profiler=module.profiler)
        ctx.fused_up = fused
        if module.debug_hook is not None:
            module.debug_hook("forward", module, (n, h, w, n_frames, nterms), saved, {"shape": module.shape})
        ctx.module, ctx.nterms, ctx.n_frames = module, nterms, n_frames
        ctx.saved_ws = saved
        ctx.dims = (n, h, w)
        ctx.param_meta = [(tuple(a.shape), a.device) for a in args[n_frames:]]
        ctx.params = args[n_frames:]
        # the backward-data weights are rebuilt from the CURRENT parameters: like torch's saved-tensor version check,
        # refuse a backward after an in-place update of the weights this forward used (forward -> optimizer.step() ->
        # backward would otherwise silently mix old activations with new weights)
        ctx.param_versions = [a._version for a in args[n_frames:]]
        ctx.weights_gen = module._wgen
        # backward calls still owed to this weight set in the current step (the pyramid shares model1 / model2 / model3
        # between 4-5 / 3 / 2 calls): when the count returns to zero the set's gradients are complete and a data-parallel
        # reducer may start their all-reduce while the rest of the backward is still running (bin_model.FlatGradAllReduce)
        module._bwd_pending = getattr(module, "_bwd_pending", 0) + 1
        return out

    @staticmethod
    def backward(ctx, gout):
        module, nterms, k = ctx.module, ctx.nterms, ctx.n_frames
        n, h, w = ctx.dims
        dev = gout.device
        gout = gout.contiguous().float()
        lib = L.lib()
        if ctx.weights_gen != module._wgen or any(a._version != v for a, v in zip(ctx.params, ctx.param_versions)):
            raise RuntimeError("bin_amd: a parameter of this RDN was modified in place between its forward and its backward "
                               "(optimizer step, load_state_dict, broadcast ...): the gradient would be computed with "
                               "weights the forward did not use")
        nt_bwd = 1 if (module.backward_precision == "f16" and nterms == 3) else nterms
        dgw = module.kernel_weights(nterms).dgrad(module, nt_bwd)
        plan = L.BinRdnBwdPlan()
        plan.N, plan.H, plan.W, plan.n_inputs, plan.nterms = n, h, w, k, nt_bwd
        if nt_bwd != nterms:
            plan.reserved = L.BWD_SAVED_X3
        dgw.fill_plan(plan)
        plan.status = status_word(dev).data_ptr()
        plan.aux_stream = (_aux_stream(dev).cuda_stream
                           if module.wgrad_side_stream and not torch.cuda.is_current_stream_capturing() else None)
        plan.profiler = module.bwd_profiler if module.bwd_profiler else None
        params = ctx.params
        direct = module._direct_grads and all(ctx.needs_input_grad[3 + k:])
        have = False
        if direct:
            states = [p.grad is not None for p in params]
            have = all(states)
            direct = (have or not any(states)) and all(
                p.grad is None or (p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.device == dev)
                for p in params)
        if direct and have:
            grads = [p.grad for p in params]
            plan.reserved |= L.BWD_ACCUMULATE
        else:
            grads = [torch.empty(shape, dtype=torch.float32, device=dev) for shape, _ in ctx.param_meta]
        for i in range(len(grads) // 2):
            plan.dw[i] = grads[2 * i].data_ptr()
            plan.db[i] = grads[2 * i + 1].data_ptr()
        fused = getattr(ctx, "fused_up", False)
        if fused:
            nl, g0 = len(grads) // 2, module.shape[0]
            plan.reserved |= L.BWD_FUSED_UPNET
            up_dw4 = torch.empty((12, g0, 5, 5), dtype=torch.float32, device=dev)
            up_db4 = torch.empty((12,), dtype=torch.float32, device=dev)
            up_dwr = torch.zeros((n, 9, 12, 25, g0), dtype=torch.float32, device=dev)      # (the kernel writes the 12 pairs with ring pixels)
            up_dbr = torch.zeros((n, 9, 12), dtype=torch.float32, device=dev)
            plan.dw[nl], plan.db[nl], plan.dw[nl + 1], plan.db[nl + 1] = (up_dw4.data_ptr(), up_db4.data_ptr(), up_dwr.data_ptr(),
                                                                        up_dbr.data_ptr())
        gins = []
        for i in range(k):
            if ctx.needs_input_grad[3 + i]:
                g = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev)
                plan.gin[i] = g.data_ptr()
                gins.append(g)
            else:
                plan.gin[i] = None
                gins.append(None)
        nbytes = lib.binhip_rdn_backward_workspace_bytes(n, h, w, k, nt_bwd, C.byref(plan.shape))
        ws = workspace(nbytes, dev, key="bwd")
        with on_device(gout):
            L.check(lib.binhip_rdn_backward(C.byref(plan), _ptr(ctx.saved_ws), ctx.saved_ws.numel(), _ptr(gout),
                                            _ptr(ws), ws.numel(), _stream()), "rdn_backward")
        if module.debug_hook is not None:          # tools/fp16_headroom.py, tests: inspect the planes of this call
            module.debug_hook("backward", module, (n, h, w, k, nt_bwd), ws, {"input_grads": any(g is not None for g in gins),
                                                                              "shape": module.shape})
        ctx.saved_ws = None
        if fused:
            # dL/dW_eff -> dL/dW0, dL/db0, dL/dW2, dL/db2: the operators are a bilinear function of the two layers' parameters
            nl = len(grads) // 2
            dW = up_dwr.sum(0)                                                  # ring layout [9, 12, 25, G0]
            dB = up_dbr.sum(0)
            dW[4], dB[4] = up_dw4.reshape(12, g0, 25).permute(0, 2, 1), up_db4
            leaves, Wr, Br = module.kernel_weights(nterms).fused_graph          # built by this weight version's forward
            gup = torch.autograd.grad([Wr, Br], leaves, [dW, dB], retain_graph=True)
            for slot, g in zip((2 * (nl - 2), 2 * (nl - 2) + 1, 2 * (nl - 1), 2 * (nl - 1) + 1), gup):
                if plan.reserved & L.BWD_ACCUMULATE:
                    grads[slot].add_(g.float())
                else:
                    grads[slot].copy_(g.float())
        module._bwd_pending = getattr(module, "_bwd_pending", 1) - 1
        if module._bwd_pending == 0 and direct:
            cb = getattr(module, "_grads_ready_cb", None)
            if cb is not None:
                cb()
        if direct:
            if not have:
                for p, g in zip(params, grads):
                    p.grad = g
            return (None, None, None, *gins, *([None] * len(grads)))
        pgrads = [g if ctx.needs_input_grad[3 + k + i] else None for i, g in enumerate(grads)]
        return (None, None, None, *gins, *pgrads)


def rdn_apply(module, frames):
    """Differentiable call of one RDN sub-network (used by _RDNBase._run when grad is enabled)."""
    nterms = train_precision(module)
    params = dict(module.named_parameters())
    flat = []
    for nm in layer_names(module.shape):
        flat += [params[nm + ".weight"], params[nm + ".bias"]]
    return _RdnFn.apply(module, nterms, len(frames), *frames, *flat)


class _ConvLstmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c_prev, h_prev, weight, bias, forget_bias):
        from . import ops
        state = None if c_prev is None else [c_prev, h_prev]
        h, (c, _) = ops.convlstm_cell(x, state, weight, bias, forget_bias)
        ctx.save_for_backward(x.contiguous().float(), None if c_prev is None else c_prev.contiguous().float(),
                              None if h_prev is None else h_prev.contiguous().float(),
                              weight.detach().contiguous().float(), bias.detach().contiguous().float())
        ctx.fb = float(forget_bias)
        return h, c

    @staticmethod
    def backward(ctx, gh, gc):
        x, cp, hp, w, b = ctx.saved_tensors
        n, _, hh, ww = x.shape
        dev = x.device
        lib = L.lib()
        gh = gh.contiguous().float() if gh is not None else None
        gc = gc.contiguous().float() if gc is not None else None
        if gh is None and gc is None:
            return None, None, None, None, None, None
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gcp = torch.empty_like(x) if (cp is not None and ctx.needs_input_grad[1]) else None
        ghp = torch.empty_like(x) if (hp is not None and ctx.needs_input_grad[2]) else None
        dw = torch.zeros_like(w)
        db = torch.zeros_like(b)
        ws = workspace(lib.binhip_convlstm_bwd_workspace_bytes(n, hh, ww), dev, key="clstm")
        L.check(lib.binhip_convlstm_bwd(_ptr(x), _ptr(cp), _ptr(hp), _ptr(w), _ptr(b), ctx.fb, n, hh, ww, _ptr(gh),
                                        _ptr(gc), _ptr(ws), ws.numel(), _ptr(gx), _ptr(ghp), _ptr(gcp), _ptr(dw),
                                        _ptr(db), _stream()), "convlstm_bwd")
        return gx, gcp, ghp, dw, db, None


# ---- ConvLSTM cells of any size (reference RDN.py:14-24): gates conv on the general kernels + the elementwise gate kernels
GENERAL_NTERMS = 3          # the general cell always computes fp32-class (it is not on bin_stage4's path; its cost is irrelevant)


class _ConvFn(torch.autograd.Function):
    """One stride-1 'same' convolution (k = 1, 3, 5) as a differentiable op on the per-op C ABI: forward binhip_conv2d_fwd,
    backward binhip_conv2d_bwd_weight / _bwd_data on gradient planes carrying a power-of-two scale."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import ops
        nt = GENERAL_NTERMS
        cw = ops.ConvWeights(weight, bias, nterms=nt)
        xp = ops.nchw_to_planes(x, nt)
        y = ops.planes_to_nchw(ops.conv2d(xp, cw), weight.shape[0])
        ctx.xp, ctx.weight, ctx.has_bias = xp, weight, bias is not None
        ctx.weight_version = weight._version        # (the backward-data pass re-lays-out `weight` as it is THEN)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        nt = GENERAL_NTERMS
        if ctx.weight._version != ctx.weight_version:
            raise RuntimeError("bin_amd: a ConvLSTM gate weight was modified in place between forward and backward "
                               "(optimizer step / broadcast / load_state_dict): its gradient would not belong to the forward")
        cout, cin, ks, _ = ctx.weight.shape
        gp, sc = ops.grad_planes(gy, nt)
        dw = db = gx = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw, db = ops.conv2d_bwd_weight(ctx.xp, gp, cout, cin, ks, nt, inv_scale=sc[1:2])
        if ctx.needs_input_grad[0]:
            gxs = ops.planes_to_nchw(ops.conv2d_bwd_data(gp, ops.DgradWeights(ctx.weight, nterms=nt)), cin)
            gx = gxs * sc[1]                              # un-scale (the dgrad is linear in the stored gradient)
        return gx, dw, (db if ctx.has_bias else None)


class _LstmGatesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gates, c_prev, forget_bias, hidden):
        from . import ops
        c1, h1 = ops.lstm_gates(gates, c_prev, forget_bias, hidden)
        ctx.save_for_backward(gates, c_prev)
        ctx.fb, ctx.hidden = float(forget_bias), hidden
        return h1, c1

    @staticmethod
    def backward(ctx, gh, gc):
        from . import ops
        gates, cp = ctx.saved_tensors
        if gh is None and gc is None:
            return None, None, None, None
        dg, gcp = ops.lstm_gates_grad(gates, cp, gh, gc, ctx.fb, ctx.hidden, cp is not None and ctx.needs_input_grad[1])
        return dg, gcp, None, None


def convlstm_general(x, state, weight, bias, forget_bias, hidden):
    """ConvLSTMCell.forward for any (input_size, hidden_size, kernel_size in 1/3/5) — differentiable when grad is enabled."""
    c_prev, h_prev = (None, None) if state is None else state
    if h_prev is None:
        h_prev = torch.zeros((x.shape[0], hidden, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
    stacked = torch.cat((x.float(), h_prev.float()), 1)
    gates = _ConvFn.apply(stacked, weight, bias)
    h1, c1 = _LstmGatesFn.apply(gates, c_prev, forget_bias, hidden)
    return h1, [c1, h1]


def convlstm_apply(x, state, weight, bias, forget_bias):
    c_prev, h_prev = (None, None) if state is None else state
    h, c = _ConvLstmFn.apply(x, c_prev, h_prev, weight, bias, forget_bias)
    return h, [c, h]
