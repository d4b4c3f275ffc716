"""bin_stage4 network — MI355X-native counterpart of the reference's models/archs/RDN.py.

Same class names, constructor arguments, forward signatures, output order and state_dict keys
(1332 aliased keys, OIHW fp32) as the reference, so `adobe_bin.pth` loads with strict=True and
`define_G(opt)` users see no difference.  What differs is everything below the nn.Module boundary:
each RDN sub-network runs as ONE C call into libbinhip.so (66 fused MFMA convolution launches,
concat-free dense blocks in fp16 chunk planes), the ConvLSTM cell is one fused HIP kernel, and the
nn.Conv2d children are parameter containers only — their ATen forward is never called.

There is no CPU path: calling forward on CPU tensors raises (the CPU restatement is oracle/, test-only).

Precision (`precision` attribute of the top module, or env BIN_AMD_PRECISION):
  "f16x3" (default) fp16 hi/lo split, 3 MFMA products, fp32-class results (~1e-6 vs reference)
  "f16"             fp16-input MFMA, fp32 accumulate; whole-net max-abs error ~3e-4 (bar: 1e-3)
"""
import ctypes as C
import os

import torch
import torch.nn as nn
import torch.nn.init as weight_init

from ... import ops
from ...rdn_plan import RdnWeights, rdn_forward

PRECISIONS = {"f16": 1, "f16x3": 3}


def default_precision():
    p = os.environ.get("BIN_AMD_PRECISION", "f16x3")
    if p not in PRECISIONS:
        raise ValueError(f"BIN_AMD_PRECISION must be one of {sorted(PRECISIONS)}, got {p!r}")
    return p


class _nullctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _DirectParamGrads:
    """Context manager behind `net.direct_param_grads()`: flips the per-module flag of the given RDN modules."""

    def __init__(self, mods, on):
        self.mods, self.on = mods, bool(on)

    def __enter__(self):
        self.prev = [m._direct_grads for m in self.mods]
        for m in self.mods:
            m._direct_grads = self.on
        return self

    def __exit__(self, *a):
        for m, v in zip(self.mods, self.prev):
            m._direct_grads = v
        return False


class ConvLSTMCell(nn.Module):
    """reference RDN.py:9-95.  (input_size, hidden_size) = (3, 3) on the live path."""

    def __init__(self, input_size, hidden_size, forget_bias=1.0, kernel_size=3, padding=3 // 2):
        super().__init__()
        if kernel_size not in (1, 3, 5) or padding != kernel_size // 2:
            raise NotImplementedError("bin_amd ConvLSTMCell: kernel_size 1 / 3 / 5 with 'same' padding (the reference's cells are 3x3, pad 1)")
        # (3, 3, 3x3) = the cell of bin_stage4's live path: ONE fused kernel; every other size: gates conv on the general
        # convolution kernels + elementwise gate kernels (bin_amd/autograd.py::convlstm_general)
        self._fused = (input_size, hidden_size, kernel_size) == (3, 3, 3)
        self.input_size, self.hidden_size = input_size, hidden_size
        self.Gates = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, kernel_size, padding=padding, bias=True)
        self._forget_bias = forget_bias
        self._initialize_weights()

    def _initialize_weights(self):
        """RDN.py:26-38: Xavier-uniform gate weights, zero bias (the cell's only conv)."""
        weight_init.xavier_uniform_(self.Gates.weight.data)
        self.Gates.bias.data.zero_()

    def forward(self, input_, prev_state):
        with torch.cuda.device(input_.device) if input_.is_cuda else _nullctx():
            return self._forward(input_, prev_state)

    def _forward(self, input_, prev_state):
        if not self._fused:
            if not input_.is_cuda:
                raise RuntimeError("bin_amd: ConvLSTMCell runs on a HIP device only (no CPU fallback; see oracle/)")
            from ...autograd import convlstm_general
            if torch.is_grad_enabled():
                return convlstm_general(input_, prev_state, self.Gates.weight, self.Gates.bias, self._forget_bias, self.hidden_size)
            with torch.no_grad():
                return convlstm_general(input_, prev_state, self.Gates.weight, self.Gates.bias, self._forget_bias, self.hidden_size)
        if torch.is_grad_enabled() and (input_.requires_grad or self.Gates.weight.requires_grad):
            from ...autograd import convlstm_apply
            return convlstm_apply(input_, prev_state, self.Gates.weight, self.Gates.bias, self._forget_bias)
        return ops.convlstm_cell(input_, prev_state, self.Gates.weight, self.Gates.bias, self._forget_bias)


def pixel_reshuffle(input, upscale_factor):
    """reference RDN.py:107-132 (space-to-depth, exact permutation), fp32 NCHW in and out.  Inside the network this
    op is fused into the input packer (binhip_pack_inputs); the standalone function exists for API parity."""
    return ops.pixel_unshuffle(input, upscale_factor)


class RDB_Conv(nn.Module):
    """reference RDN.py:135-147 (parameter container; executed inside the fused RDN plan)."""

    def __init__(self, inChannels, growRate, kSize=3):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(inChannels, growRate, kSize, padding=(kSize - 1) // 2, stride=1),
                                  nn.ReLU())


class RDB(nn.Module):
    """reference RDN.py:149-165 (parameter container)."""

    def __init__(self, growRate0, growRate, nConvLayers, kSize=3):
        super().__init__()
        self.convs = nn.Sequential(*[RDB_Conv(growRate0 + c * growRate, growRate) for c in range(nConvLayers)])
        self.LFF = nn.Conv2d(growRate0 + nConvLayers * growRate, growRate0, 1, padding=0, stride=1)


class _RDNBase(nn.Module):
    """Shared body of RDN_residual_interp_{2,2_1,4_1}_input (reference RDN.py:167-334): they differ only
    in the number of input frames."""
    N_INPUTS = 0

    def __init__(self, G0=64, D=6, C=4, G=32):
        super().__init__()
        from ...rdn_plan import check_shape
        self.shape = check_shape((G0, D, C, G))     # raises NotImplementedError outside what the HIP plan runs
        self.G0, self.D, self.C, self.G = G0, D, C, G
        kSize = 3
        self.SFENet1 = nn.Conv2d(12 * self.N_INPUTS, G0, 5, padding=2, stride=1)
        self.SFENet2 = nn.Conv2d(G0, G0, kSize, padding=1, stride=1)
        self.RDBs = nn.ModuleList([RDB(growRate0=G0, growRate=G, nConvLayers=C) for _ in range(D)])
        self.GFF = nn.Sequential(nn.Conv2d(D * G0, G0, 1, padding=0, stride=1),
                                 nn.Conv2d(G0, G0, kSize, padding=1, stride=1))
        self.UPNet = nn.Sequential(nn.Conv2d(G0, 256, kSize, padding=1, stride=1), nn.PixelShuffle(2),
                                   nn.Conv2d(64, 3, kSize, padding=1, stride=1))
        self.precision = None          # None -> inherit default_precision()
        # per-object switches of the HIP path (see bin_amd/autograd.py; deliberately NOT module globals)
        from ...autograd import default_backward_precision, default_wgrad_side_stream
        from ...rdn_plan import default_plan_flags
        self.backward_precision = default_backward_precision()   # "f16" = single-product backward behind an f16x3 forward
        self._direct_grads = False                               # kernels accumulate weight gradients straight into .grad
        self.allow_f16_training = False                          # "f16" is the inference mode; training in it is gated (autograd.py)
        self.wgrad_side_stream = default_wgrad_side_stream()     # weight-gradient kernels beside the backward-data chain
        self.plan_flags = default_plan_flags()                   # BINHIP_PLAN_* bits of every call of this sub-network
        self.profiler = None                                     # BinhipProfiler handle (bench.py's roofline leg)
        self.bwd_profiler = None                                 # same, for the weight-gradient launches of the backward
        self.debug_hook = None                                   # callable(kind, module, dims, workspace, info) after a training
                                                                 # forward / backward (tools/fp16_headroom.py)
        self._wcache = None            # (key, RdnWeights)
        self._wgen = 0                 # bumped by invalidate_kernel_weights()

    def direct_param_grads(self, on=True):
        """Context (same name as on the whole network, so a bare RDN sub-network can be a wrapper's netG): while active the
        backward kernels of this RDN write / accumulate their weight gradients straight into the parameters' .grad."""
        return _DirectParamGrads([self], on)

    def _weights_key(self, nterms):
        params = list(self.parameters())
        return (nterms, str(params[0].device), self._wgen, tuple((p.data_ptr(), p._version) for p in params))

    def invalidate_kernel_weights(self):
        """Force a relayout on the next forward.  The cache key holds every parameter's (data_ptr, _version); a write
        through `.data` (p.data.copy_(), EMA / clipping code, dist.broadcast(p.data)) does NOT bump _version, so code
        that mutates parameters that way must call this (load_network and the DP parameter broadcast do)."""
        self._wgen += 1

    def kernel_weights(self, nterms):
        """Relayout the 66 convolutions into kernel layout, cached on the parameters' version counters."""
        key = self._weights_key(nterms)
        if self._wcache is None or self._wcache[0] != key:
            with torch.no_grad():
                self._wcache = (key, RdnWeights(dict(self.named_parameters()), self.N_INPUTS, nterms, shape=self.shape))
        return self._wcache[1]

    def _run(self, *frames):
        if len(frames) != self.N_INPUTS:
            raise TypeError(f"{type(self).__name__}.forward takes {self.N_INPUTS} frames")
        if not all(f.is_cuda for f in frames):
            raise RuntimeError("bin_amd: RDN sub-networks run on a HIP device only (no CPU fallback; see oracle/)")
        with torch.cuda.device(frames[0].device):      # kernels launch on the CURRENT device: make it the tensors' one
            if torch.is_grad_enabled() and (any(f.requires_grad for f in frames) or
                                             any(p.requires_grad for p in self.parameters())):
                from ...autograd import rdn_apply        # training path (HIP backward)
                return rdn_apply(self, frames)
            nterms = PRECISIONS[self.precision or default_precision()]
            return rdn_forward(self.kernel_weights(nterms), list(frames), flags=self.plan_flags, profiler=self.profiler)


class RDN_residual_interp_2_input(_RDNBase):
    N_INPUTS = 2

    def forward(self, B0, B1):
        return self._run(B0, B1)


class RDN_residual_interp_2_1_input(_RDNBase):
    N_INPUTS = 3

    def forward(self, I0, I1, I2):
        return self._run(I0, I1, I2)


class RDN_residual_interp_4_1_input(_RDNBase):
    N_INPUTS = 5

    def forward(self, B0, B1, B2, B3, B4):
        return self._run(B0, B1, B2, B3, B4)


class RDN_residual_interp_5_input(nn.Module):
    """4-stage pyramid (reference RDN.py:337-405); lstm=True is the only constructible branch there."""

    def __init__(self, lstm=False, GO=64, D=6):
        super().__init__()
        if not lstm:
            raise NotImplementedError("reference RDN.py:358 references an undefined class when lstm=False")
        self.lstm = lstm
        self.model1_1 = RDN_residual_interp_2_input(G0=GO, D=D)
        self.model1_2 = self.model1_1
        self.model1_3 = self.model1_1
        self.model1_4 = self.model1_1
        self.model2_1 = RDN_residual_interp_2_1_input(G0=GO, D=D)
        self.model2_2 = self.model2_1
        self.model2_3 = self.model2_1
        self.model3_1 = RDN_residual_interp_4_1_input(G0=GO, D=D)
        self.model3_2 = self.model3_1
        self.model4_1 = RDN_residual_interp_4_1_input(G0=GO, D=D)

    def forward(self, B1, B3, B5, B7, B9, previous_input=None, stage1=None):
        """`stage1`: optional precomputed (I2', I4', I6', I8') entries (None = compute) — lets the
        2-window wrapper reuse the three stage-1 results that window 2 shares with window 1."""
        s1 = stage1 or (None, None, None, None)
        I2 = s1[0] if s1[0] is not None else self.model1_1(B1, B3)
        I4 = s1[1] if s1[1] is not None else self.model1_2(B3, B5)
        I6 = s1[2] if s1[2] is not None else self.model1_3(B5, B7)
        I8 = s1[3] if s1[3] is not None else self.model1_4(B7, B9)
        if previous_input is not None and previous_input[0] is not None:
            p4, p6, p8, p5, p7, p6b = previous_input
            I3 = self.model2_1(p4, I2, I4)
            I5 = self.model2_2(p6, I4, I6)
            I7 = self.model2_3(p8, I6, I8)
            I4pp = self.model3_1(p5, B3, I3, I5, B5)
            I6pp = self.model3_2(p7, B5, I5, I7, B7)
            I5ppp = self.model4_1(p6b, I4, I4pp, I6pp, I6)
        else:
            I3 = self.model2_1(I2, I2, I4)
            I5 = self.model2_2(I4, I4, I6)
            I7 = self.model2_3(I6, I6, I8)
            I4pp = self.model3_1(I3, B3, I3, I5, B5)
            I6pp = self.model3_2(I5, B5, I5, I7, B7)
            I5ppp = self.model4_1(I4, I4, I4pp, I6pp, I6)
        self.I2_prime, self.I4_prime, self.I6_prime, self.I8_prime = I2, I4, I6, I8
        self.I3_prime, self.I5_prime, self.I7_prime = I3, I5, I7
        self.I4_prime_prime, self.I6_prime_prime, self.I5_prime_prime = I4pp, I6pp, I5ppp
        return I2, I4, I6, I8, I3, I5, I7, I4pp, I6pp, I5ppp


class RDN_residual_interp_5_input_ConvLSTM_L(nn.Module):
    """Two overlapping 5-frame windows bridged by six ConvLSTM cells (reference RDN.py:408-465).

    `reuse_schedule=True` (default) skips work the reference recomputes or discards, without changing
    any returned value (verified bit-identical): window 2's stage-1 calls on (B3,B5),
    (B5,B7), (B7,B9) equal window 1's I4', I6', I8' (same kernels, same inputs => same bits), and the six
    ConvLSTM calls after window 2 feed nothing.  20 -> 17 RDN calls, 12 -> 6 cells.
    `reuse_schedule=False` runs the reference's literal schedule (all 20 + 12)."""

    def __init__(self, modelType="lstm"):
        super().__init__()
        self.modelType = modelType
        self.clstm_4_prime = ConvLSTMCell(3, 3)
        self.clstm_6_prime = ConvLSTMCell(3, 3)
        self.clstm_8_prime = ConvLSTMCell(3, 3)
        self.clstm_5_prime_prime = ConvLSTMCell(3, 3)
        self.clstm_7_prime_prime = ConvLSTMCell(3, 3)
        self.clstm_6_prime_prime_prime = ConvLSTMCell(3, 3)
        self.model = RDN_residual_interp_5_input(lstm=True, GO=96, D=12)
        self.prev_state = None         # inert in the reference too (RDN.py:419-420)
        self.hidden_state = None
        self.reuse_schedule = True
        self.precision = None
        # concurrent RDN calls in inference; None = by precision (f16: 3 streams fill each other's launch tails;
        # f16x3: 1 — the chip is power-capped there and one call's 231 MB block buffer stays Infinity-Cache
        # resident only when nothing else streams beside it: 74.1 vs 76.4 ms per 720p window, same box)
        self.n_streams = int(os.environ["BIN_AMD_STREAMS"]) if os.environ.get("BIN_AMD_STREAMS") else None
        # training (grad enabled): the whole pyramid as FOUR RDN calls, one per weight set (see _forward_four_calls)
        self.four_calls = os.environ.get("BIN_AMD_FOUR_CALLS", "1") != "0"
        # inference: use the four-call schedule too when a launch of the per-call schedule would not fill the chip
        # (small frames); "1"/"0" force it on / off, default "auto" (see _use_four_calls_infer)
        self.four_calls_infer = os.environ.get("BIN_AMD_INFER_FOUR", "auto")
        self._streams = None

    def set_precision(self, precision):
        if precision is not None and precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        self.precision = precision
        for m in self.modules():
            if isinstance(m, _RDNBase):
                m.precision = precision
        return self

    def rdn_modules(self):
        """The four RDN weight sets (model1_1 .. model4_1), each once."""
        seen, out = set(), []
        for m in self.modules():
            if isinstance(m, _RDNBase) and id(m) not in seen:
                seen.add(id(m))
                out.append(m)
        return out

    def set_backward_precision(self, precision):
        """"f16": single-product backward behind an f16x3 forward (bin_amd/autograd.py); None = same as the forward."""
        if precision not in (None, "f16", "f16x3"):
            raise ValueError("backward_precision must be None, 'f16' or 'f16x3'")
        for m in self.rdn_modules():
            m.backward_precision = None if precision == "f16x3" else precision
        return self

    def set_profiler(self, handle, backward=False):
        """Attach (or, with None, detach) a BinhipProfiler handle to every RDN call of THIS network (`backward`: to the
        weight-gradient launches of its backward calls instead)."""
        for m in self.rdn_modules():
            if backward:
                m.bwd_profiler = handle
            else:
                m.profiler = handle
        return self

    def direct_param_grads(self, on=True):
        """Context: while active, the backward kernels of THIS network's RDN calls write / accumulate their weight
        gradients straight into the parameters' .grad buffers (bin_amd/autograd.py)."""
        return _DirectParamGrads(self.rdn_modules(), on)

    def forward(self, B1, B3, B5, B7, B9, B11, stage1_cache=None, input_events=None):
        """`input_events` (bin_amd extension, inference only): a list of torch.cuda.Event after which the six frames are
        complete (empty list = they already are, e.g. resident frames).  The side streams of the inference schedule
        then wait on those events instead of on the caller's stream, so back-to-back forwards pipeline: the next
        window's stage-1 calls start on idle streams while the previous window's lone stage-4 call is still running.
        The outputs are joined onto the caller's stream as always.
        `stage1_cache` (bin_amd extension, inference only): a dict owned by a streaming caller.  Consecutive
        windows of a clip share 4 of their 5 stage-1 frame pairs, and — round 4 — the first window of
        the next forward also repeats two stage-2 calls and one stage-3 call of this one (it sees no ConvLSTM state), so
        every LSTM-free call is memoised on the identity of its (cached, hence long-lived) input tensors: 17 -> 10 RDN
        calls per window (rounds 1-3: 13), outputs unchanged bit for bit.  (The name is kept from round 1.)  The key also
        carries the weight set's state (precision, parameter versions, relayout generation), so a dict that outlives a
        weight change misses instead of serving stale results.  While a cache is live the OUTPUTS are shared with it (this
        window's I5 comes back as the next window's I3): do not modify them in place."""
        for t in (B1, B3, B5, B7, B9, B11):
            if not t.is_cuda:
                raise RuntimeError("bin_amd: bin_stage4 runs on a HIP device only (no CPU fallback; "
                                   "see oracle/ for the test-only CPU restatement)")
        with torch.cuda.device(B1.device):             # kernels launch on the CURRENT device: make it the frames' one
            return self._forward(B1, B3, B5, B7, B9, B11, stage1_cache, input_events)

    def resolved_streams(self):
        if self.n_streams is not None:
            return max(1, int(self.n_streams))
        return 1 if (self.precision or default_precision()) == "f16x3" else 3

    def _use_four_calls_infer(self, frame):
        if self.four_calls_infer in ("0", "1"):
            return self.four_calls_infer == "1"
        n, _, h, w = frame.shape
        tiles = n * ((h // 2 + 15) // 16) * ((w // 2 + 31) // 32)      # 16x32 tiles of one dense-block conv launch
        return tiles < 256                                             # fewer than one workgroup per CU

    def _forward(self, B1, B3, B5, B7, B9, B11, stage1_cache=None, input_events=None):
        if (self.reuse_schedule and self.modelType == "lstm" and not torch.is_grad_enabled() and stage1_cache is None
                and input_events is None and getattr(self, "_graph_mode", None) is None
                and self._use_four_calls_infer(B1)):
            return self._forward_four_calls((B1, B3, B5, B7, B9, B11))
        if self.reuse_schedule and self.modelType == "lstm" and not (
                torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or
                                             any(t.requires_grad for t in (B1, B3, B5, B7, B9, B11)))):
            return self._forward_streams((B1, B3, B5, B7, B9, B11), stage1_cache, input_events)
        if stage1_cache is not None or input_events is not None:
            raise RuntimeError("bin_amd: stage1_cache / input_events need the inference schedule (reuse_schedule, no grad)")
        if self.four_calls and self.reuse_schedule and self.modelType == "lstm":
            return self._forward_four_calls((B1, B3, B5, B7, B9, B11))
        cells = (self.clstm_4_prime, self.clstm_6_prime, self.clstm_8_prime, self.clstm_5_prime_prime,
                 self.clstm_7_prime_prime, self.clstm_6_prime_prime_prime)
        picks = (1, 2, 3, 5, 6, 8)
        states = [None] * 6
        hidden = [None] * 6
        res = []
        windows = ((B1, B3, B5, B7, B9), (B3, B5, B7, B9, B11))
        for wi, win in enumerate(windows):
            stage1 = None
            if wi == 1 and self.reuse_schedule:
                stage1 = (res[0][1], res[0][2], res[0][3], None)
            out = self.model(*win, hidden, stage1=stage1)
            self.Ft_p_1 = out
            last = wi == len(windows) - 1
            if self.modelType == "lstm" and not (last and self.reuse_schedule):
                hidden = []
                for k, (idx, cell) in enumerate(zip(picks, cells)):
                    h, states[k] = cell(out[idx], states[k])
                    hidden.append(h)
            elif self.modelType != "lstm":
                hidden = [out[i] for i in picks]
            res.append(out)
        return (res[0][0], res[0][1], res[0][2], res[0][3], res[0][4], res[0][5], res[0][6],
                res[0][7], res[0][8], res[0][9], res[1][3], res[1][6], res[1][8], res[1][9])


def _forward_streams(self, B, stage1_cache=None, input_events=None):
    """Inference schedule that runs the INDEPENDENT RDN calls of each pyramid stage on separate HIP streams
    (stage 1: 4 calls, stage 2: 3, stage 3: 2; window 2's only new stage-1 call rides along with window 1's stage 4).
    Every kernel of one call still runs in order on its stream; calls on different streams overlap, so one call's
    kernel tail / launch gap / prologue is filled by another call's workgroups.  Same kernels, same arithmetic
    => bit-identical to the serial schedule (tests/test_gpu_net.py)."""
    from ...rdn_plan import rdn_forward, workspace
    from ... import _lib as L
    dev = B[0].device
    main = torch.cuda.current_stream(dev)
    ns = self.resolved_streams()
    if ns == 1:
        streams = [main]                  # serial: everything on the caller's stream, no cross-stream events at all
    else:
        if self._streams is None or len(self._streams) != ns or self._streams[0].device != dev:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        streams = self._streams
    m = self.model
    mods = {1: m.model1_1, 2: m.model2_1, 3: m.model3_1, 4: m.model4_1}
    nterms = {k: PRECISIONS[v.precision or default_precision()] for k, v in mods.items()}
    wkeys = {k: v._weights_key(nterms[k]) for k, v in mods.items()}     # (precision, device, generation, every parameter's version)
    stale = [v._wcache is None or v._wcache[0] != wkeys[k] for k, v in mods.items()]
    relayout_pending = any(stale)
    kw = {k: v.kernel_weights(nterms[k]) for k, v in mods.items()}       # relayouts (if any) on the main stream
    for k, v in mods.items():
        # ... and the fused UPNet's operands (BINHIP_PLAN_FUSED_UPNET): built HERE, on the main stream, which the side streams wait for
        # below — rdn_forward would otherwise build them lazily on whichever side stream calls first, unordered against the others
        if (v.plan_flags & L.PLAN_FUSED_UPNET) and kw[k].fused_up is None and kw[k]._up_src is not None:
            kw[k].ensure_fused_upnet()
            relayout_pending = True
    lib = L.lib()
    n, _, h, w = B[0].shape
    ready = {}                                                          # id(tensor) -> (event recorded after its call, stream)
    mode = getattr(self, "_graph_mode", None)                           # None | "capture" | "replay" (harness.GraphedNet)
    if mode == "capture":
        self._call_graphs = []
    pools = {}
    counter = [0]

    def launch(si, fn, ins):
        if ns == 1 and mode is None:      # serial on the caller's stream: program order is the dependency order
            return fn()
        # a consumer waits on the event recorded right after the producing CALL (exact dependency, eager events)
        s = streams[si % len(streams)]
        for t in ins:
            hit = ready.get(id(t))
            if hit is not None and hit[1] is not s:
                s.wait_event(hit[0])
        if mode == "capture":
            # one hipGraph per call (the 67 launches of an RDN, or a ConvLSTM cell), captured on the call's own stream.
            # Capturing the WHOLE multi-stream forward into one graph crashes hipStreamEndCapture on ROCm 7.2 as soon
            # as two captured streams depend on each other in both directions (tools/probe_graph.py), so the
            # cross-stream edges stay eager events and only the launch-heavy bodies are graphs.
            # Graphs captured on ONE stream replay in order, so they share a memory pool (a call's workspace, freed when
            # its capture ends, is reused by the next call's); graphs of different streams replay concurrently and
            # must not.
            g = torch.cuda.CUDAGraph()
            k = si % len(streams)
            if k not in pools:                      # (setdefault would build a fresh pool handle on every call)
                pools[k] = torch.cuda.graph_pool_handle()
            pool = pools[k]
            with torch.cuda.graph(g, stream=s, pool=pool):
                out = fn()
            self._call_graphs.append((g, out))
        elif mode == "replay":
            g, out = self._call_graphs[counter[0]]
            with torch.cuda.stream(s):
                g.replay()
        else:
            with torch.cuda.stream(s):
                out = fn()
        counter[0] += 1
        e = torch.cuda.Event()
        e.record(s)
        for o in (out if isinstance(out, (tuple, list)) else (out,)):
            ready[id(o)] = (e, s)
        return out

    def rdn(si, k, *ins):
        from ...rdn_plan import c_shape
        nb = lib.binhip_rdn_workspace_bytes(n, h, w, len(ins), nterms[k], C.byref(c_shape(mods[k].shape)))

        def fn():
            # layout / dtype conversions (no-ops for the usual contiguous fp32 frames) run on the call's own stream,
            # after its waits on the producers — not on the caller's stream, where nothing would order them
            ws = workspace(nb, dev, key=f"fwd-stream{si % len(streams)}")
            return rdn_forward(kw[k], [t.contiguous().float() for t in ins], ws=ws, flags=mods[k].plan_flags,
                               profiler=mods[k].profiler)
        return launch(si, fn, ins)

    for s in streams:
        # every stream that runs a first consumer waits for the frames' own events — including the caller's stream in
        # the serial schedule (the frames may have been produced on a copy / decode stream the caller never joined)
        for ev in (input_events or ()):
            s.wait_event(ev)
        if s is main:
            continue
        # without events the side streams wait for what the caller's stream has queued so far (the frames); they also
        # do so for the weight relayouts kernel_weights() may just have issued there (first call / after an optimizer step)
        if input_events is None or relayout_pending:
            s.wait_stream(main)
    cells = (self.clstm_4_prime, self.clstm_6_prime, self.clstm_8_prime, self.clstm_5_prime_prime,
             self.clstm_7_prime_prime, self.clstm_6_prime_prime_prime)

    def cell(si, c, x):
        return launch(si, lambda: c(x, None)[0], [x])

    B1, B3, B5, B7, B9, B11 = B

    touched = set()

    def memo(si, k, *ins):
        """An LSTM-free RDN call, memoised across forwards on the IDENTITY of its input tensors when the caller streams
        consecutive windows (`stage1_cache`).  Window k + 1's first window is window k's shifted by one frame and sees no
        ConvLSTM state, so besides four of the five stage-1 pairs (round 1) it repeats two stage-2 calls and one stage-3 call
        of window k exactly — and because a hit returns the very tensor object the earlier forward produced, the deeper
        calls' keys match on their own (I3' = model2(I2', I2', I4') = model2(I4, I4, I6) = I5 of the window before).  17
        independent -> 10 RDN calls per window, same kernels on the same inputs => the same bits."""
        if stage1_cache is None:
            return rdn(si, k, *ins)
        # the weight set's state is part of the key (advisor r04): a cache dict that survives an optimizer step,
        # load_state_dict, set_precision or invalidate_kernel_weights() must miss, not serve the old weights' outputs
        key = (k, wkeys[k]) + tuple(id(t) for t in ins)      # the weight state itself, not its hash (advisor r05)
        touched.add(key)
        hit = stage1_cache.get(key)
        if hit is not None and all(a is b for a, b in zip(hit[1], ins)):
            if hit[2] is not None:
                ready[id(hit[0])] = hit[2]        # computed by an earlier forward: consumers still wait on ITS event
            return hit[0]
        out = rdn(si, k, *ins)
        stage1_cache[key] = (out, ins, ready.get(id(out)))     # (holding `ins` keeps their ids from being recycled)
        return out

    # ---- window 1 (no ConvLSTM input: every call is a pure function of frames / earlier window-1 results)
    I2 = memo(0, 1, B1, B3); I4 = memo(1, 1, B3, B5); I6 = memo(2, 1, B5, B7); I8 = memo(3, 1, B7, B9)
    I3 = memo(0, 2, I2, I2, I4); I5 = memo(1, 2, I4, I4, I6); I7 = memo(2, 2, I6, I6, I8)
    h4 = cell(3, cells[0], I4); h6 = cell(3, cells[1], I6); h8 = cell(3, cells[2], I8)
    I4pp = memo(0, 3, I3, B3, I3, I5, B5); I6pp = memo(1, 3, I5, B5, I5, I7, B7)
    h5 = cell(2, cells[3], I5); h7 = cell(2, cells[4], I7)
    I8b = memo(2, 1, B9, B11)                                           # window 2's only new stage-1 call
    I5ppp = memo(0, 4, I4, I4, I4pp, I6pp, I6)
    h6pp = cell(1, cells[5], I6pp)
    # ---- window 2 (stage-1 outputs I4, I6, I8 of window 1 are its I2', I4', I6'; its deeper calls see the cells' states)
    J3 = rdn(0, 2, h4, I4, I6); J5 = rdn(1, 2, h6, I6, I8); J7 = rdn(2, 2, h8, I8, I8b)
    J4pp = rdn(0, 3, h5, B5, J3, J5, B7); J6pp = rdn(1, 3, h7, B7, J5, J7, B9)
    J5ppp = rdn(0, 4, h6pp, I6, J4pp, J6pp, I8)
    if stage1_cache is not None:         # keep only what this forward touched: whatever can recur in a neighbouring window
        for key in [key for key in stage1_cache if key not in touched]:
            del stage1_cache[key]
    for s in streams:
        if s is not main:
            main.wait_stream(s)
    outs = (I2, I4, I6, I8, I3, I5, I7, I4pp, I6pp, I5ppp, I8b, J7, J6pp, J5ppp)
    if ns > 1 and mode is None and not torch.cuda.is_current_stream_capturing():
        # caching-allocator bookkeeping for tensors that crossed streams (graph outputs live in the graphs' private pools)
        for t in outs + (h4, h6, h8, h5, h7, h6pp, J3, J5, J4pp):
            for s in streams:
                t.record_stream(s)
            t.record_stream(main)
    self.Ft_p_1 = (I6, I8, I8b, None, J3, J5, J7, J4pp, J6pp, J5ppp)
    return outs


RDN_residual_interp_5_input_ConvLSTM_L._forward_streams = _forward_streams      # (defined below the class for readability)


def _forward_four_calls(self, B):
    """The differentiable schedule: both windows' pyramid as FOUR RDN calls, one per weight set (reference RDN.py:342-363
    shares model1 between 4, model2 between 3, model3 between 2 calls of a window; :435-459 runs two windows).
    Window 2 depends on window 1 only through the ConvLSTM hand-offs, and each of those needs nothing deeper than the
    stage it feeds: h4/h6/h8 <- stage-1 outputs, h5/h7 <- stage 2, h6'' <- stage 3.  So stage s of BOTH windows can run
    as one batch along N once stage s-1 of both is done:
        model1  N = 5n : (B1,B3) (B3,B5) (B5,B7) (B7,B9) (B9,B11)             -> I2 I4 I6 I8 I8b   (3 pairs are shared)
        model2  N = 6n : window 1 (I2,I2,I4) (I4,I4,I6) (I6,I6,I8) ; window 2 (h4,I4,I6) (h6,I6,I8) (h8,I8,I8b)
        model3  N = 4n : window 1 (I3,B3,I3,I5,B5) (I5,B5,I5,I7,B7) ; window 2 (h5,B5,J3,J5,B7) (h7,B7,J5,J7,B9)
        model4  N = 2n : window 1 (I4,I4,I4'',I6'',I6) ; window 2 (h6'',I6,J4'',J6'',I8)
    Per image the kernels compute exactly what the 17 separate calls compute (forward values are bit-identical); the
    weight gradients are the same sums taken in one reduction per layer instead of up to five.  Launches per training
    step drop from ~4 400 to ~1 100 and every launch carries 2-6x the tiles, which is what the small training crops
    (8 x 128x128 half-resolution pixels = one workgroup per CU per launch) were missing."""
    m = self.model
    n = B[0].shape[0]
    B1, B3, B5, B7, B9, B11 = B
    cat = torch.cat
    cells = (self.clstm_4_prime, self.clstm_6_prime, self.clstm_8_prime, self.clstm_5_prime_prime,
             self.clstm_7_prime_prime, self.clstm_6_prime_prime_prime)

    def parts(t, k):
        # ONE split node, not k slices: the backward of a slice allocates and zero-fills a full-size tensor per slice and adds
        # them up (k x (fill + add) over the whole batched output: ~6 GB of pure autograd traffic per 8 x 256x256 step); the
        # backward of split is a single concatenation of the k slice gradients (same box: 129.5 -> 129.0 ms per step)
        out = torch.split(t, n, 0)
        assert len(out) == k
        return list(out)

    I2, I4, I6, I8, I8b = parts(m.model1_1(cat((B1, B3, B5, B7, B9), 0), cat((B3, B5, B7, B9, B11), 0)), 5)
    h4, h6, h8 = (cells[k](x, None)[0] for k, x in enumerate((I4, I6, I8)))
    I3, I5, I7, J3, J5, J7 = parts(m.model2_1(cat((I2, I4, I6, h4, h6, h8), 0), cat((I2, I4, I6, I4, I6, I8), 0),
                                              cat((I4, I6, I8, I6, I8, I8b), 0)), 6)
    h5, h7 = cells[3](I5, None)[0], cells[4](I7, None)[0]
    I4pp, I6pp, J4pp, J6pp = parts(m.model3_1(cat((I3, I5, h5, h7), 0), cat((B3, B5, B5, B7), 0), cat((I3, I5, J3, J5), 0),
                                              cat((I5, I7, J5, J7), 0), cat((B5, B7, B7, B9), 0)), 4)
    h6pp = cells[5](I6pp, None)[0]
    I5ppp, J5ppp = parts(m.model4_1(cat((I4, h6pp), 0), cat((I4, I6), 0), cat((I4pp, J4pp), 0), cat((I6pp, J6pp), 0),
                                    cat((I6, I8), 0)), 2)
    self.Ft_p_1 = (I6, I8, I8b, None, J3, J5, J7, J4pp, J6pp, J5ppp)
    return (I2, I4, I6, I8, I3, I5, I7, I4pp, I6pp, I5ppp, I8b, J7, J6pp, J5ppp)


RDN_residual_interp_5_input_ConvLSTM_L._forward_four_calls = _forward_four_calls


def bin_stage4_lstm():
    """reference RDN.py:469-471."""
    return RDN_residual_interp_5_input_ConvLSTM_L()
