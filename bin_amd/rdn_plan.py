"""Host side of one RDN sub-network (reference RDN.py:167-334): relayout its 66 convolutions once,
then run the whole sub-network with a single C call (binhip_rdn_forward)."""
import ctypes as C

import torch

from . import _lib as L
from .ops import ConvWeights, _ptr, _stream, _need_cuda, on_device, relayout_batch, relayout_item, status_word

STAGE4_SHAPE = (96, 12, 4, 32)          # (G0, D, C, G) of bin_stage4's sub-networks (reference RDN.py:418, :171-172)


def check_shape(shape):
    """The (G0, D, C, G) configurations the HIP plan runs (include/binhip.h, BinRdnShape); anything else raises."""
    G0, D, C, G = (int(v) for v in shape)
    if G0 % 32 or not 32 <= G0 <= 256 or G % 32 or not 32 <= G <= 128 or not 1 <= C <= L.RDN_MAX_CONVS or not 1 <= D <= 20 \
            or 2 + D * (C + 1) + 4 > L.RDN_MAX_LAYERS:
        raise NotImplementedError(
            f"bin_amd RDN: unsupported configuration G0={G0}, D={D}, C={C}, G={G} (built: G0 and G multiples of 32 with "
            f"32 <= G0 <= 256, G <= 128; 1 <= C <= {L.RDN_MAX_CONVS}; 1 <= D <= 20)")
    return G0, D, C, G


def c_shape(shape):
    s = L.BinRdnShape()
    s.G0, s.D, s.C, s.G = shape
    return s


def layer_names(shape=STAGE4_SHAPE):
    """Local parameter prefixes of the 2 + D (C + 1) + 4 layers in launch order (matches binhip_plan.hip; 66 for bin_stage4)."""
    _, D, C, _ = shape
    names = ["SFENet1", "SFENet2"]
    for d in range(D):
        names += [f"RDBs.{d}.convs.{c}.conv.0" for c in range(C)]
        names.append(f"RDBs.{d}.LFF")
    names += ["GFF.0", "GFF.1", "UPNet.0", "UPNet.2"]
    return names


class RdnWeights:
    """Kernel-layout weights of one RDN weight set + the BinRdnPlan pointer table."""

    def __init__(self, params, n_inputs, nterms, prefix="", shape=STAGE4_SHAPE):
        self.n_inputs, self.nterms, self.shape = n_inputs, nterms, check_shape(shape)
        self.layers = []
        items = []                         # the relayouts of the set (66 for bin_stage4) go out as a few batched launches
        for i, nm in enumerate(layer_names(self.shape)):
            w, b = params[f"{prefix}{nm}.weight"], params[f"{prefix}{nm}.bias"]
            shuffle = nm == "UPNet.0"
            cin_chunks = None
            if nm == "SFENet1":
                cin_chunks = (12 * n_inputs + 15) // 16
            self.layers.append(ConvWeights(w, b, nterms=nterms, shuffle=shuffle, cin_chunks=cin_chunks, defer=items))
        assert len(self.layers) == 2 + self.shape[1] * (self.shape[2] + 1) + 4
        relayout_batch(items)

        self._dgrad = None

    def fill_plan(self, plan):
        plan.shape = c_shape(self.shape)
        for i, cw in enumerate(self.layers):
            plan.w_hi[i] = cw.w_hi.data_ptr()
            plan.w_lo[i] = cw.w_lo.data_ptr() if cw.w_lo is not None else None
            plan.bias[i] = cw.bias.data_ptr()

    def dgrad(self, module, nterms=None):
        """Backward-data weights (transposed + flipped), built on first use for the same parameter version; `nterms`
        other than the forward's for the single-product backward behind an f16x3 forward."""
        nterms = self.nterms if nterms is None else nterms
        if self._dgrad is None:
            self._dgrad = {}
        if nterms not in self._dgrad:
            with torch.no_grad():
                self._dgrad[nterms] = RdnDgradWeights(dict(module.named_parameters()), self.n_inputs, nterms,
                                                      shape=self.shape)
        return self._dgrad[nterms]


class RdnDgradWeights:
    """Backward-data weights of the 66 layers in kernel layout.

    Plain layers: W'[ci][co][dy][dx] = W[co][ci][k-1-dy][k-1-dx] (binhip_weights_relayout_dgrad).
    The four 3x3 convs of each residual dense block are stored in GATHER form (binhip_weights_relayout_rdb_gather):
    slot `RDBs.d.convs.g` holds the weights that produce concat-group g of the block from the stacked output
    gradients of convs g..3 - what binhip_rdn_backward expects (include/binhip.h, BinRdnBwdPlan)."""

    def __init__(self, params, n_inputs, nterms, prefix="", shape=STAGE4_SHAPE):
        lib = L.lib()
        self.shape = G0, D, C, G = check_shape(shape)
        self.w_hi, self.w_lo = [], []
        dev = None
        names = layer_names(self.shape)
        fp32 = {nm: params[f"{prefix}{nm}.weight"].detach().contiguous().float() for nm in names}

        def alloc(rows, chunks, ks):
            nbytes = lib.binhip_weights_bytes(rows, chunks, ks)
            hi = torch.empty(nbytes // 2, dtype=torch.float16, device=dev)
            lo = torch.empty(nbytes // 2, dtype=torch.float16, device=dev) if nterms == 3 else None
            return hi, lo, torch.empty(rows, dtype=torch.float32, device=dev)

        items, scratch = [], []              # scratch: per-layer zero-bias outputs, only needed until the launch is queued
        self._src = fp32
        for nm in names:
            w = fp32[nm]
            dev = w.device
            cout, cin, ks, _ = w.shape
            if ".convs." in nm:
                d, g = int(nm.split(".")[1]), int(nm.split(".")[3])
                rows, chunks = (G0 if g == 0 else G), (C - g) * G // 16
                hi, lo, zb = alloc(rows, chunks, 3)
                srcs = [fp32[f"RDBs.{d}.convs.{c}.conv.0"] for c in range(C)]
                cb = lib.binhip_conv_cout_block(3, rows, nterms)
                items.append(relayout_item(L.RELAYOUT_RDB_GATHER, srcs, None, hi, lo, zb, G, G0 + G * g, 3, rows, chunks, cb, g,
                                           shape=self.shape))
            else:
                rows_pad = lib.binhip_dgrad_rows_pad(ks, cin)
                cin_chunks = (cout + 15) // 16
                cb = lib.binhip_conv_cout_block(ks, rows_pad, nterms)
                hi, lo, zb = alloc(rows_pad, cin_chunks, ks)
                items.append(relayout_item(L.RELAYOUT_DGRAD, [w], None, hi, lo, zb, cout, cin, ks, rows_pad, cin_chunks, cb,
                                           1 if nm == "UPNet.0" else 0))
            self.w_hi.append(hi)
            self.w_lo.append(lo)
            scratch.append(zb)
        relayout_batch(items)
        del scratch
        self.zero_bias = torch.zeros(max(D * G0, G0 + C * G, 256, 1152), dtype=torch.float32, device=dev)

    def fill_plan(self, plan):
        plan.shape = c_shape(self.shape)
        for i in range(len(self.w_hi)):
            plan.wt_hi[i] = self.w_hi[i].data_ptr()
            plan.wt_lo[i] = self.w_lo[i].data_ptr() if self.w_lo[i] is not None else None
        plan.zero_bias = self.zero_bias.data_ptr()


from collections import OrderedDict as _OrderedDict

_workspaces = _OrderedDict()          # (device type, index, stream handle, purpose) -> uint8 tensor, least recently used first
WORKSPACE_CACHE_ENTRIES = 12          # per process; an evicted entry is simply re-allocated by its next user


def workspace(nbytes, device, key="fwd"):
    """One cached workspace per (device, CURRENT STREAM, purpose), grown on demand.  The stream is part of the key
    because a workspace is only ordered against its own stream's work: two host threads (or two networks) driving
    different streams of one device must never share one.
    Lifetime: the cache is a small LRU (a stream that stops calling loses its multi-GB workspace once
    WORKSPACE_CACHE_ENTRIES other (stream, purpose) pairs have been used; `release_workspaces(stream)` drops a dead
    stream's entries at once), and a workspace allocated DURING graph capture is never cached: it belongs to the
    graph's private pool, which keeps the memory alive for the graph's replays and for nobody else."""
    stream = torch.cuda.current_stream(device)
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    key = (device.type, device.index, stream.cuda_stream, key)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
        while len(_workspaces) > WORKSPACE_CACHE_ENTRIES:
            _workspaces.popitem(last=False)
    _workspaces.move_to_end(key)
    return ws


def release_workspaces(stream=None):
    """Drop the cached workspaces of `stream` (a torch.cuda.Stream about to die — its raw handle may be recycled for an
    unrelated stream), or all of them.  The caching allocator keeps the memory ordered against the stream it was used on."""
    if stream is None:
        _workspaces.clear()
        return
    for k in [k for k in _workspaces if k[2] == stream.cuda_stream and k[1] == stream.device.index]:
        del _workspaces[k]


import os as _os


def default_plan_flags():
    """Plan flags a freshly built RDN module starts with (its `plan_flags` attribute; tests pass BINHIP_PLAN_NO_FUSE /
    BINHIP_PLAN_RDB3 per call).  BIN_AMD_RDB3=1: convs 0-2 of every dense block as three phases of one launch."""
    return L.PLAN_RDB3 if _os.environ.get("BIN_AMD_RDB3", "0") == "1" else 0


def rdn_forward(weights, inputs, out=None, ws=None, flags=0, profiler=None):
    """inputs: list of fp32 [N,3,H,W] device tensors -> fp32 [N,3,H,W].  `flags`: BINHIP_PLAN_* bits; `profiler`:
    optional BinhipProfiler handle (bench.py's roofline leg).  Both are per call — the owning module keeps its own
    (`_RDNBase.plan_flags` / `.profiler`); there is no module-level switch."""
    _need_cuda(*inputs)
    with on_device(inputs[0]):
        return _rdn_forward(weights, inputs, out, ws, flags, profiler)


def _rdn_forward(weights, inputs, out, ws, flags, profiler):
    inputs = [t.contiguous().float() for t in inputs]
    n, c, h, w = inputs[0].shape
    assert c == 3 and len(inputs) == weights.n_inputs
    lib = L.lib()
    plan = L.BinRdnPlan()
    plan.N, plan.H, plan.W, plan.n_inputs, plan.nterms = n, h, w, weights.n_inputs, weights.nterms
    plan.reserved = int(flags or 0)
    plan.status = status_word(inputs[0].device).data_ptr()
    plan.profiler = profiler if profiler else None
    weights.fill_plan(plan)
    nbytes = lib.binhip_rdn_workspace_bytes(n, h, w, weights.n_inputs, weights.nterms, C.byref(plan.shape))
    if nbytes == 0:
        raise RuntimeError(f"bin_amd: unsupported RDN shape N={n} H={h} W={w} (H, W must be even)")
    if ws is None:
        ws = workspace(nbytes, inputs[0].device)
    if out is None:
        out = torch.empty((n, 3, h, w), dtype=torch.float32, device=inputs[0].device)
    arr = (C.c_void_p * len(inputs))(*[t.data_ptr() for t in inputs])
    L.check(lib.binhip_rdn_forward(C.byref(plan), arr, _ptr(out), _ptr(ws), ws.numel(), _stream()), "rdn_forward")
    return out
