"""Host side of one RDN sub-network (reference RDN.py:167-334): relayout its 66 convolutions once,
then run the whole sub-network with a single C call (binhip_rdn_forward)."""
import ctypes as C

import torch

from . import _lib as L
from .ops import ConvWeights, _ptr, _stream, _need_cuda

D_BLOCKS, C_CONVS = 12, 4


def layer_names():
    """Local parameter prefixes of the 66 layers in launch order (matches binhip_plan.hip)."""
    names = ["SFENet1", "SFENet2"]
    for d in range(D_BLOCKS):
        names += [f"RDBs.{d}.convs.{c}.conv.0" for c in range(C_CONVS)]
        names.append(f"RDBs.{d}.LFF")
    names += ["GFF.0", "GFF.1", "UPNet.0", "UPNet.2"]
    return names


class RdnWeights:
    """Kernel-layout weights of one RDN weight set + the BinRdnPlan pointer table."""

    def __init__(self, params, n_inputs, nterms, prefix=""):
        self.n_inputs, self.nterms = n_inputs, nterms
        self.layers = []
        for i, nm in enumerate(layer_names()):
            w, b = params[f"{prefix}{nm}.weight"], params[f"{prefix}{nm}.bias"]
            shuffle = nm == "UPNet.0"
            cin_chunks = None
            if nm == "SFENet1":
                cin_chunks = (12 * n_inputs + 15) // 16
            self.layers.append(ConvWeights(w, b, nterms=nterms, shuffle=shuffle, cin_chunks=cin_chunks))
        assert len(self.layers) == L.RDN_LAYERS

    def fill_plan(self, plan):
        for i, cw in enumerate(self.layers):
            plan.w_hi[i] = cw.w_hi.data_ptr()
            plan.w_lo[i] = cw.w_lo.data_ptr() if cw.w_lo is not None else None
            plan.bias[i] = cw.bias.data_ptr()


_workspaces = {}


def workspace(nbytes, device):
    """One cached workspace per device, grown on demand (activations of a single RDN call)."""
    key = (device.type, device.index)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def rdn_forward(weights, inputs, out=None, ws=None):
    """inputs: list of fp32 [N,3,H,W] device tensors -> fp32 [N,3,H,W]."""
    _need_cuda(*inputs)
    inputs = [t.contiguous().float() for t in inputs]
    n, c, h, w = inputs[0].shape
    assert c == 3 and len(inputs) == weights.n_inputs
    lib = L.lib()
    plan = L.BinRdnPlan()
    plan.N, plan.H, plan.W, plan.n_inputs, plan.nterms = n, h, w, weights.n_inputs, weights.nterms
    weights.fill_plan(plan)
    nbytes = lib.binhip_rdn_workspace_bytes(n, h, w, weights.n_inputs, weights.nterms)
    if nbytes == 0:
        raise RuntimeError(f"bin_amd: unsupported RDN shape N={n} H={h} W={w} (H, W must be even)")
    if ws is None:
        ws = workspace(nbytes, inputs[0].device)
    if out is None:
        out = torch.empty((n, 3, h, w), dtype=torch.float32, device=inputs[0].device)
    arr = (C.c_void_p * len(inputs))(*[t.data_ptr() for t in inputs])
    L.check(lib.binhip_rdn_forward(C.byref(plan), arr, _ptr(out), _ptr(ws), ws.numel(), _stream()), "rdn_forward")
    return out
