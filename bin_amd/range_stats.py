"""Measured dynamic range of the fp16 chunk planes an RDN call stores (diagnostics, not on the product path).

The reference computes and stores fp32 (RDN.py:141, no AMP); here every activation / gradient between two layers is an
fp16 hi plane (+ an fp16 lo plane in the fp32-class mode), so |v| <= 65504 is a contract (include/binhip.h, "Dynamic
range") that the kernels police with a status bit.  This module answers the other question — how FAR from the limits a
given set of weights runs — by reading the planes back out of a call's workspace (`binhip_rdn_workspace_layout`,
`binhip_rdn_backward_workspace_layout` say where they are) and reducing them with torch: per stored tensor the largest
magnitude, the smallest non-zero magnitude, and the share of values below the fp16 normal range.  The pretrained
`adobe_bin.pth` is not available (model_weights/download_adobe_bin.txt:1 is a Drive link), so the numbers are taken on
the synthetic initialisation and on weights after a number of real optimisation steps
(tools/fp16_headroom.py -> profiles/r03_fp16_headroom.md; tests/test_gpu_train.py asserts the headroom).
"""
import ctypes as C

import torch

from . import _lib as L

F16_MAX = 65504.0
F16_MIN_NORMAL = 6.103515625e-05
F16_MIN_SUBNORMAL = 5.960464477539063e-08


def _layout(fn, dims, words, shape=None):
    n, h, w, k, nt = dims
    out = (C.c_int64 * words)()
    sh = L.BinRdnShape()
    if shape is not None:
        sh.G0, sh.D, sh.C, sh.G = shape
    L.check(fn(n, h, w, k, nt, C.byref(sh), out, words), "workspace_layout")
    return list(out)


def _view(ws, off, size):
    """fp16 view of `size` elements at element offset `off` of a uint8 workspace tensor (256-B aligned base)."""
    base = (-ws.data_ptr()) % 256
    return ws[base + 2 * off: base + 2 * (off + size)].view(torch.float16)


def _stats(name, hi, lo=None, scale=1.0, cls=None, kind="activation"):
    a = hi.float().abs()
    nz = a[a > 0]
    row = {"tensor": name, "class": cls or name, "kind": kind, "elements": int(a.numel()),
           "amax": float(a.max()) if a.numel() else 0.0,
           "min_nonzero": float(nz.min()) if nz.numel() else 0.0,
           "zero_share": 1.0 - nz.numel() / max(1, a.numel()),
           "subnormal_share": float((nz < F16_MIN_NORMAL).float().mean()) if nz.numel() else 0.0,
           "scale": scale}
    row["headroom"] = F16_MAX / row["amax"] if row["amax"] > 0 else float("inf")
    if lo is not None:
        # what the hi+lo pair cannot represent: the residual of values whose lo part fell below the fp16 subnormal step
        row["lo_amax"] = float(lo.float().abs().max()) if lo.numel() else 0.0
    return row


def forward_stats(ws, dims, tag="", shape=(96, 12, 4, 32)):
    """Rows for X0, F1, every dense block's input and its four conv outputs, G0, G1, U of ONE forward call whose
    workspace `ws` was filled with BINHIP_PLAN_KEEP_ACTS (the training forward).  dims = (N, H, W, n_inputs, nterms)."""
    v = _layout(L.lib().binhip_rdn_workspace_layout, dims, L.RDN_LAYOUT_WORDS, shape)
    G0, D, Cc, G = shape
    c0, cg, cb = G0 // 16, G // 16, (G0 + Cc * G) // 16
    P, PF, kc0 = v[0], v[1], v[2]
    has_lo = bool(v[15])
    rows = []

    def add(name, off, size, total, cls=None):   # `total`: hi size of the WHOLE tensor the range belongs to (lo = hi + total)
        hi = _view(ws, off, size)
        lo = _view(ws, off + total, size) if has_lo else None
        rows.append(_stats(tag + name, hi, lo, cls=cls or name))

    add("X0 (packed frames)", v[3], v[4], v[4])
    add("F1 = SFENet1", v[5], v[6], v[6])
    blk, s_blk = v[7], v[8]
    for d in range(D + 1):
        b = blk + d * cb * P
        add("SFENet2 out" if d == 0 else f"RDB{d - 1} out", b, c0 * P, s_blk, "SFENet2 / dense-block outputs")
        if d < D:
            for c in range(Cc):
                o = b + (c0 + cg * c) * P
                add(f"RDB{d}.conv{c} out", o, cg * P, s_blk, "dense-block conv outputs (post-ReLU)")
    add("G0 = GFF.0", v[9], v[10], v[10])
    add("G1 = GFF.1 + F1", v[11], v[12], v[12])
    add("U = UPNet.0 shuffled", v[13], v[14], v[14])
    return rows


def backward_stats(ws, dims, tag="", input_grads=True, shape=(96, 12, 4, 32)):
    """Rows for the gradient planes left in the backward workspace of ONE call (stored x its power-of-two scale).
    `input_grads=False`: the call produced no input-frame gradients (stage 1 of the pyramid reads the raw frames), so its
    gX0 planes were never written and are reported as empty."""
    v = _layout(L.lib().binhip_rdn_backward_workspace_layout, dims, L.RDN_BWD_LAYOUT_WORDS, shape)
    G0, D = shape[0], shape[1]
    c0 = G0 // 16
    P = v[0]
    nt = dims[4]
    base = (-ws.data_ptr()) % 256
    sc = ws[base + v[23]: base + v[23] + 8].view(torch.float32)
    scale = float(sc[0])
    rows = []

    def add(name, off, size, total, cls=None):
        hi = _view(ws, off, size)
        lo = _view(ws, off + total, size) if nt == 3 else None
        rows.append(_stats(tag + name, hi, lo, scale, cls=cls or name, kind="gradient"))

    names = ("g out", "g U", "g U unshuffled", "g G1", "g G0", "g F1")
    for i, nm in enumerate(names):
        add(nm, v[3 + 2 * i], v[4 + 2 * i], v[4 + 2 * i])
    gy, s_gy = v[15], v[16]
    for d in range(D + 1):
        add("g SFENet2 out" if d == 0 else f"g RDB{d - 1} out", gy + d * c0 * P, c0 * P, s_gy, "g SFENet2 / dense-block outputs")
    add("g concat (last even block)", v[17], v[18], v[18], "g dense-block concat (blocks 0, 1)")
    add("g concat (last odd block)", v[19], v[20], v[20], "g dense-block concat (blocks 0, 1)")
    if input_grads:
        add("g X0", v[21], v[22], v[22])
    else:
        rows.append({"tensor": tag + "g X0", "class": "g X0", "kind": "gradient", "elements": 0, "amax": 0.0, "min_nonzero": 0.0,
                     "zero_share": 1.0, "subnormal_share": 0.0, "scale": scale, "headroom": float("inf")})
    return rows


def summarize(rows):
    """Worst cases over a list of rows."""
    live = [r for r in rows if r["amax"] > 0]
    return {"tensors": len(rows),
            "min_headroom": min(r["headroom"] for r in live) if live else float("inf"),
            "worst_tensor": min(live, key=lambda r: r["headroom"])["tensor"] if live else None,
            "amax": max(r["amax"] for r in live) if live else 0.0,
            "min_nonzero": min(r["min_nonzero"] for r in live if r["min_nonzero"] > 0) if live else 0.0,
            "max_subnormal_share": max(r["subnormal_share"] for r in live) if live else 0.0}


class Recorder:
    """debug_hook for the RDN modules of a network: collects forward / backward rows while `armed`."""

    def __init__(self):
        self.armed = False
        self.rows = []
        self.tag = ""

    def attach(self, net):
        for m in net.rdn_modules():
            m.debug_hook = self
        return self

    def detach(self, net):
        for m in net.rdn_modules():
            m.debug_hook = None

    def __call__(self, kind, module, dims, ws, info):
        if not self.armed:
            return
        torch.cuda.synchronize()
        tag = f"{self.tag}{type(module).__name__[len('RDN_residual_interp_'):]} N={dims[0]} "
        shape = info.get("shape", (96, 12, 4, 32))
        self.rows += (forward_stats(ws, dims, tag, shape) if kind == "forward"
                      else backward_stats(ws, dims, tag, info.get("input_grads", True), shape))
