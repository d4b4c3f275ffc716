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


def fused_upnet_weights(w0, b0, w2, b2):
    """UPNet (RDN.py:203-207) = conv3x3(G0 -> 256) -> PixelShuffle(2) -> conv3x3(64 -> 3) has no activation between its layers, so it is
    ONE linear map: a 5x5 convolution G0 -> 12 at half resolution whose output channel c * 4 + i * 2 + j is colour c at sub-pixel (i, j):
        O[c, 2y+i, 2x+j] = b2[c] + sum_{k,dy,dx} W2[c,k,dy,dx] * U[k, 2y+i+dy-1, 2x+j+dx-1],   U[k, 2y'+i', 2x'+j'] = (W0 * x + b0)[4k+2i'+j', y', x']
    with y' = y + floor((i+dy-1)/2), i' = (i+dy-1) mod 2: the 3x3 of W0 lands at offset (oy, ox) = (floor((i+dy-1)/2), floor((j+dx-1)/2))
    inside the 5x5.  UPNet.2 zero-pads U, not x: where its window leaves the full-resolution image the tap contributes NOTHING (not even b0), so
    the outermost full-resolution pixel ring has its own operators.  Returns (W [9, 12, G0, 5, 5], b [9, 12]) in float64; variant 3 vy + vx,
    v = 0 first row (column), 1 interior, 2 last row (column); only the sub-pixels that lie ON the ring differ from the interior form.
    3.4 x fewer multiply-adds than the two layers (25 * 12 * G0 against 9 * 256 * G0 + 4 * 9 * 64 * 3 per half-resolution pixel)."""
    w0, b0, w2, b2 = (t.double() for t in (w0, b0, w2, b2))          # differentiable: a caller may backpropagate dW_eff -> dW0, dW2
    dev, g0 = w0.device, w0.shape[1]
    assert w0.shape == (256, g0, 3, 3) and w2.shape == (3, 64, 3, 3) and b0.shape == (256,) and b2.shape == (3,)
    # (i, dy) -> (oy, i'): s = i + dy - 1 in {-1, 0, 1, 2}, oy = floor(s / 2), i' = s mod 2 — the same table serves (j, dx)
    sub = torch.tensor([[1, 0, 1], [0, 1, 0]], device=dev)              # i'[i][dy]
    off = torch.tensor([[-1, 0, 0], [0, 0, 1]], device=dev)             # oy[i][dy]
    w0r = w0.view(64, 2, 2, g0, 3, 3)                                   # [k, i', j', g, a, b]  (channel 4 k + 2 i' + j')
    w0g = w0r[:, sub.reshape(-1)].view(64, 2, 3, 2, g0, 3, 3)           # [k, i, dy, j', g, a, b]
    w0g = w0g[:, :, :, sub.reshape(-1)].view(64, 2, 3, 2, 3, g0, 3, 3)  # [k, i, dy, j, dx, g, a, b]
    b0g = b0.view(64, 2, 2)[:, sub.reshape(-1)].view(64, 2, 3, 2)[:, :, :, sub.reshape(-1)].view(64, 2, 3, 2, 3)
    T = torch.einsum("ckyx,kiyjxgab->ciyjxgab", w2, w0g)                # UPNet.2's tap (dy, dx) through UPNet.0, per sub-pixel (i, j)
    Tb = torch.einsum("ckyx,kiyjx->ciyjx", w2, b0g)
    # where a tap's 3x3 lands inside the 5x5: row oy + 1 + a
    place = torch.zeros(2, 3, 3, 5, dtype=torch.float64, device=dev)    # [i, dy, a, t]
    for i in range(2):
        for dy in range(3):
            for a in range(3):
                place[i, dy, a, int(off[i, dy]) + 1 + a] = 1.0
    # which taps exist: the first row's upper sub-pixel cannot look up, the last row's lower sub-pixel cannot look down
    keep = torch.ones(3, 2, 3, dtype=torch.float64, device=dev)         # [v, i, dy]
    keep[0, 0, 0] = 0.0
    keep[2, 1, 2] = 0.0
    W = torch.einsum("ciyjxgab,viy,wjx,iyat,jxbu->vwcijgtu", T, keep, keep, place, place).reshape(9, 12, g0, 5, 5)
    B = (b2.view(1, 1, 3, 1, 1) + torch.einsum("ciyjx,viy,wjx->vwcij", Tb, keep, keep)).reshape(9, 12)
    return W, B


_FUSED_MAPS = {}


def _fused_upnet_maps(device):
    """Constant index maps of `fused_upnet_operator` (cached per device): the operator is BILINEAR in (W2, W0), so for fixed W2 it is the
    matrix product A(W2) @ W0 with A[(vy,vx,c,i,j,t,u), (k,i',j',a,b)] = W2[c,k,dy,dx] for the ONE tap (dy, dx) that sends UPNet.0's 3x3
    entry (a, b) of channel 4k+2i'+j' to entry (t, u) of the 5x5 of sub-pixel (i, j) — or 0 (no such tap, or the tap is cut by the border
    variant).  `idx` indexes W2.flatten() extended by one zero."""
    key = str(device)
    if key in _FUSED_MAPS:
        return _FUSED_MAPS[key]
    import numpy as np
    # tap[i, i', t, a] = dy with (i + dy - 1) mod 2 == i' and floor((i + dy - 1) / 2) + 1 + a == t, else -1 (same table for j / dx / u / b)
    tap = -np.ones((2, 2, 5, 3), dtype=np.int64)
    sub = np.zeros((2, 3), dtype=np.int64)
    for i in range(2):
        for dy in range(3):
            oy, ip = divmod(i + dy - 1, 2)
            sub[i, dy] = ip
            for a in range(3):
                tap[i, ip, oy + 1 + a, a] = dy
    keep = np.ones((3, 2, 3), dtype=bool)          # [v, i, dy]: the first row's upper sub-pixel cannot look up, the last row's lower one not down
    keep[0, 0, 0] = False
    keep[2, 1, 2] = False
    # broadcast to [vy, vx, c, i, j, t, u, k, i', j', a, b]
    sh = (3, 3, 3, 2, 2, 5, 5, 64, 2, 2, 3, 3)
    ax = {n: k for k, n in enumerate(("vy", "vx", "c", "i", "j", "t", "u", "k", "ip", "jp", "a", "b"))}

    def bc(arr, names):
        shape = [1] * len(sh)
        for d, nme in zip(arr.shape, names):
            shape[ax[nme]] = d
        order = sorted(range(len(names)), key=lambda q: ax[names[q]])
        return np.transpose(arr, order).reshape(shape)
    dy = bc(tap, ("i", "ip", "t", "a"))
    dx = bc(tap, ("j", "jp", "u", "b"))
    ok = (dy >= 0) & (dx >= 0)
    vy = np.arange(3).reshape(bc(np.arange(3), ("vy",)).shape)
    vx = np.arange(3).reshape(bc(np.arange(3), ("vx",)).shape)
    ii = np.arange(2).reshape(bc(np.arange(2), ("i",)).shape)
    jj = np.arange(2).reshape(bc(np.arange(2), ("j",)).shape)
    valid = ok & keep[vy, ii, np.where(dy >= 0, dy, 0)] & keep[vx, jj, np.where(dx >= 0, dx, 0)]
    cc = np.arange(3).reshape(bc(np.arange(3), ("c",)).shape)
    kk = np.arange(64).reshape(bc(np.arange(64), ("k",)).shape)
    flat = ((cc * 64 + kk) * 3 + np.where(dy >= 0, dy, 0)) * 3 + np.where(dx >= 0, dx, 0)
    idx = np.where(np.broadcast_to(valid, sh), np.broadcast_to(flat, sh), 3 * 64 * 9).reshape(-1)
    # the inverse map for the backward: every W2 entry sits at <= 324 places of A; gathering those (padded with an index one past the
    # end = a zero) and summing is deterministic, where index_select's own backward is 6.2 M atomic adds onto 1728 addresses
    nA, nW = idx.size, 3 * 64 * 9
    order = np.argsort(idx, kind="stable")
    srt = idx[order]
    counts = np.bincount(srt, minlength=nW + 1)
    starts = np.concatenate(([0], np.cumsum(counts)[:-1]))
    rank = np.arange(nA) - starts[srt]
    lmax = int(counts[:nW].max())
    inv = np.full((nW, lmax), nA, dtype=np.int64)
    real = srt < nW
    inv[srt[real], rank[real]] = order[real]
    maps = {"idx": torch.from_numpy(idx.astype(np.int32)).to(device), "inv": torch.from_numpy(inv.astype(np.int32)).to(device),
            "sub": torch.from_numpy(sub).to(device), "keep": torch.from_numpy(keep.astype(np.float32)).to(device)}
    _FUSED_MAPS[key] = maps
    return maps


class _GatherW2(torch.autograd.Function):
    """A = [W2.flatten(), 0][idx] with a deterministic, atomics-free backward (gather through the inverse map + sum)."""

    @staticmethod
    def forward(ctx, w2, idx, inv):
        ctx.save_for_backward(inv)
        ctx.shape = w2.shape
        return torch.cat((w2.reshape(-1), w2.new_zeros(1))).index_select(0, idx)

    @staticmethod
    def backward(ctx, gA):
        inv, = ctx.saved_tensors
        ge = torch.cat((gA.reshape(-1), gA.new_zeros(1)))
        return ge.index_select(0, inv.reshape(-1)).view(inv.shape).sum(1).view(ctx.shape), None, None


def fused_upnet_operator(w0, b0, w2, b2):
    """The same operators as `fused_upnet_weights`, as ONE gather + ONE matrix product in the parameters' dtype (differentiable: a handful of
    launches forward and backward) — what a training step builds per weight set.  Returns (ring layout [9, 12, 25, G0], [9, 12]);
    operator v as a convolution weight is `out[v].permute(0, 2, 1).reshape(12, G0, 5, 5)`."""
    m = _fused_upnet_maps(w0.device)
    g0 = w0.shape[1]
    A = _GatherW2.apply(w2, m["idx"], m["inv"]).view(9 * 12 * 25, 64 * 2 * 2 * 9)
    w0p = w0.view(64, 2, 2, g0, 3, 3).permute(0, 1, 2, 4, 5, 3).reshape(64 * 2 * 2 * 9, g0)
    W = (A @ w0p).view(9, 12, 25, g0)
    sub, keep = m["sub"], m["keep"].to(w0.dtype)
    tb = torch.einsum("ckyx,kpq->cyxpq", w2, b0.view(64, 2, 2))                      # [c, dy, dx, i', j']
    tg = tb[:, :, :, sub.reshape(-1)].view(3, 3, 3, 2, 3, 2)                          # [c, dy, dx, i, dy', j'] -> take dy' == dy below
    tg = torch.diagonal(tg, dim1=1, dim2=4)                                           # [c, dx, i, j', dy]
    tg = tg[:, :, :, sub.reshape(-1)].reshape(3, 3, 2, 2, 3, 3)                       # [c, dx, i, j, dx', dy]
    tg = torch.diagonal(tg, dim1=1, dim2=4)                                           # [c, i, j, dy, dx]
    B = b2.view(1, 1, 3, 1, 1) + torch.einsum("cijyx,viy,wjx->vwcij", tg, keep, keep)
    return W, B.reshape(9, 12)


def fused_upnet_reference(x, W, B):
    """Plain-torch statement of what the fused UPNet computes from `fused_upnet_weights` (tests; the device path is
    BINHIP_PLAN_FUSED_UPNET): the interior operator everywhere, then the full-resolution border ring from its own variants."""
    import torch.nn.functional as F
    x = x.double()
    out = F.pixel_shuffle(F.conv2d(x, W[4], B[4], padding=2), 2)
    h2, w2 = out.shape[-2:]
    for vy in range(3):
        for vx in range(3):
            if vy == 1 and vx == 1:
                continue
            full = F.pixel_shuffle(F.conv2d(x, W[3 * vy + vx], B[3 * vy + vx], padding=2), 2)
            ys = {0: slice(0, 1), 1: slice(1, h2 - 1), 2: slice(h2 - 1, h2)}[vy]
            xs = {0: slice(0, 1), 1: slice(1, w2 - 1), 2: slice(w2 - 1, w2)}[vx]
            out[..., ys, xs] = full[..., ys, xs]
    return out


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
        # the fused UPNet's operands (BINHIP_PLAN_FUSED_UPNET) are built on the first INFERENCE call of this weight version
        # (ensure_fused_upnet): a training step rebuilds its RdnWeights every step and never needs them
        self._up_src = tuple(params[f"{prefix}UPNet.{k}.{t}"] for k in (0, 2) for t in ("weight", "bias"))
        self.fused_up = None
        self.fused_w4 = None
        self.fused_graph = None

    def ensure_fused_upnet(self, train=False):
        """(ConvWeights of the [12][G0][5][5] interior operator, fp32 [9][12][25][G0] ring operators, fp32 [9][12] ring biases) or None when
        this weight set's UPNet is not the 256 -> shuffle -> 3 one.  `train`: build them with `fused_upnet_operator` under autograd and
        keep the graph (`fused_graph` = leaves, operators, biases): the backward maps the operators' gradients to UPNet.0 / UPNet.2."""
        if self.fused_up is None and self._up_src is not None and train:
            w0, b0, w2, b2 = self._up_src
            if tuple(w0.shape[2:]) == (3, 3) and w0.shape[0] == 256 and tuple(w2.shape) == (3, 64, 3, 3):
                with torch.enable_grad():
                    leaves = [t.detach().float().requires_grad_() for t in (w0, b0, w2, b2)]
                    Wr, Br = fused_upnet_operator(*leaves)
                self.fused_graph = (leaves, Wr, Br)
                with torch.no_grad():
                    g0 = Wr.shape[3]
                    self.fused_w4 = Wr[4].permute(0, 2, 1).reshape(12, g0, 5, 5).contiguous()
                    main = ConvWeights(self.fused_w4, Br[4].detach().contiguous(), nterms=self.nterms)
                    self.fused_up = (main, Wr.detach().contiguous(), Br.detach().contiguous())
            self._up_src = None
        if self.fused_up is None and self._up_src is not None:
            with torch.no_grad():
                w0, b0, w2, b2 = self._up_src
                if tuple(w0.shape[2:]) == (3, 3) and w0.shape[0] == 256 and tuple(w2.shape) == (3, 64, 3, 3):
                    W, B = fused_upnet_weights(w0, b0, w2, b2)
                    self.fused_w4 = W[4].float().contiguous()               # (RdnDgradWeights relayouts its transpose for the backward)
                    main = ConvWeights(self.fused_w4, B[4].float().contiguous(), nterms=self.nterms)
                    ring_w = W.permute(0, 1, 3, 4, 2).reshape(9, 12, 25, W.shape[2]).float().contiguous()
                    self.fused_up = (main, ring_w, B.float().contiguous())
            self._up_src = None
        return self.fused_up

    def fill_plan(self, plan):
        plan.shape = c_shape(self.shape)
        for i, cw in enumerate(self.layers):
            plan.w_hi[i] = cw.w_hi.data_ptr()
            plan.w_lo[i] = cw.w_lo.data_ptr() if cw.w_lo is not None else None
            plan.bias[i] = cw.bias.data_ptr()
        n = len(self.layers)
        if self.fused_up is not None:
            main, ring_w, ring_b = self.fused_up
            plan.w_hi[n], plan.bias[n] = main.w_hi.data_ptr(), main.bias.data_ptr()
            plan.w_lo[n] = main.w_lo.data_ptr() if main.w_lo is not None else None
            plan.w_hi[n + 1], plan.w_lo[n + 1], plan.bias[n + 1] = ring_w.data_ptr(), None, ring_b.data_ptr()

    def dgrad(self, module, nterms=None):
        """Backward-data weights (transposed + flipped), built on first use for the same parameter version; `nterms`
        other than the forward's for the single-product backward behind an f16x3 forward."""
        nterms = self.nterms if nterms is None else nterms
        if self._dgrad is None:
            self._dgrad = {}
        if nterms not in self._dgrad:
            with torch.no_grad():
                self._dgrad[nterms] = RdnDgradWeights(dict(module.named_parameters()), self.n_inputs, nterms,
                                                      shape=self.shape,
                                                      fused=(self.fused_w4, self.fused_up[1]) if self.fused_up is not None else None)
        return self._dgrad[nterms]


class RdnDgradWeights:
    """Backward-data weights of the 66 layers in kernel layout.

    Plain layers: W'[ci][co][dy][dx] = W[co][ci][k-1-dy][k-1-dx] (binhip_weights_relayout_dgrad).
    The four 3x3 convs of each residual dense block are stored in GATHER form (binhip_weights_relayout_rdb_gather):
    slot `RDBs.d.convs.g` holds the weights that produce concat-group g of the block from the stacked output
    gradients of convs g..3 - what binhip_rdn_backward expects (include/binhip.h, BinRdnBwdPlan)."""

    def __init__(self, params, n_inputs, nterms, prefix="", shape=STAGE4_SHAPE, fused=None):
        lib = L.lib()
        self.shape = G0, D, C, G = check_shape(shape)
        self.w_hi, self.w_lo = [], []
        self.fused_ring_w = None
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
        if fused is not None:
            # BINHIP_BWD_FUSED_UPNET: slot L = the transposed interior operator of the fused UPNet ([12][G0][5][5] -> 5x5, 12 -> G0),
            # slot L + 1 = its fp32 ring operators as the forward uses them
            w4, self.fused_ring_w = fused
            rows_pad = lib.binhip_dgrad_rows_pad(5, G0)
            cb = lib.binhip_conv_cout_block(5, rows_pad, nterms)
            hi, lo, zb = alloc(rows_pad, 1, 5)
            items.append(relayout_item(L.RELAYOUT_DGRAD, [w4], None, hi, lo, zb, 12, G0, 5, rows_pad, 1, cb, 0))
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
        if self.fused_ring_w is not None:
            plan.wt_hi[len(self.w_hi)] = self.fused_ring_w.data_ptr()
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
    flags = L.PLAN_RDB3 if _os.environ.get("BIN_AMD_RDB3", "0") == "1" else 0
    # round 6: UPNet (conv3x3 -> PixelShuffle -> conv3x3, no activation in between) as ONE 5x5 convolution in inference (fp32-class
    # mode; same function up to fp32 summation order, 3.4 x fewer multiply-adds).  BIN_AMD_FUSED_UPNET=0: the two layers, as in training
    if _os.environ.get("BIN_AMD_FUSED_UPNET", "1") != "0":
        flags |= L.PLAN_FUSED_UPNET
    return flags


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
    if (plan.reserved & L.PLAN_FUSED_UPNET) and (not (plan.reserved & L.PLAN_KEEP_ACTS) or (plan.reserved & L.PLAN_FUSED_UPNET_TRAIN)):
        weights.ensure_fused_upnet()
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
