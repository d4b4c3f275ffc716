"""Thin torch-tensor wrappers over the libbinhip C ABI (include/binhip.h).

PyTorch is plumbing here (device memory, streams); every computation is a hand-written HIP kernel.
Activations between layers are "chunk planes" (CP): fp16 [C/16][N][H][W][16], optionally hi+lo.
"""
import ctypes as C
import os

import torch

from . import _lib as L


def _stream(device=None):
    """The caller's current stream ON `device` (default: the current device).  Library calls launch on the current
    HIP device, so entry points that take tensors run under `on_device(t)`."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_device(t):
    """Context that makes `t`'s device current (a no-op when it already is): kernels launch on the CURRENT device and
    on its stream, so a tensor living on another GPU of the same process must switch first."""
    return torch.cuda.device(t.device)


_status = {}


def status_word(device):
    """Per-device uint32 status word the kernels OR into (include/binhip.h BINHIP_STATUS_*)."""
    key = _device_key(device)
    w = _status.get(key)
    if w is None:
        w = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", key))
        _status[key] = w
    return w


def _device_key(device):
    device = torch.device(device)
    return device.index if device.index is not None else torch.cuda.current_device()


def check_status(device=None, reset=True):
    """Raise if any kernel since the last check reported a problem in its device status word: a stored value outside
    the fp16 range (or a NaN), a neighbour-flag timeout of the three-phase dense-block launch, or any bit this host
    code does not know.  Reads the device word, i.e. synchronises: call it where the host syncs anyway (after a step,
    before images leave the device).  `device=None` checks every device that has a word; a device without an index
    means the current one (as in `status_word`)."""
    keys = list(_status) if device is None else [_device_key(device)]
    for k in keys:
        w = _status.get(k)
        if w is None:
            continue
        v = int(w.item()) & 0xFFFFFFFF
        if not v:
            continue
        if reset:
            w.zero_()
        if v & L.STATUS_SYNC_TIMEOUT:
            raise RuntimeError(
                "bin_amd: dense-block launch on cuda:%d timed out waiting for a neighbour tile (BINHIP_STATUS_SYNC_TIMEOUT): "
                "a tile was computed from inputs that may not have been published; results since the last check are "
                "invalid" % k)
        if v & L.STATUS_SATURATED:
            raise RuntimeError(
                "bin_amd: fp16 range exceeded on cuda:%d — an activation / gradient left +-65504 (or was NaN) and was "
                "saturated; results since the last check are not those of the fp32 reference (include/binhip.h, "
                "'Dynamic range')" % k)
        raise RuntimeError("bin_amd: unknown status bits 0x%x on cuda:%d (library newer than this host code?)" % (v, k))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("bin_amd: tensors must live on a HIP device (there is no CPU path; "
                               "the CPU restatement lives in oracle/ and is test-only)")


def chunks(c):
    return (c + 15) // 16


class CP:
    """A chunk-plane tensor: `hi` (and `lo` when split) are fp16 [nchunks, N, H, W, 16]."""

    def __init__(self, hi, lo, channels):
        self.hi, self.lo, self.channels = hi, lo, channels

    @property
    def shape(self):
        return tuple(self.hi.shape)

    @staticmethod
    def empty(nchunks, n, h, w, nterms, device, channels=None):
        hi = torch.empty((nchunks, n, h, w, 16), dtype=torch.float16, device=device)
        lo = torch.empty_like(hi) if nterms == 3 else None
        return CP(hi, lo, channels if channels is not None else nchunks * 16)

    def sub(self, c0, nch):
        return CP(self.hi[c0:c0 + nch], self.lo[c0:c0 + nch] if self.lo is not None else None, nch * 16)


def nchw_to_planes(x, nterms=1):
    _need_cuda(x)
    x = x.contiguous().float()
    n, c, h, w = x.shape
    y = CP.empty(chunks(c), n, h, w, nterms, x.device, c)
    L.check(L.lib().binhip_nchw_to_planes(_ptr(x), n, c, h, w, _ptr(y.hi), _ptr(y.lo), _ptr(status_word(x.device)),
                                          _stream()), "nchw_to_planes")
    return y


def planes_to_nchw(cp, channels=None):
    c = channels if channels is not None else cp.channels
    _, n, h, w, _ = cp.hi.shape
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=cp.hi.device)
    L.check(L.lib().binhip_planes_to_nchw(_ptr(cp.hi), _ptr(cp.lo), n, c, h, w, _ptr(y), _stream()), "planes_to_nchw")
    return y


def pack_inputs(images, nterms=1):
    """K1: pixel_reshuffle(cat(images), 2) -> CP at half resolution (reference RDN.py:107-132)."""
    _need_cuda(*images)
    images = [im.contiguous().float() for im in images]
    n, c, h, w = images[0].shape
    assert c == 3 and h % 2 == 0 and w % 2 == 0
    k = len(images)
    y = CP.empty(chunks(12 * k), n, h // 2, w // 2, nterms, images[0].device, 12 * k)
    arr = (C.c_void_p * k)(*[im.data_ptr() for im in images])
    L.check(L.lib().binhip_pack_inputs(arr, k, n, h, w, _ptr(y.hi), _ptr(y.lo), _ptr(status_word(images[0].device)),
                                       _stream()), "pack_inputs")
    return y


def relayout_item(kind, srcs, bias, w_hi, w_lo, bias_out, cout, cin, ks, rows_pad, cin_chunks, cout_block, shuffle_or_group,
                  shape=None):
    """One BinRelayoutItem (include/binhip.h).  `srcs`: the fp32 OIHW source tensor(s); the caller keeps them alive until
    relayout_batch() has enqueued the launch."""
    it = L.BinRelayoutItem()
    for i, t in enumerate(srcs):
        it.w[i] = t.data_ptr() if t is not None else None
    it.bias = bias.data_ptr() if bias is not None else None
    it.w_hi, it.w_lo = w_hi.data_ptr(), (w_lo.data_ptr() if w_lo is not None else None)
    it.bias_out = bias_out.data_ptr()
    it.kind, it.cout, it.cin, it.ksize, it.rows_pad = kind, cout, cin, ks, rows_pad
    it.cin_chunks, it.cout_block, it.shuffle_or_group = cin_chunks, cout_block, shuffle_or_group
    if shape is not None:                  # RDB_GATHER of a block other than bin_stage4's
        it.shape.G0, it.shape.D, it.shape.C, it.shape.G = shape
    return it


def relayout_batch(items):
    """Enqueue the relayouts of `items` (list of BinRelayoutItem) in as few launches as the library needs.
    BIN_AMD_RELAYOUT_BATCH=0 (A/B switch of tools/): one launch per item, as before round 3."""
    if not items:
        return
    lib = L.lib()
    if os.environ.get("BIN_AMD_RELAYOUT_BATCH", "1") == "0":
        for it in items:
            L.check(lib.binhip_weights_relayout_batch(C.byref(it), 1, _stream()), "weights_relayout_batch")
        return
    arr = (L.BinRelayoutItem * len(items))(*items)
    L.check(lib.binhip_weights_relayout_batch(arr, len(items), _stream()), "weights_relayout_batch")


class ConvWeights:
    """Kernel-layout weights of one convolution (see binhip_weights_relayout).  `defer`: a list that receives this
    layer's relayout item instead of launching it (RdnWeights batches the 66 layers of a weight set)."""

    def __init__(self, weight, bias, nterms=1, shuffle=False, cout_pad=None, cin_chunks=None, defer=None):
        _need_cuda(weight)
        cout, cin, ks, _ = weight.shape
        self.cout, self.cin, self.ks, self.nterms, self.shuffle = cout, cin, ks, nterms, shuffle
        self.cout_pad = cout_pad if cout_pad is not None else ((cout + 31) // 32) * 32
        self.cin_chunks = cin_chunks if cin_chunks is not None else chunks(cin)
        lib = L.lib()
        self.cout_block = lib.binhip_conv_cout_block(ks, self.cout_pad, nterms)
        nbytes = lib.binhip_weights_bytes(self.cout_pad, self.cin_chunks, ks)
        dev = weight.device
        self.w_hi = torch.empty(nbytes // 2, dtype=torch.float16, device=dev)
        self.w_lo = torch.empty(nbytes // 2, dtype=torch.float16, device=dev) if nterms == 3 else None
        self.bias = torch.empty(self.cout_pad, dtype=torch.float32, device=dev)
        w = weight.detach().contiguous().float()
        b = bias.detach().contiguous().float() if bias is not None else None
        if defer is not None:
            self._src = (w, b)
            defer.append(relayout_item(L.RELAYOUT_FWD, [w], b, self.w_hi, self.w_lo, self.bias, cout, cin, ks, self.cout_pad,
                                       self.cin_chunks, self.cout_block, 1 if shuffle else 0))
            return
        L.check(lib.binhip_weights_relayout(_ptr(w), _ptr(b), cout, cin, ks, self.cout_pad, self.cin_chunks,
                                            self.cout_block, 1 if shuffle else 0, _ptr(self.w_hi), _ptr(self.w_lo),
                                            _ptr(self.bias), _stream()), "weights_relayout")


def conv2d(x, cw, relu=False, residual=None, out=None, epilogue=L.EPI_PLANES, images=None, x_cpg=0,
           x_group_stride=0, cin_chunks=None):
    """One fused convolution launch.  x: CP (its first `cin_chunks` planes are read);
    PLANES/SHUFFLE return a CP, FINAL returns fp32 NCHW."""
    nch, n, h, w, _ = x.hi.shape
    d = L.BinConvDesc()
    d.N, d.H, d.W, d.ksize = n, h, w, cw.ks
    d.cin_chunks = cin_chunks if cin_chunks is not None else cw.cin_chunks
    d.cout, d.cout_pad, d.nterms, d.epilogue, d.relu = cw.cout, cw.cout_pad, cw.nterms, epilogue, int(relu)
    d.x_cpg, d.x_group_stride = x_cpg, x_group_stride
    d.n_images = len(images) if images else 0
    # BINHIP_CONV_HALF_LAST_CHUNK: a 5x5 layer whose last chunk holds <= 8 real channels (x from nchw_to_planes / pack_inputs and
    # the relayouted weights are zero there) may spend that chunk's K on tap pairs
    if cw.ks == 5 and 1 <= cw.cin % 16 <= 8 and d.cin_chunks == cw.cin_chunks:
        d.reserved = L.CONV_HALF_LAST_CHUNK
    dev = x.hi.device
    d.status = status_word(dev).data_ptr()
    y_f32, arr = None, None
    if epilogue == L.EPI_PLANES:
        if out is None:
            out = CP.empty(chunks(cw.cout), n, h, w, cw.nterms, dev, cw.cout)
    elif epilogue == L.EPI_SHUFFLE:
        if out is None:
            out = CP.empty(chunks(cw.cout // 4), n, 2 * h, 2 * w, cw.nterms, dev, cw.cout // 4)
    else:
        y_f32 = torch.empty((n, cw.cout, h, w), dtype=torch.float32, device=dev)
        if images:
            images = [im.contiguous().float() for im in images]
            arr = (C.c_void_p * len(images))(*[im.data_ptr() for im in images])
    rc = L.lib().binhip_conv2d_fwd(
        C.byref(d), _ptr(x.hi), _ptr(x.lo), _ptr(cw.w_hi), _ptr(cw.w_lo), _ptr(cw.bias),
        _ptr(residual.hi) if residual is not None else C.c_void_p(0),
        _ptr(residual.lo) if residual is not None else C.c_void_p(0),
        _ptr(out.hi) if out is not None else C.c_void_p(0),
        _ptr(out.lo) if out is not None else C.c_void_p(0),
        _ptr(y_f32), arr, _stream())
    L.check(rc, "conv2d_fwd")
    return y_f32 if epilogue == L.EPI_FINAL else out


def convlstm_cell(x, state, weight, bias, forget_bias=1.0):
    """ConvLSTMCell.forward (reference RDN.py:50-95).  Returns (h', [c', h'])."""
    _need_cuda(x, weight)
    x = x.contiguous().float()
    n, c, h, w = x.shape
    assert c == 3 and tuple(weight.shape) == (12, 6, 3, 3)
    cn = torch.empty_like(x)
    hn = torch.empty_like(x)
    cp = state[0].contiguous().float() if state is not None else None
    hp = state[1].contiguous().float() if state is not None else None
    L.check(L.lib().binhip_convlstm_fwd(_ptr(x), _ptr(cp), _ptr(hp), _ptr(weight.detach().contiguous().float()),
                                        _ptr(bias.detach().contiguous().float()), float(forget_bias), n, h, w,
                                        _ptr(cn), _ptr(hn), _stream()), "convlstm_fwd")
    return hn, [cn, hn]


def pixel_loss(kind, x, y, eps=1e-6):
    """One of bin_model's pixel criteria (L.LOSS_*), forward only: Charbonnier mean, L1 sum, L2 sum."""
    _need_cuda(x, y)
    x = x.contiguous().float()
    y = y.contiguous().float()
    lib = L.lib()
    part = torch.empty(lib.binhip_charbonnier_partials(x.numel()), dtype=torch.float32, device=x.device)
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    with on_device(x):
        L.check(lib.binhip_pixel_loss_fwd(kind, _ptr(x), _ptr(y), x.numel(), float(eps), _ptr(part), _ptr(loss), _stream()),
                "pixel_loss_fwd")
    return loss


def pixel_loss_grad(kind, x, y, gloss, eps=1e-6):
    x = x.contiguous().float()
    y = y.contiguous().float()
    gx = torch.empty_like(x)
    g = gloss.reshape(1).contiguous().float()
    with on_device(x):
        L.check(L.lib().binhip_pixel_loss_bwd(kind, _ptr(x), _ptr(y), x.numel(), float(eps), _ptr(g), _ptr(gx),
                                              C.c_void_p(0), _stream()), "pixel_loss_bwd")
    return gx


def multi_pixel_loss(kind, pairs, eps=1e-6):
    """All terms of bin_model.get_loss in two launches (binhip_multi_loss_fwd): `pairs` = [(x, y)] contiguous fp32 device tensors
    of one size.  Returns (loss = sum(terms) / T as Python's sum(list) / len(list) computes it, terms [T])."""
    xs = [p[0] for p in pairs] + [p[1] for p in pairs]
    _need_cuda(*xs)
    n = pairs[0][0].numel()
    T = len(pairs)
    if T > L.LOSS_MAX_TERMS or any(t.numel() != n or t.dtype != torch.float32 or not t.is_contiguous() for t in xs):
        raise ValueError("multi_pixel_loss: up to %d pairs of contiguous fp32 tensors of one size" % L.LOSS_MAX_TERMS)
    lib = L.lib()
    dev = pairs[0][0].device
    t = L.BinLossTerms()
    t.n_terms = T
    for i, (x, y) in enumerate(pairs):
        t.x[i], t.y[i] = x.data_ptr(), y.data_ptr()
    part = torch.empty(T * lib.binhip_charbonnier_partials(n), dtype=torch.float32, device=dev)
    terms = torch.empty(T, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    with on_device(pairs[0][0]):
        L.check(lib.binhip_multi_loss_fwd(kind, C.byref(t), n, float(eps), _ptr(part), _ptr(terms), _ptr(loss), _stream()),
                "multi_loss_fwd")
    return loss, terms


def multi_pixel_loss_grad(kind, pairs, gloss, wanted, eps=1e-6):
    """Gradients of that loss in one launch (binhip_multi_loss_bwd).  `wanted`: [(like, [(term, sign), ...])] — one entry per
    tensor that needs a gradient, with the (at most two) terms it appears in; returns the gradient tensors in that order."""
    n = pairs[0][0].numel()
    t = L.BinLossTerms()
    t.n_terms = len(pairs)
    for i, (x, y) in enumerate(pairs):
        t.x[i], t.y[i] = x.data_ptr(), y.data_ptr()
    g = L.BinLossGrads()
    g.n_out = len(wanted)
    outs = []
    for k, (like, where) in enumerate(wanted):
        if not 1 <= len(where) <= 2:
            raise ValueError("multi_pixel_loss_grad: a tensor may appear in one or two terms")
        o = torch.empty_like(like)
        outs.append(o)
        g.out[k] = o.data_ptr()
        g.term_a[k], g.sign_a[k] = where[0]
        g.term_b[k], g.sign_b[k] = where[1] if len(where) == 2 else (-1, 0.0)
    gl = gloss.reshape(1).contiguous().float()
    with on_device(pairs[0][0]):
        L.check(L.lib().binhip_multi_loss_bwd(kind, C.byref(t), n, float(eps), _ptr(gl), C.byref(g), _stream()), "multi_loss_bwd")
    return outs


def charbonnier(x, y, eps=1e-6):
    """mean(sqrt((x-y)^2 + eps)) (reference loss.py:137-141), forward only."""
    return pixel_loss(L.LOSS_CHARBONNIER, x, y, eps)


def charbonnier_grad(x, y, gloss, eps=1e-6):
    return pixel_loss_grad(L.LOSS_CHARBONNIER, x, y, gloss, eps)


# --------------------------------------------------------------------------------------------- backward ops
class DgradWeights:
    """Backward-data weights of one convolution (binhip_weights_relayout_dgrad)."""

    def __init__(self, weight, nterms=1, shuffle=False):
        _need_cuda(weight)
        cout, cin, ks, _ = weight.shape
        lib = L.lib()
        self.ks, self.nterms = ks, nterms
        self.cout = cin                                   # the dgrad conv's outputs = original inputs
        self.cout_pad = lib.binhip_dgrad_rows_pad(ks, cin)
        self.cin_chunks = chunks(cout)
        cb = lib.binhip_conv_cout_block(ks, self.cout_pad, nterms)
        nbytes = lib.binhip_weights_bytes(self.cout_pad, self.cin_chunks, ks)
        dev = weight.device
        self.w_hi = torch.empty(nbytes // 2, dtype=torch.float16, device=dev)
        self.w_lo = torch.empty(nbytes // 2, dtype=torch.float16, device=dev) if nterms == 3 else None
        self.bias = torch.empty(self.cout_pad, dtype=torch.float32, device=dev)
        w = weight.detach().contiguous().float()
        L.check(lib.binhip_weights_relayout_dgrad(_ptr(w), cout, cin, ks, self.cout_pad, self.cin_chunks, cb,
                                                  1 if shuffle else 0, _ptr(self.w_hi), _ptr(self.w_lo),
                                                  _ptr(self.bias), _stream()), "weights_relayout_dgrad")


class RdbGatherWeights:
    """Backward-data weights of ONE concat group of a residual dense block in gather form
    (binhip_weights_relayout_rdb_gather, include/binhip.h): group g = 0 produces the gradient of the block's 96 input
    channels, g = 1..3 that of conv g-1's 32 outputs, each as one forward-shaped 3x3 conv over the stacked output
    gradients of convs g..3.  `weights4`: the block's four OIHW fp32 conv weights."""

    def __init__(self, weights4, group, nterms=1):
        _need_cuda(*weights4)
        lib = L.lib()
        self.ks, self.nterms = 3, nterms
        self.cout = 96 if group == 0 else 32
        self.cout_pad = self.cout
        self.cin_chunks = 2 * (4 - group)
        cb = lib.binhip_conv_cout_block(3, self.cout_pad, nterms)
        nbytes = lib.binhip_weights_bytes(self.cout_pad, self.cin_chunks, 3)
        dev = weights4[0].device
        self.w_hi = torch.empty(nbytes // 2, dtype=torch.float16, device=dev)
        self.w_lo = torch.empty(nbytes // 2, dtype=torch.float16, device=dev) if nterms == 3 else None
        self.bias = torch.empty(self.cout_pad, dtype=torch.float32, device=dev)
        self._src = [w.detach().contiguous().float() for w in weights4]
        arr = (C.c_void_p * 4)(*[w.data_ptr() for w in self._src])
        L.check(lib.binhip_weights_relayout_rdb_gather(arr, group, cb, _ptr(self.w_hi), _ptr(self.w_lo), _ptr(self.bias),
                                                       _stream()), "weights_relayout_rdb_gather")


def rdb_tail(blk, cw3, cwl, out=None, store_o3=False):
    """Fused tail of a residual dense block (binhip_rdb_tail_fwd): o3 = relu(conv3x3(blk[0:192])), y = LFF(cat(blk[0:192],
    o3)) + blk[0:96].  blk: 14-chunk CP (planes 12, 13 receive o3 when store_o3); returns the 6-chunk output CP."""
    _, n, h, w, _ = blk.hi.shape
    nt = cw3.nterms
    if out is None:
        out = CP.empty(6, n, h, w, nt, blk.hi.device, 96)
    L.check(L.lib().binhip_rdb_tail_fwd(n, h, w, nt, _ptr(blk.hi), _ptr(blk.lo), _ptr(cw3.w_hi), _ptr(cw3.w_lo),
                                        _ptr(cw3.bias), _ptr(cwl.w_hi), _ptr(cwl.w_lo), _ptr(cwl.bias), _ptr(out.hi),
                                        _ptr(out.lo), 1 if store_o3 else 0, _ptr(status_word(blk.hi.device)), _stream()),
            "rdb_tail_fwd")
    return out


def conv2d_bwd_data(gy, dw, res=None, res_chunks=0, acc=None, mask=None, mask_from=0, out=None, y_unshuf=0):
    """gx = [mask](conv_{W'}(gy) [+ res] [+ acc]) on chunk planes (binhip_conv2d_bwd_data).  `y_unshuf` > 0: the result leaves
    through an inverse PixelShuffle(2) — 4 * y_unshuf planes at half resolution (BinConvDesc.reserved)."""
    _, n, h, w, _ = gy.hi.shape
    d = L.BinConvDesc()
    d.N, d.H, d.W, d.ksize = n, h, w, dw.ks
    d.cin_chunks, d.cout, d.cout_pad, d.nterms = dw.cin_chunks, dw.cout, dw.cout_pad, dw.nterms
    d.epilogue, d.relu, d.x_cpg, d.x_group_stride, d.n_images = L.EPI_PLANES, 0, 0, 0, 0
    d.status = status_word(gy.hi.device).data_ptr()
    d.reserved = int(y_unshuf)
    if out is None:
        out = (CP.empty(4 * y_unshuf, n, h // 2, w // 2, dw.nterms, gy.hi.device) if y_unshuf
               else CP.empty(chunks(dw.cout), n, h, w, dw.nterms, gy.hi.device, dw.cout))
    z = C.c_void_p(0)
    rc = L.lib().binhip_conv2d_bwd_data(
        C.byref(d), _ptr(gy.hi), _ptr(gy.lo), _ptr(dw.w_hi), _ptr(dw.w_lo), _ptr(dw.bias),
        _ptr(res.hi) if res is not None else z, _ptr(res.lo) if res is not None else z, res_chunks,
        _ptr(acc.hi) if acc is not None else z, _ptr(acc.lo) if acc is not None else z,
        _ptr(mask.hi) if mask is not None else z, mask_from, 0, 0, _ptr(out.hi), _ptr(out.lo), _stream())
    L.check(rc, "conv2d_bwd_data")
    return out


def conv2d_bwd_weight(x, gy, cout, cin, ks, nterms, inv_scale=None, shuffle=False):
    """(dW [cout,cin,ks,ks], db [cout]) fp32 from saved input planes x and gradient planes gy."""
    _, n, h, w, _ = x.hi.shape
    lib = L.lib()
    d = L.BinConvDesc()
    d.N, d.H, d.W, d.ksize = n, h, w, ks
    d.cin_chunks, d.cout, d.cout_pad, d.nterms = chunks(cin), cout, 0, nterms
    d.epilogue, d.relu, d.x_cpg, d.x_group_stride, d.n_images = 0, 0, 0, 0, 0
    dev = x.hi.device
    ws = torch.empty(lib.binhip_wgrad_workspace_bytes(ks, n, h, w, chunks(cin), cout), dtype=torch.uint8, device=dev)
    dw = torch.empty((cout, cin, ks, ks), dtype=torch.float32, device=dev)
    db = torch.empty((cout,), dtype=torch.float32, device=dev)
    L.check(lib.binhip_conv2d_bwd_weight(C.byref(d), _ptr(x.hi), _ptr(x.lo), _ptr(gy.hi), _ptr(gy.lo),
                                         _ptr(inv_scale), _ptr(ws), ws.numel(), _ptr(dw), _ptr(db), cin,
                                         1 if shuffle else 0, 0, _stream()), "conv2d_bwd_weight")
    return dw, db


def grad_planes(g, nterms):
    """fp32 NCHW gradient -> chunk planes stored times a power-of-two scale that maps max|g| into (8, 16] (fp16 range), plus
    the device pair [scale, 1 / scale] (binhip_grad_scale + binhip_nchw_to_planes_scaled)."""
    _need_cuda(g)
    g = g.contiguous().float()
    n, c, h, w = g.shape
    lib = L.lib()
    part = torch.empty(lib.binhip_charbonnier_partials(g.numel()), dtype=torch.float32, device=g.device)
    sc = torch.empty(2, dtype=torch.float32, device=g.device)
    y = CP.empty(chunks(c), n, h, w, nterms, g.device, c)
    with on_device(g):
        L.check(lib.binhip_grad_scale(_ptr(g), g.numel(), 16.0, _ptr(part), _ptr(sc), _stream()), "grad_scale")
        L.check(lib.binhip_nchw_to_planes_scaled(_ptr(g), n, c, h, w, _ptr(sc), _ptr(y.hi), _ptr(y.lo),
                                                 _ptr(status_word(g.device)), _stream()), "nchw_to_planes_scaled")
    return y, sc


def lstm_gates(gates, c_prev, forget_bias, hidden):
    """(c', h') from the gates conv output [N, 4*hidden, H, W] of a ConvLSTM cell of any size (reference RDN.py:74-82)."""
    _need_cuda(gates)
    gates = gates.contiguous().float()
    n, c4, h, w = gates.shape
    assert c4 == 4 * hidden
    cn = torch.empty((n, hidden, h, w), dtype=torch.float32, device=gates.device)
    hn = torch.empty_like(cn)
    cp = c_prev.contiguous().float() if c_prev is not None else None
    with on_device(gates):
        L.check(L.lib().binhip_lstm_gates_fwd(_ptr(gates), _ptr(cp), float(forget_bias), n, hidden, h, w, _ptr(cn), _ptr(hn),
                                              _stream()), "lstm_gates_fwd")
    return cn, hn


def lstm_gates_grad(gates, c_prev, g_h, g_c, forget_bias, hidden, need_cprev):
    gates = gates.contiguous()                      # the kernel indexes plain NCHW
    n, _, h, w = gates.shape
    dg = torch.empty(gates.shape, dtype=gates.dtype, device=gates.device)
    gcp = torch.empty((n, hidden, h, w), dtype=torch.float32, device=gates.device) if need_cprev else None
    cp = c_prev.contiguous().float() if c_prev is not None else None
    gh = g_h.contiguous().float() if g_h is not None else None
    gc = g_c.contiguous().float() if g_c is not None else None
    with on_device(gates):
        L.check(L.lib().binhip_lstm_gates_bwd(_ptr(gates), _ptr(cp), _ptr(gh), _ptr(gc), float(forget_bias), n, hidden, h, w,
                                              _ptr(dg), _ptr(gcp), _stream()), "lstm_gates_bwd")
    return dg, gcp


# --------------------------------------------------------------------------------------------- harness glue (N1)
def u8_to_frame(img_u8, pads):
    """HWC BGR uint8 device tensor -> padded fp32 [1,3,Hp,Wp] RGB frame (read_image + ReplicationPad2d)."""
    _need_cuda(img_u8)
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.shape[2] == 3
    img_u8 = img_u8.contiguous()
    h, w, _ = img_u8.shape
    l, r, t, b = pads
    out = torch.empty((1, 3, h + t + b, w + l + r), dtype=torch.float32, device=img_u8.device)
    L.check(L.lib().binhip_u8_to_frame(_ptr(img_u8), h, w, l, r, t, b, _ptr(out), _stream()), "u8_to_frame")
    return out


def frame_to_u8(frame, top, left, h, w):
    """fp32 [1,3,Hp,Wp] (or [3,Hp,Wp]) RGB -> cropped HWC BGR uint8 (tensor2img + crop)."""
    _need_cuda(frame)
    f = frame.reshape(3, frame.shape[-2], frame.shape[-1]).contiguous().float()
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=f.device)
    L.check(L.lib().binhip_frame_to_u8(_ptr(f), f.shape[1], f.shape[2], top, left, h, w, _ptr(out), _stream()), "frame_to_u8")
    return out


def pixel_unshuffle(x, r=2):
    """Exact space-to-depth: out[b, c*r*r + i*r + j, y, x] = in[b, c, y*r+i, x*r+j] (reference RDN.py:107-132)."""
    _need_cuda(x)
    x = x.contiguous().float()
    n, c, h, w = x.shape
    if h % r or w % r or r < 1:
        raise RuntimeError(f"bin_amd: pixel_unshuffle needs H, W divisible by r (got {h}x{w}, r={r})")
    y = torch.empty((n, c * r * r, h // r, w // r), dtype=torch.float32, device=x.device)
    L.check(L.lib().binhip_pixel_unshuffle_f32(_ptr(x), n, c, h, w, r, _ptr(y), _stream()), "pixel_unshuffle")
    return y
