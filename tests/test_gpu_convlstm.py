"""-m gpu: ConvLSTMCell (reference RDN.py:9-95) — fused forward/backward kernels, the four-pixel variants, cells of other sizes on the general path."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden

pytestmark = pytest.mark.gpu


def test_convlstm_golden(canon_gpu):
    from bin_amd import ops
    g = load_golden("g2_convlstm")
    w, b = canon_gpu["clstm_6_prime.Gates.weight"], canon_gpu["clstm_6_prime.Gates.bias"]
    h1, st1 = ops.convlstm_cell(torch.from_numpy(g["x1"]).cuda(), None, w, b)
    h2, st2 = ops.convlstm_cell(torch.from_numpy(g["x2"]).cuda(), st1, w, b)
    for got, key in ((h1, "h1"), (st1[0], "c1"), (h2, "h2"), (st2[0], "c2")):
        assert float((got.cpu() - torch.from_numpy(g[key])).abs().max()) <= 2e-6, key


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def test_convlstm_backward_vs_autograd(canon_cpu, canon_gpu):
    from bin_amd.autograd import convlstm_apply
    from oracle import rdn_oracle as O
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 18, 70, generator=gen)
    c0, h0 = torch.rand(2, 3, 18, 70, generator=gen), torch.rand(2, 3, 18, 70, generator=gen)
    for with_state in (False, True):
        xr = x.clone().requires_grad_(True)
        w = canon_cpu["clstm_6_prime.Gates.weight"].clone().requires_grad_(True)
        b = (canon_cpu["clstm_6_prime.Gates.bias"] + 0.1).clone().requires_grad_(True)
        st = [c0.clone().requires_grad_(True), h0.clone().requires_grad_(True)] if with_state else None
        h, (c, _) = O.convlstm_cell(xr, st, w, b)
        gh, gc = torch.rand_like(h), torch.rand_like(c)
        (h * gh).sum().backward(retain_graph=True) if False else ((h * gh).sum() + (c * gc).sum()).backward()
        xg = x.cuda().requires_grad_(True)
        wg = w.detach().cuda().requires_grad_(True)
        bg = b.detach().cuda().requires_grad_(True)
        stg = [c0.cuda().requires_grad_(True), h0.cuda().requires_grad_(True)] if with_state else None
        hh, (cc, _) = convlstm_apply(xg, stg, wg, bg, 1.0)
        ((hh * gh.cuda()).sum() + (cc * gc.cuda()).sum()).backward()
        assert _rel(hh.detach().cpu(), h.detach()) <= 1e-5
        assert _rel(xg.grad.cpu(), xr.grad) <= 2e-5
        assert _rel(wg.grad.cpu(), w.grad) <= 2e-4
        assert _rel(bg.grad.cpu(), b.grad) <= 2e-4
        if with_state:
            assert _rel(stg[0].grad.cpu(), st[0].grad) <= 2e-5
            assert _rel(stg[1].grad.cpu(), st[1].grad) <= 2e-5


REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tag", ["lstm_5_7_k3_state", "lstm_3_16_k5_nostate", "lstm_20_4_k1_state"])
def test_convlstm_cells_of_other_sizes(tag):
    """ConvLSTMCell(input_size, hidden_size, kernel_size) other than bin_stage4's (3, 3, 3x3) (reference RDN.py:14-24): the
    gates convolution on the general conv / weight-gradient / backward-data kernels + the elementwise gate kernels, against
    the REFERENCE cell's outputs and autograd gradients (fixture g10_rdn_shapes)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from conftest import load_golden
    from shape_cases import LSTM_CASES
    from bin_amd.models.archs import RDN as A
    from bin_amd.weights import general_lstm_weights
    a, b, ks, n, h, w, with_state = LSTM_CASES[tag]
    g = load_golden("g10_rdn_shapes")
    T = lambda k: torch.from_numpy(g[f"{tag}.{k}"])
    Wg, Bg = (torch.from_numpy(v) for v in general_lstm_weights(0, a, b, ks))
    cell = A.ConvLSTMCell(a, b, kernel_size=ks, padding=ks // 2)
    cell.load_state_dict({"Gates.weight": Wg, "Gates.bias": Bg}, strict=True)
    cell = cell.cuda()
    x = T("x").cuda().requires_grad_(True)
    state = [T("c0").cuda().requires_grad_(True), T("h0").cuda().requires_grad_(True)] if with_state else None
    with torch.no_grad():                                   # inference path
        h_inf, (c_inf, _) = cell(x.detach(), [t.detach() for t in state] if state else None)
    h1, (c1, h1b) = cell(x, state)
    assert h1b is h1
    for got in ((h1, c1), (h_inf, c_inf)):
        assert float((got[0].detach().cpu() - T("h")).abs().max()) <= 2e-6
        assert float((got[1].detach().cpu() - T("c")).abs().max()) <= 2e-6
    ((h1 * T("gh").cuda()).sum() + (c1 * T("gc").cuda()).sum()).backward()
    rel = lambda u, v: float((u - v).abs().max() / v.abs().max().clamp_min(1e-12))
    assert rel(x.grad.cpu(), T("gx")) <= 3e-5
    assert rel(cell.Gates.weight.grad.cpu(), T("dw")) <= 3e-5
    assert rel(cell.Gates.bias.grad.cpu(), T("db")) <= 3e-5
    if with_state:
        assert rel(state[0].grad.cpu(), T("gc0")) <= 3e-5 and rel(state[1].grad.cpu(), T("gh0")) <= 3e-5
    with pytest.raises(NotImplementedError):
        A.ConvLSTMCell(3, 3, kernel_size=7, padding=3)


def test_general_convlstm_conv_refuses_a_weight_changed_before_backward():
    from bin_amd.autograd import _ConvFn
    x = torch.rand(1, 8, 16, 16, device="cuda", requires_grad=True)
    w = (torch.rand(12, 8, 3, 3, device="cuda") - 0.5).requires_grad_()
    b = torch.zeros(12, device="cuda", requires_grad=True)
    y = _ConvFn.apply(x, w, b)
    y.sum().backward()                                             # untouched weight: fine
    assert w.grad is not None and torch.isfinite(w.grad).all()
    y = _ConvFn.apply(x, w, b)
    with torch.no_grad():
        w.mul_(0.5)                                                # what an optimizer step does
    with pytest.raises(RuntimeError, match="modified in place"):
        y.sum().backward()


def _off1(t):
    """A copy of `t` whose data pointer is 4 bytes past a 16-byte boundary (forces the one-pixel ConvLSTM kernels)."""
    buf = torch.empty(t.numel() + 4, dtype=t.dtype, device=t.device)
    assert buf.data_ptr() % 16 == 0
    v = buf[1:1 + t.numel()].view(t.shape)
    v.copy_(t)
    assert v.data_ptr() % 16 == 4
    return v


@pytest.mark.parametrize("with_state", [False, True])
@pytest.mark.parametrize("shape", [(1, 16, 24), (2, 9, 20), (1, 33, 4), (1, 5, 64)])
def test_convlstm_four_pixel_kernels_equal_the_one_pixel_kernels(shape, with_state):
    """Round 5: `binhip_convlstm_fwd` / `_bwd` run four pixels per thread (float4 rows, weights as ds_read_b128, every epilogue
    load before the first store) when W % 4 == 0 and the planes are 16-byte aligned, and the round-1 one-pixel kernels otherwise.
    Same fmaf chains per pixel -> the two must agree BIT FOR BIT: the same data is run through both by mis-aligning the planes
    by one float.  (Both are pinned to the reference by test_convlstm_golden above.)"""
    import ctypes as C
    from bin_amd import _lib as L
    lib = L.lib()
    n, h, w = shape
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + w)
    dev = torch.device("cuda")
    mk = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(dev)
    x, cp, hp, gh, gc = (mk(n, 3, h, w) for _ in range(5))
    wt, b = mk(12, 6, 3, 3), mk(12)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nbytes = lib.binhip_convlstm_bwd_workspace_bytes(n, h, w)

    def run(conv):
        X, CP, HP, GH, GC = (conv(t) for t in (x, cp, hp, gh, gc))
        if not with_state:
            CP = HP = None
        cn, hn, gx, ghp, gcp = (conv(torch.zeros_like(x)) for _ in range(5))
        L.check(lib.binhip_convlstm_fwd(p(X), p(CP), p(HP), p(wt), p(b), 1.0, n, h, w, p(cn), p(hn), stream), "fwd")
        ws = torch.empty(nbytes + 512, dtype=torch.uint8, device=dev)
        dw, db = torch.zeros_like(wt), torch.zeros_like(b)
        L.check(lib.binhip_convlstm_bwd(p(X), p(CP), p(HP), p(wt), p(b), 1.0, n, h, w, p(GH), p(GC), p(ws), nbytes, p(gx),
                                        p(ghp) if with_state else None, p(gcp) if with_state else None, p(dw), p(db), stream), "bwd")
        torch.cuda.synchronize()
        return [t.clone() for t in (cn, hn, gx, dw, db)] + ([ghp.clone(), gcp.clone()] if with_state else [])

    fast = run(lambda t: t.clone())
    slow = run(_off1)
    for i, (a, c) in enumerate(zip(fast, slow)):
        assert torch.equal(a, c), (i, float((a - c).abs().max()))
    # and against plain torch (the formula of RDN.py:74-92), forward only: 1e-6
    xin = torch.cat((x, hp if with_state else torch.zeros_like(x)), 1)
    gates = torch.nn.functional.conv2d(xin, wt, b, padding=1)
    i_, j_, f_, o_ = gates.chunk(4, 1)
    c_ref = (cp if with_state else 0) * torch.sigmoid(f_ + 1.0) + torch.sigmoid(i_) * torch.tanh(j_)
    h_ref = torch.tanh(c_ref) * torch.sigmoid(o_)
    assert float((fast[0] - c_ref).abs().max()) <= 2e-6 and float((fast[1] - h_ref).abs().max()) <= 2e-6
