"""-m gpu: weight-gradient kernels (3x3 X-row, 5x5, 1x1 streaming, the batched reduction) and the conv backward-data/weight fixtures, against the reference's autograd outputs and float64."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


CONVS_BWD = {
    "k2_sfe1_24": ("model1.SFENet1", 5), "k2_sfe1_36": ("model2.SFENet1", 5), "k2_sfe1_60": ("model3.SFENet1", 5),
    "k3_sfe2": ("model1.SFENet2", 3),
    "k4_rdbconv0": ("model1.RDBs.0.convs.0.conv.0", 3), "k4_rdbconv1": ("model1.RDBs.0.convs.1.conv.0", 3),
    "k4_rdbconv2": ("model1.RDBs.0.convs.2.conv.0", 3), "k4_rdbconv3": ("model1.RDBs.0.convs.3.conv.0", 3),
    "k5_lff": ("model1.RDBs.0.LFF", 1), "k6_gff0": ("model1.GFF.0", 1), "k8_up0": ("model1.UPNet.0", 3),
    "k9_up2": ("model1.UPNet.2", 3),
}


TOL_BWD = {1: 3e-3, 3: 3e-5}


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("key", sorted(CONVS_BWD))
def test_conv_dgrad_wgrad_golden(key, nterms, canon_gpu):
    """dX, dW, db of every live conv shape vs the reference autograd (g1_convs)."""
    from bin_amd import ops
    g = load_golden("g1_convs")
    wname, ks = CONVS_BWD[key]
    w = canon_gpu[wname + ".weight"]
    cout, cin = w.shape[0], w.shape[1]
    x = torch.from_numpy(g[key + ".x"]).cuda()
    gy = torch.from_numpy(g[key + ".gy"]).cuda()
    gyp = ops.nchw_to_planes(gy, nterms)
    gx = ops.planes_to_nchw(ops.conv2d_bwd_data(gyp, ops.DgradWeights(w, nterms)), cin)
    assert _rel(gx, torch.from_numpy(g[key + ".gx"]).cuda()) <= TOL_BWD[nterms], "dgrad"
    dw, db = ops.conv2d_bwd_weight(ops.nchw_to_planes(x, nterms), gyp, cout, cin, ks, nterms)
    assert _rel(dw, torch.from_numpy(g[key + ".gw"]).cuda()) <= TOL_BWD[nterms], "wgrad"
    assert _rel(db, torch.from_numpy(g[key + ".gb"]).cuda()) <= TOL_BWD[nterms], "dbias"


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("shape", [(1, 5, 7), (2, 19, 45), (1, 8, 32)])
def test_wgrad_ragged(nterms, shape):
    from bin_amd import ops
    n, h, w = shape
    gen = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn(n, 40, h, w, generator=gen, dtype=torch.float64)
    gy = torch.randn(n, 35, h, w, generator=gen, dtype=torch.float64)
    wt = torch.zeros(35, 40, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(35, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x, wt, b, padding=1).backward(gy)
    dw, db = ops.conv2d_bwd_weight(ops.nchw_to_planes(x.float().cuda(), nterms),
                                   ops.nchw_to_planes(gy.float().cuda(), nterms), 35, 40, 3, nterms)
    assert _rel(dw.cpu().double(), wt.grad) <= TOL_BWD[nterms]
    assert _rel(db.cpu().double(), b.grad) <= TOL_BWD[nterms]


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("cfg", [(2, 33, 70, 192, 32), (1, 16, 32, 96, 96), (3, 17, 40, 224, 35), (1, 40, 33, 16, 32),
                                 (2, 64, 64, 160, 32), (1, 130, 31, 128, 64)])
def test_wgrad_3x3_shapes(nterms, cfg):
    """the 3x3 weight-gradient kernel (eight waves, two LDS stages) over 1-7 channel pairs, 1-3 output tiles, tiles that hang
    over the right / bottom edge, more workgroups than tiles, several images (reference: autograd of F.conv2d(padding=1),
    RDN.py:141,187-207).  The same shapes validated the rolling-row experiment of the tuning build."""
    from bin_amd import ops
    n, h, w, cin, cout = cfg
    gen = torch.Generator().manual_seed(h * 1000 + w + cin)
    x = torch.randn(n, cin, h, w, generator=gen, dtype=torch.float64)
    gy = torch.randn(n, cout, h, w, generator=gen, dtype=torch.float64)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x, wt, b, padding=1).backward(gy)
    xp, gp = ops.nchw_to_planes(x.float().cuda(), nterms), ops.nchw_to_planes(gy.float().cuda(), nterms)
    dw, db = ops.conv2d_bwd_weight(xp, gp, cout, cin, 3, nterms)
    assert _rel(dw.cpu().double(), wt.grad) <= TOL_BWD[nterms]
    assert _rel(db.cpu().double(), b.grad) <= TOL_BWD[nterms]
    dw2, db2 = ops.conv2d_bwd_weight(xp, gp, cout, cin, 3, nterms)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "fixed summation order"


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("cfg", [(1, 5, 7, 40, 35), (2, 19, 45, 224, 96), (1, 8, 32, 16, 96), (1, 33, 70, 600, 64),
                                 (3, 6, 40, 272, 96), (1, 130, 64, 1152, 96)])
def test_wgrad_1x1_ragged(nterms, cfg):
    """the streaming 1x1 kernel: one / two channel pairs per wave, 1-3 workgroup columns, odd chunk counts, strips that hang
    over the right and bottom edges, more workgroups than strips (reference: autograd of F.conv2d, RDN.py:141,162)."""
    from bin_amd import ops
    n, h, w, cin, cout = cfg
    gen = torch.Generator().manual_seed(h * 100 + w + cin)
    x = torch.randn(n, cin, h, w, generator=gen, dtype=torch.float64)
    gy = torch.randn(n, cout, h, w, generator=gen, dtype=torch.float64)
    wt = torch.zeros(cout, cin, 1, 1, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x, wt, b).backward(gy)
    dw, db = ops.conv2d_bwd_weight(ops.nchw_to_planes(x.float().cuda(), nterms),
                                   ops.nchw_to_planes(gy.float().cuda(), nterms), cout, cin, 1, nterms)
    assert _rel(dw.cpu().double(), wt.grad) <= TOL_BWD[nterms]
    assert _rel(db.cpu().double(), b.grad) <= TOL_BWD[nterms]
    dw2, db2 = ops.conv2d_bwd_weight(ops.nchw_to_planes(x.float().cuda(), nterms),
                                     ops.nchw_to_planes(gy.float().cuda(), nterms), cout, cin, 1, nterms)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "fixed summation order"


# ------------------------------------------------------------------------------------------------ live wgrad timing
def test_backward_profiler_times_the_weight_gradient_launches():
    import ctypes
    from bin_amd import _lib as L
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.weights import reference_state_dict, synthetic_frames
    net = bin_stage4_lstm()
    net.load_state_dict(reference_state_dict(0), strict=True)
    net = net.cuda().train()
    frames = [f.cuda() for f in synthetic_frames(3, 1, 64, 64, 6)]
    lib = L.lib()
    handle = ctypes.c_void_p(0)
    L.check(lib.binhip_profiler_create(3, 32, L.PROF_WGRAD, 512, ctypes.byref(handle)), "profiler_create")
    try:
        net.set_profiler(handle, backward=True)
        loss = sum((o * o).mean() for o in net(*frames))
        loss.backward()
        torch.cuda.synchronize()
        net.set_profiler(None, backward=True)
        ms, n = ctypes.c_double(0), ctypes.c_int(0)
        L.check(lib.binhip_profiler_read(handle, ctypes.byref(ms), ctypes.byref(n)), "profiler_read")
        assert n.value == 4 * 12 * 4                      # four RDN calls x 12 dense blocks x 4 convs (3x3, 32 outputs)
        assert 0.0 < ms.value < 1e4
    finally:
        lib.binhip_profiler_destroy(handle)
