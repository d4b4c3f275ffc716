"""-m gpu: per-op parity of the HIP kernels (through the C ABI) against the golden fixtures produced
by the reference itself (tests/golden/make_golden.py) and against the oracle on seeded inputs.
Tolerances: nterms=3 (fp16 hi/lo split, fp32-class) 2e-5 relative-to-scale; nterms=1 (fp16 inputs,
fp32 accumulate) 2e-3 relative-to-scale per op (the whole-net bar of 1e-3 is tested in test_gpu_net)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

CONVS = {
    "k2_sfe1_24": ("model1.SFENet1", 5), "k2_sfe1_36": ("model2.SFENet1", 5), "k2_sfe1_60": ("model3.SFENet1", 5),
    "k3_sfe2": ("model1.SFENet2", 3),
    "k4_rdbconv0": ("model1.RDBs.0.convs.0.conv.0", 3), "k4_rdbconv1": ("model1.RDBs.0.convs.1.conv.0", 3),
    "k4_rdbconv2": ("model1.RDBs.0.convs.2.conv.0", 3), "k4_rdbconv3": ("model1.RDBs.0.convs.3.conv.0", 3),
    "k5_lff": ("model1.RDBs.0.LFF", 1), "k6_gff0": ("model1.GFF.0", 1),
}
TOL = {1: 2e-3, 3: 2e-5}


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


@pytest.mark.parametrize("nterms", [3, 1])
def test_layout_roundtrip(nterms):
    from bin_amd import ops
    x = torch.randn(2, 37, 9, 13, device="cuda")
    cp = ops.nchw_to_planes(x, nterms)
    assert cp.hi.shape == (3, 2, 9, 13, 16)
    y = ops.planes_to_nchw(cp)
    tol = 1e-6 if nterms == 3 else 1e-3
    assert float((x - y).abs().max()) <= tol * float(x.abs().max())


def test_pixel_reshuffle_function_exact():
    """The module-level pixel_reshuffle (API parity with reference RDN.py:107-132) is an exact permutation."""
    from bin_amd.models.archs.RDN import pixel_reshuffle
    g = load_golden("g1_pixel_reshuffle")
    assert torch.equal(pixel_reshuffle(torch.from_numpy(g["x"]).cuda(), 2).cpu(), torch.from_numpy(g["y"]))
    x = torch.randn(2, 5, 12, 18)
    assert torch.equal(pixel_reshuffle(x.cuda(), 3).cpu(), torch.nn.functional.pixel_unshuffle(x, 3))


@pytest.mark.parametrize("nterms", [3, 1])
def test_pixel_reshuffle_pack(nterms):
    """K1 against the reference's own pixel_reshuffle output (g1_pixel_reshuffle) — 6 = 2 frames x 3."""
    from bin_amd import ops
    g = load_golden("g1_pixel_reshuffle")
    x = torch.from_numpy(g["x"]).cuda()
    y = ops.planes_to_nchw(ops.pack_inputs([x[:, :3], x[:, 3:]], nterms), 24)
    ref = torch.from_numpy(g["y"]).cuda()
    assert float((y - ref).abs().max()) <= (1e-6 if nterms == 3 else 2e-3) * float(ref.abs().max())


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("key", sorted(CONVS))
def test_conv_forward_golden(key, nterms, canon_gpu):
    from bin_amd import ops
    g = load_golden("g1_convs")
    wname, ks = CONVS[key]
    x = torch.from_numpy(g[key + ".x"]).cuda()
    ref = torch.from_numpy(g[key + ".y"]).cuda()
    cw = ops.ConvWeights(canon_gpu[wname + ".weight"], canon_gpu[wname + ".bias"], nterms=nterms)
    y = ops.planes_to_nchw(ops.conv2d(ops.nchw_to_planes(x, nterms), cw), cw.cout)
    assert y.shape == ref.shape
    assert _rel(y, ref) <= TOL[nterms], (key, _rel(y, ref))


@pytest.mark.parametrize("nterms", [3, 1])
def test_conv_shuffle_golden(nterms, canon_gpu):
    """K8: conv 96->256 + PixelShuffle(2) fused store vs F.pixel_shuffle of the reference conv output."""
    from bin_amd import ops
    g = load_golden("g1_convs")
    x = torch.from_numpy(g["k8_up0.x"]).cuda()
    ref = torch.nn.functional.pixel_shuffle(torch.from_numpy(g["k8_up0.y"]).cuda(), 2)
    cw = ops.ConvWeights(canon_gpu["model1.UPNet.0.weight"], canon_gpu["model1.UPNet.0.bias"], nterms=nterms,
                         shuffle=True)
    from bin_amd import _lib as L
    y = ops.planes_to_nchw(ops.conv2d(ops.nchw_to_planes(x, nterms), cw, epilogue=L.EPI_SHUFFLE), 64)
    assert y.shape == ref.shape
    assert _rel(y, ref) <= TOL[nterms]


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("nimg", [0, 2, 3, 5])
def test_conv_final_golden(nterms, nimg, canon_gpu):
    """K9: conv 64->3 + mean(inputs) -> fp32 NCHW."""
    from bin_amd import ops, _lib as L
    g = load_golden("g1_convs")
    x = torch.from_numpy(g["k9_up2.x"]).cuda()
    ref = torch.from_numpy(g["k9_up2.y"]).cuda()
    imgs = [torch.rand_like(ref) for _ in range(nimg)]
    if nimg:
        s = imgs[0]
        for t in imgs[1:]:
            s = s + t
        ref = ref + s / nimg
    cw = ops.ConvWeights(canon_gpu["model1.UPNet.2.weight"], canon_gpu["model1.UPNet.2.bias"], nterms=nterms)
    y = ops.conv2d(ops.nchw_to_planes(x, nterms), cw, epilogue=L.EPI_FINAL, images=imgs)
    assert y.shape == ref.shape
    assert _rel(y, ref) <= TOL[nterms]


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 7, 5), (2, 33, 65), (1, 40, 100), (3, 17, 31)])
@pytest.mark.parametrize("ks,cin,cout,relu,res", [(3, 96, 32, True, False), (3, 96, 96, False, True),
                                                  (1, 224, 96, False, True), (5, 36, 96, False, False),
                                                  (3, 64, 64, True, True)])
def test_conv_vs_oracle_ragged(nterms, shape, ks, cin, cout, relu, res):
    """Ragged / tiny / multi-batch shapes (partial tiles on every side) vs plain F.conv2d in fp64."""
    from bin_amd import ops
    n, h, w = shape
    gen = torch.Generator(device="cpu").manual_seed(h * 1000 + w + ks)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, ks, ks, generator=gen) / (cin * ks * ks) ** 0.5
    b = torch.randn(cout, generator=gen)
    r = torch.randn(n, cout, h, w, generator=gen) if res else None
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), padding=ks // 2)
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.relu()
    cw = ops.ConvWeights(wt.cuda(), b.cuda(), nterms=nterms)
    y = ops.conv2d(ops.nchw_to_planes(x.cuda(), nterms), cw, relu=relu,
                   residual=ops.nchw_to_planes(r.cuda(), nterms) if res else None)
    y = ops.planes_to_nchw(y, cout).cpu().double()
    assert _rel(y, ref) <= TOL[nterms]


@pytest.mark.parametrize("nterms", [3, 1])
def test_resblock_nobn_golden(nterms):
    """ResidualBlock_noBN(64) (dead in the reference) through the same conv kernels."""
    from bin_amd import ops
    g = load_golden("g5_resblock")
    x = torch.from_numpy(g["x"]).cuda()
    cw1 = ops.ConvWeights(torch.from_numpy(g["w1"]).cuda(), torch.from_numpy(g["b1"]).cuda(), nterms=nterms)
    cw2 = ops.ConvWeights(torch.from_numpy(g["w2"]).cuda(), torch.from_numpy(g["b2"]).cuda(), nterms=nterms)
    xp = ops.nchw_to_planes(x, nterms)
    y = ops.conv2d(ops.conv2d(xp, cw1, relu=True), cw2, residual=xp)
    ref = torch.from_numpy(g["y"]).cuda()
    assert _rel(ops.planes_to_nchw(y, 64), ref) <= TOL[nterms]


def test_convlstm_golden(canon_gpu):
    from bin_amd import ops
    g = load_golden("g2_convlstm")
    w, b = canon_gpu["clstm_6_prime.Gates.weight"], canon_gpu["clstm_6_prime.Gates.bias"]
    h1, st1 = ops.convlstm_cell(torch.from_numpy(g["x1"]).cuda(), None, w, b)
    h2, st2 = ops.convlstm_cell(torch.from_numpy(g["x2"]).cuda(), st1, w, b)
    for got, key in ((h1, "h1"), (st1[0], "c1"), (h2, "h2"), (st2[0], "c2")):
        assert float((got.cpu() - torch.from_numpy(g[key])).abs().max()) <= 2e-6, key


def test_charbonnier_golden():
    from bin_amd import ops
    g = load_golden("g1_charbonnier")
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    loss = ops.charbonnier(x, y)
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    gx = ops.charbonnier_grad(x, y, torch.ones((), device="cuda"))
    assert float((gx.cpu() - torch.from_numpy(g["gx"])).abs().max()) <= 1e-9 + 1e-5 * float(np.abs(g["gx"]).max())


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("set_name,k", [("model1", 2), ("model2", 3), ("model3", 5), ("model4", 5)])
def test_rdn_golden(set_name, k, nterms, canon_gpu):
    """One whole RDN sub-network (66 fused launches from C) vs the reference module's output."""
    from bin_amd.rdn_plan import RdnWeights, rdn_forward
    g = load_golden(f"g2_rdn_{set_name}")
    ins = [torch.from_numpy(g[f"in{i}"]).cuda() for i in range(k)]
    wts = RdnWeights(canon_gpu, k, nterms, prefix=set_name + ".")
    y = rdn_forward(wts, ins)
    err = float((y.cpu() - torch.from_numpy(g["y"])).abs().max())
    assert err <= (2e-5 if nterms == 3 else 1e-3), err
