"""CPU: the oracle (oracle/rdn_oracle.py) against every golden fixture the reference produced
(tests/golden/make_golden.py imports the reference in the build container and writes them)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from oracle import rdn_oracle as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_pixel_reshuffle():
    g = load_golden("g1_pixel_reshuffle")
    assert torch.equal(O.pixel_reshuffle(T(g["x"]), 2), T(g["y"]))


def test_convs_are_plain_conv2d(canon_cpu):
    g = load_golden("g1_convs")
    names = {"k3_sfe2": ("model1.SFENet2", 1), "k5_lff": ("model1.RDBs.0.LFF", 0), "k2_sfe1_36": ("model2.SFENet1", 2)}
    for key, (wn, pad) in names.items():
        y = F.conv2d(T(g[key + ".x"]), canon_cpu[wn + ".weight"], canon_cpu[wn + ".bias"], padding=pad)
        assert float((y - T(g[key + ".y"])).abs().max()) <= 1e-5


def test_charbonnier():
    g = load_golden("g1_charbonnier")
    x = T(g["x"]).requires_grad_(True)
    l = O.charbonnier(x, T(g["y"]))
    l.backward()
    assert abs(float(l) - float(g["loss"])) <= 1e-7
    assert float((x.grad - T(g["gx"])).abs().max()) <= 1e-9


def test_rdb(canon_cpu):
    g = load_golden("g2_rdb")
    y = O.rdb(T(g["x"]), canon_cpu, "model1.RDBs.0")
    assert float((y - T(g["y"])).abs().max()) <= 1e-5


@pytest.mark.parametrize("set_name,k", [("model1", 2), ("model2", 3), ("model3", 5), ("model4", 5)])
def test_rdn(set_name, k, canon_cpu):
    g = load_golden(f"g2_rdn_{set_name}")
    y = O.rdn([T(g[f"in{i}"]) for i in range(k)], canon_cpu, set_name)
    assert float((y - T(g["y"])).abs().max()) <= 1e-5


def test_convlstm(canon_cpu):
    g = load_golden("g2_convlstm")
    w, b = canon_cpu["clstm_6_prime.Gates.weight"], canon_cpu["clstm_6_prime.Gates.bias"]
    h1, st1 = O.convlstm_cell(T(g["x1"]), None, w, b)
    h2, st2 = O.convlstm_cell(T(g["x2"]), st1, w, b)
    for got, key in ((h1, "h1"), (st1[0], "c1"), (h2, "h2"), (st2[0], "c2")):
        assert float((got - T(g[key])).abs().max()) <= 1e-6


def test_resblock():
    g = load_golden("g5_resblock")
    y = O.residual_block_nobn(T(g["x"]), T(g["w1"]), T(g["b1"]), T(g["w2"]), T(g["b2"]))
    assert float((y - T(g["y"])).abs().max()) <= 1e-6


@pytest.mark.parametrize("tag", ["a", "b"])
def test_whole_net(tag, canon_cpu):
    from bin_amd.weights import synthetic_frames
    g = load_golden(f"g3_net_{tag}")
    n, _, h, w = [int(v) for v in g["shape"]]
    with torch.no_grad():
        out = O.bin_stage4_forward(synthetic_frames(int(g["seed_x"]), n, h, w, 6), canon_cpu)
    ref = T(g["out"])
    assert max(float((o - r).abs().max()) for o, r in zip(out, ref)) <= 1e-5


def test_harness_helpers():
    g = load_golden("g4_harness")
    i1, i2 = O.tensor2img(T(g["t1"])), O.tensor2img(T(g["t2"]))
    assert np.array_equal(i1, g["img1"]) and np.array_equal(i2, g["img2"])
    assert O.calculate_psnr(i1, i2) == float(g["psnr"])
    for key in g.files:
        if key.startswith("pad."):
            h, w = [int(v) for v in key[4:].split("x")]
            assert O.pad_sizes(h, w) == tuple(int(v) for v in g[key])


def test_rdb_backward_data_gather_form_identity(canon_cpu):
    """The algorithm behind binhip_weights_relayout_rdb_gather / binhip_rdn_backward, pinned on the CPU against torch
    autograd of the oracle's dense block (RDN.py:132-165): concat group g (g = 0: the 96 block inputs, g = 1..3: conv
    g-1's outputs) receives L_g (the LFF share, + gy on group 0) plus ONE forward-shaped conv of the stacked masked output
    gradients of convs g..3 with weights Wg[r][32(c-g)+co][dy][dx] = W_c[co][base_g+r][2-dy][2-dx]."""
    import torch.nn.functional as F
    from oracle import rdn_oracle as O
    W = {k: v.double() for k, v in canon_cpu.items() if k.startswith("model1.RDBs.3.")}
    pre = "model1.RDBs.3"
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 96, 10, 14, generator=g, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(1, 96, 10, 14, generator=g, dtype=torch.float64)
    y = O.rdb(x, W, pre)
    (gx_ref,) = torch.autograd.grad(y, x, gy)
    # forward activations of the four convs (for the ReLU masks)
    acts, cat = [], x.detach()
    for c in range(4):
        a = F.relu(F.conv2d(cat, W[f"{pre}.convs.{c}.conv.0.weight"], W[f"{pre}.convs.{c}.conv.0.bias"], padding=1))
        acts.append(a)
        cat = torch.cat((cat, a), 1)
    # L = LFF^T gy on all 224 channels (1x1 dgrad), + gy on the first 96 (the block's `+ x`)
    L = F.conv2d(gy, W[f"{pre}.LFF.weight"].permute(1, 0, 2, 3))
    L[:, :96] += gy

    def gather_weight(group):
        base = 0 if group == 0 else 96 + 32 * (group - 1)
        rows = 96 if group == 0 else 32
        parts = []
        for c in range(group, 4):                                        # convs that read this group
            wc = W[f"{pre}.convs.{c}.conv.0.weight"]                      # [32, 96+32c, 3, 3]
            parts.append(wc[:, base:base + rows].flip(2, 3).permute(1, 0, 2, 3))      # [rows, 32, 3, 3]
        return torch.cat(parts, 1)                                        # [rows, 32*(4-group), 3, 3]

    G = [None] * 4                                                        # masked output gradients of conv 0..3
    G[3] = L[:, 192:224] * (acts[3] > 0)
    for group in (3, 2, 1):                                               # group g holds conv g-1's output
        stacked = torch.cat(G[group:], 1)
        tot = L[:, 96 + 32 * (group - 1):96 + 32 * group] + F.conv2d(stacked, gather_weight(group), padding=1)
        G[group - 1] = tot * (acts[group - 1] > 0)
    gx = L[:, :96] + F.conv2d(torch.cat(G, 1), gather_weight(0), padding=1)
    assert float((gx - gx_ref).abs().max()) <= 1e-12 * max(1.0, float(gx_ref.abs().max()))


def test_f16x3_split_arithmetic_error_model():
    """The numeric model of the `f16x3` precision mode, emulated in numpy: x = hi + lo with hi = fp16(x),
    lo = fp16(x - hi); a dot product is Ah.Bh + Al.Bh + Ah.Bl accumulated in fp32 (the Al.Bl term is dropped).  For
    conv-sized reductions of unit-scale data that is fp32-class (relative error ~1e-6), where a single fp16 product
    (`f16` mode) is ~3e-4 and bf16 would be ~3e-3 — the reason the kernels use fp16 and not bf16 inputs."""
    g = np.random.Generator(np.random.PCG64(0))
    K = 9 * 192                                                         # the longest reduction of an RDB conv
    a = g.standard_normal((64, K)) / np.sqrt(K)
    b = g.uniform(0.0, 1.0, (K, 64))
    exact = a @ b

    def split(x):
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float64)).astype(np.float16)
        return hi.astype(np.float32), lo.astype(np.float32)

    ah, al = split(a)
    bh, bl = split(b)
    one = ah @ bh
    three = ah @ bh + al @ bh + ah @ bl
    scale = np.abs(exact).max()
    e1 = np.abs(one - exact).max() / scale
    e3 = np.abs(three - exact).max() / scale
    bf = lambda x: (x.astype(np.float32).view(np.uint32) & 0xFFFF0000).view(np.float32)      # bf16 by truncation
    eb = np.abs(bf(a) @ bf(b) - exact).max() / scale
    assert e3 < 2e-6 and 2e-5 < e1 < 2e-3 and eb > 4 * e1
    # the split itself is exact to ~22 bits
    assert np.abs((ah.astype(np.float64) + al) - a).max() <= 2.0 ** -21 * np.abs(a).max()
