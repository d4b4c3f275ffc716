"""ORACLE — test infrastructure, NOT product code.

CPU restatement (plain PyTorch fp32, NCHW, functional) of the reference's bin_stage4 hot path:
    /root/reference/models/archs/RDN.py      (network)
    /root/reference/models/loss.py:130-141   (CharbonnierLoss)
    /root/reference/models/bin_model.py:395-425, 486-542 (loss assembly)
    /root/reference/utils/util.py:113-137, 201-208 (tensor2img / PSNR)
    /root/reference/test.py:348-366          (padding rule)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (bin_amd/) never does; it fails loudly when the HIP library is missing.

Pinning: every function here is checked against the *imported reference itself* in the build
container by tests/golden/make_golden.py, which also writes the golden fixtures under
tests/golden/ that the CPU test-suite re-checks this oracle against (the reference has no tests
or golden vectors of its own).

Weights are passed as a flat dict {canonical name: tensor} (see bin_amd/weights.py); the 1332-key
reference state_dict maps onto it through `canon_from_state_dict`.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# alias (reference attribute under `model.`) -> canonical weight set; RDN.py:342-363
ALIASES = OrderedDict([
    ("model1_1", "model1"), ("model1_2", "model1"), ("model1_3", "model1"), ("model1_4", "model1"),
    ("model2_1", "model2"), ("model2_2", "model2"), ("model2_3", "model2"),
    ("model3_1", "model3"), ("model3_2", "model3"),
    ("model4_1", "model4"),
])
D_BLOCKS, C_CONVS = 12, 4


def canon_from_state_dict(sd):
    """1332-key reference state_dict -> {canonical name: tensor} (first alias wins; all aliases of a
    set hold the same tensor in the reference because they are one module object)."""
    out = OrderedDict()
    for k, v in sd.items():
        if k.startswith("model."):
            alias, local = k[len("model."):].split(".", 1)
            name = f"{ALIASES[alias]}.{local}"
            if name not in out:
                out[name] = v
        else:
            out[k] = v
    return out


# ----------------------------------------------------------------------------- RDN.py:107-132
def pixel_reshuffle(x, r=2):
    """space-to-depth: out[b, c*r*r + i*r + j, y, x] = in[b, c, y*r+i, x*r+j]  (RDN.py:123-132)."""
    b, c, h, w = x.shape
    oh, ow = h // r, w // r
    v = x.contiguous().view(b, c, oh, r, ow, r)
    return v.permute(0, 1, 3, 5, 2, 4).contiguous().view(b, c * r * r, oh, ow)


# ----------------------------------------------------------------------------- RDN.py:135-165
def rdb_conv(x, w, b):
    """RDB_Conv: cat(x, relu(conv3x3(x)))  (RDN.py:141-147)."""
    return torch.cat((x, F.relu(F.conv2d(x, w, b, padding=1))), 1)


def rdb(x, W, prefix):
    """RDB: C x rdb_conv -> LFF 1x1 -> + x  (RDN.py:156-165); C = the number of convs the weight dict holds for the block."""
    y = x
    n_convs = 0
    while f"{prefix}.convs.{n_convs}.conv.0.weight" in W:
        n_convs += 1
    for c in range(n_convs):
        y = rdb_conv(y, W[f"{prefix}.convs.{c}.conv.0.weight"], W[f"{prefix}.convs.{c}.conv.0.bias"])
    return F.conv2d(y, W[f"{prefix}.LFF.weight"], W[f"{prefix}.LFF.bias"]) + x


# ----------------------------------------------------------------------------- RDN.py:167-334
def rdn(inputs, W, set_name):
    """RDN_residual_interp_{2,2_1,4_1}_input.forward (RDN.py:210-222, 268-280, 322-334):
    the three classes differ only in the number of input frames (2/3/5)."""
    p = set_name
    shuffled = pixel_reshuffle(torch.cat(tuple(inputs), 1), 2)
    f1 = F.conv2d(shuffled, W[f"{p}.SFENet1.weight"], W[f"{p}.SFENet1.bias"], padding=2)
    x = F.conv2d(f1, W[f"{p}.SFENet2.weight"], W[f"{p}.SFENet2.bias"], padding=1)
    outs = []
    n_blocks = 0
    while f"{p}.RDBs.{n_blocks}.LFF.weight" in W:       # D = the number of dense blocks the weight dict holds (RDN.py:195-198)
        n_blocks += 1
    for d in range(n_blocks):
        x = rdb(x, W, f"{p}.RDBs.{d}")
        outs.append(x)
    x = F.conv2d(torch.cat(outs, 1), W[f"{p}.GFF.0.weight"], W[f"{p}.GFF.0.bias"])
    x = F.conv2d(x, W[f"{p}.GFF.1.weight"], W[f"{p}.GFF.1.bias"], padding=1)
    x = x + f1
    u = F.conv2d(x, W[f"{p}.UPNet.0.weight"], W[f"{p}.UPNet.0.bias"], padding=1)
    u = F.pixel_shuffle(u, 2)
    u = F.conv2d(u, W[f"{p}.UPNet.2.weight"], W[f"{p}.UPNet.2.bias"], padding=1)
    s = inputs[0]
    for t in inputs[1:]:
        s = s + t                      # (B0 + B1 + ...) left to right, RDN.py:221/279/333
    return u + s / len(inputs)


# ----------------------------------------------------------------------------- RDN.py:9-95
def convlstm_cell(x, state, w, b, forget_bias=1.0):
    """ConvLSTMCell.forward (RDN.py:50-95).  state = [c, h] or None (zeros).  Returns (h', [c', h'])."""
    if state is None:
        hidden = w.shape[0] // 4                          # zero state of the cell's hidden size (RDN.py:57-68)
        z = torch.zeros((x.shape[0], hidden) + tuple(x.shape[2:]), dtype=x.dtype)
        state = [z, z]
    c, h = state
    gates = F.conv2d(torch.cat((x, h), 1), w, b, padding=w.shape[-1] // 2)
    i, j, f, o = gates.chunk(4, 1)
    new_c = c * torch.sigmoid(f + forget_bias) + torch.sigmoid(i) * torch.tanh(j)
    new_h = torch.tanh(new_c) * torch.sigmoid(o)
    return new_h, [new_c, new_h]


# ----------------------------------------------------------------------------- RDN.py:337-405
def pyramid(B1, B3, B5, B7, B9, prev, W):
    """RDN_residual_interp_5_input.forward, lstm=True branch (RDN.py:369-389, 403-405)."""
    I2 = rdn((B1, B3), W, "model1")
    I4 = rdn((B3, B5), W, "model1")
    I6 = rdn((B5, B7), W, "model1")
    I8 = rdn((B7, B9), W, "model1")
    if prev[0] is not None:
        p4, p6, p8, p5, p7, p6b = prev
        I3 = rdn((p4, I2, I4), W, "model2")
        I5 = rdn((p6, I4, I6), W, "model2")
        I7 = rdn((p8, I6, I8), W, "model2")
        I4pp = rdn((p5, B3, I3, I5, B5), W, "model3")
        I6pp = rdn((p7, B5, I5, I7, B7), W, "model3")
        I5ppp = rdn((p6b, I4, I4pp, I6pp, I6), W, "model4")
    else:
        I3 = rdn((I2, I2, I4), W, "model2")
        I5 = rdn((I4, I4, I6), W, "model2")
        I7 = rdn((I6, I6, I8), W, "model2")
        I4pp = rdn((I3, B3, I3, I5, B5), W, "model3")
        I6pp = rdn((I5, B5, I5, I7, B7), W, "model3")
        I5ppp = rdn((I4, I4, I4pp, I6pp, I6), W, "model4")
    return I2, I4, I6, I8, I3, I5, I7, I4pp, I6pp, I5ppp


# ----------------------------------------------------------------------------- RDN.py:408-465
CLSTM_FOR_OUTPUT = ((1, "clstm_4_prime"), (2, "clstm_6_prime"), (3, "clstm_8_prime"),
                    (5, "clstm_5_prime_prime"), (6, "clstm_7_prime_prime"),
                    (8, "clstm_6_prime_prime_prime"))


def bin_stage4_forward(frames, W):
    """RDN_residual_interp_5_input_ConvLSTM_L.forward (RDN.py:422-465): two overlapping 5-frame
    windows, six ConvLSTM hand-offs.  Returns the 14-tuple in the reference's order."""
    B1, B3, B5, B7, B9, B11 = frames
    states = [None] * 6
    hidden = [None] * 6
    res = []
    for win in ((B1, B3, B5, B7, B9), (B3, B5, B7, B9, B11)):
        out = pyramid(*win, hidden, W)
        hidden = []
        for k, (idx, nm) in enumerate(CLSTM_FOR_OUTPUT):
            h, states[k] = convlstm_cell(out[idx], states[k], W[f"{nm}.Gates.weight"], W[f"{nm}.Gates.bias"])
            hidden.append(h)
        res.append(out)
    return tuple(res[0]) + (res[1][3], res[1][6], res[1][8], res[1][9])


# ----------------------------------------------------------------------------- loss.py:130-141
def charbonnier(x, y, eps=1e-6):
    d = x - y
    return torch.mean(torch.sqrt(d * d + eps))


def bin_loss(Ft_p, I, eps=1e-6):
    """bin_model.get_loss for nframes=6 / version=2 (bin_model.py:395-425, gt order :530-534).
    `I` maps frame number (2..10) -> sharp frame.  Returns (loss, 14-entry loss_list)."""
    gt = [I[2], I[4], I[6], I[8], I[3], I[5], I[7], I[4], I[6], I[5], I[10], I[9], I[8], I[7]]
    ll = [charbonnier(Ft_p[k], g, eps) for k, g in enumerate(gt)]
    ll.append(charbonnier(Ft_p[1], Ft_p[7], eps))
    ll.append(charbonnier(Ft_p[5], Ft_p[9], eps))
    ll.append(charbonnier(Ft_p[2], Ft_p[8], eps))
    loss = sum(ll) / len(ll)
    return loss, ll[:14]


# ----------------------------------------------------------------------------- module_util.py:34-52
def residual_block_nobn(x, w1, b1, w2, b2):
    """ResidualBlock_noBN.forward (dead code in the reference):
    x + conv3x3(relu(conv3x3(x)))."""
    return x + F.conv2d(F.relu(F.conv2d(x, w1, b1, padding=1)), w2, b2, padding=1)


# ----------------------------------------------------------------------------- util.py:113-137
def tensor2img(t):
    """tensor2img for a 3-D CHW RGB tensor: clamp [0,1], x255, round, RGB->BGR, HWC uint8."""
    a = t.detach().squeeze().float().cpu().clamp(0, 1).numpy()
    a = np.transpose(a[[2, 1, 0], :, :], (1, 2, 0))
    return (a * 255.0).round().astype(np.uint8)


def calculate_psnr(img1, img2):
    """util.py:201-208."""
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))


# ----------------------------------------------------------------------------- test.py:348-366
def pad_sizes(h, w):
    """test.py padding rule -> (left, right, top, bottom)."""
    def one(n):
        if n != ((n >> 7) << 7):
            padded = ((n >> 7) + 1) << 7
            a = int((padded - n) / 2)
            return a, padded - n - a
        return 32, 32
    l, r = one(w)
    t, b = one(h)
    return l, r, t, b


def replicate_pad(x, pads):
    return F.pad(x, pads, mode="replicate")
