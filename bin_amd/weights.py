"""Deterministic synthetic weights for the bin_stage4 network (SURVEY.md §8c "G0").

The reference ships no weights (model_weights/download_adobe_bin.txt is a Drive URL) and
there is no network, so every parity test / bench / smoke run uses this generator.  A tensor's
values depend only on (seed, canonical parameter name), so the oracle, the reference (when it is
importable in the build container) and the HIP path all see bit-identical fp32 weights without a
46 MB blob in the repo.

Magnitudes follow the reference's initialisers: nn.Conv2d default (kaiming-uniform a=sqrt(5)
=> U(+-1/sqrt(fan_in)) for weight and bias) for every RDN conv (reference models/archs/RDN.py:
141,162,187-208 never call a custom init) and xavier-uniform weight / zero bias for the ConvLSTM
gates (RDN.py:26-38).
"""
import math
import zlib
from collections import OrderedDict

import numpy as np

# (sub-net attribute, number of input frames) in construction order, RDN.py:342-363
RDN_SETS = (("model1", 2), ("model2", 3), ("model3", 5), ("model4", 5))
# alias names under `model.` -> canonical weight set (RDN.py:342-363: shared module objects)
RDN_ALIASES = OrderedDict([
    ("model1_1", "model1"), ("model1_2", "model1"), ("model1_3", "model1"), ("model1_4", "model1"),
    ("model2_1", "model2"), ("model2_2", "model2"), ("model2_3", "model2"),
    ("model3_1", "model3"), ("model3_2", "model3"),
    ("model4_1", "model4"),
])
CLSTM_NAMES = ("clstm_4_prime", "clstm_6_prime", "clstm_8_prime",
               "clstm_5_prime_prime", "clstm_7_prime_prime", "clstm_6_prime_prime_prime")

G0, D, C, G = 96, 12, 4, 32   # RDN.py:418 (GO=96, D=12), RDN.py:171-172 (C=4, G=32)


def rdn_param_shapes(n_in, shape=None):
    """Ordered {local name: shape} of one RDN sub-net with `n_in` input frames (RDN.py:167-334); `shape` = (G0, D, C, G)
    constructor arguments, default bin_stage4's."""
    G0, D, C, G = shape if shape is not None else (96, 12, 4, 32)
    s = OrderedDict()
    s["SFENet1.weight"] = (G0, 12 * n_in, 5, 5)
    s["SFENet1.bias"] = (G0,)
    s["SFENet2.weight"] = (G0, G0, 3, 3)
    s["SFENet2.bias"] = (G0,)
    for d in range(D):
        for c in range(C):
            s[f"RDBs.{d}.convs.{c}.conv.0.weight"] = (G, G0 + c * G, 3, 3)
            s[f"RDBs.{d}.convs.{c}.conv.0.bias"] = (G,)
        s[f"RDBs.{d}.LFF.weight"] = (G0, G0 + C * G, 1, 1)
        s[f"RDBs.{d}.LFF.bias"] = (G0,)
    s["GFF.0.weight"] = (G0, D * G0, 1, 1)
    s["GFF.0.bias"] = (G0,)
    s["GFF.1.weight"] = (G0, G0, 3, 3)
    s["GFF.1.bias"] = (G0,)
    s["UPNet.0.weight"] = (256, G0, 3, 3)
    s["UPNet.0.bias"] = (256,)
    s["UPNet.2.weight"] = (3, 64, 3, 3)
    s["UPNet.2.bias"] = (3,)
    return s


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(name.encode())]))


def _uniform(seed, name, shape, bound):
    return _rng(seed, name).uniform(-bound, bound, size=shape).astype(np.float32)


def canonical_weights(seed=0):
    """{canonical name: float32 ndarray} for the 4 RDN weight sets + 6 ConvLSTM cells (540 tensors)."""
    out = OrderedDict()
    for nm in CLSTM_NAMES:
        shape = (12, 6, 3, 3)
        fan_in, fan_out = 6 * 9, 12 * 9
        out[f"{nm}.Gates.weight"] = _uniform(seed, f"{nm}.Gates.weight", shape,
                                             math.sqrt(6.0 / (fan_in + fan_out)))
        out[f"{nm}.Gates.bias"] = np.zeros((12,), np.float32)
    for set_name, n_in in RDN_SETS:
        for local, shape in rdn_param_shapes(n_in).items():
            wshape = shape if len(shape) == 4 else None
            if wshape is None:
                # bias bound uses the fan_in of the matching weight
                wshape = rdn_param_shapes(n_in)[local.replace(".bias", ".weight")]
            fan_in = wshape[1] * wshape[2] * wshape[3]
            out[f"{set_name}.{local}"] = _uniform(seed, f"{set_name}.{local}", shape,
                                                  1.0 / math.sqrt(fan_in))
    return out


def general_rdn_weights(seed, n_in, shape):
    """{local name: float32 ndarray} for ONE RDN sub-network of an arbitrary (G0, D, C, G) — the parity tests of the general
    constructor arguments (reference RDN.py:168-186).  Same initialiser magnitudes as canonical_weights; the values depend on
    (seed, n_in, shape, local name)."""
    shapes = rdn_param_shapes(n_in, shape)
    tag = "rdn%d@%d.%d.%d.%d." % ((n_in,) + tuple(shape))
    out = OrderedDict()
    for local, shp in shapes.items():
        wshape = shp if len(shp) == 4 else shapes[local.replace(".bias", ".weight")]
        fan_in = wshape[1] * wshape[2] * wshape[3]
        out[local] = _uniform(seed, tag + local, shp, 1.0 / math.sqrt(fan_in))
    return out


def general_lstm_weights(seed, input_size, hidden_size, ksize):
    """(Gates.weight, Gates.bias) of a ConvLSTMCell of any size: xavier-uniform weight, zero bias (reference RDN.py:26-38)."""
    shape = (4 * hidden_size, input_size + hidden_size, ksize, ksize)
    fan_in, fan_out = shape[1] * ksize * ksize, shape[0] * ksize * ksize
    tag = "clstm@%d.%d.%d." % (input_size, hidden_size, ksize)
    return (_uniform(seed, tag + "Gates.weight", shape, math.sqrt(6.0 / (fan_in + fan_out))),
            _uniform(seed, tag + "Gates.bias", (shape[0],), 0.05))      # non-zero bias: the fixture should see it


def trained_like_weights(seed=0, boost_layer="model2.RDBs.5.convs.2.conv.0.weight", boost=50.0):
    """Canonical-name weights with the value DISTRIBUTION of a trained network rather than of an initialiser (VERDICT r05 item 5b;
    the real `adobe_bin.pth` is a Drive link): inside a conv layer the magnitudes are log-uniform over THREE decades below the
    layer's largest weight, 30 % of the elements are exact zeros, biases are zero, every layer has its own gain (log-uniform
    0.3 ... 2 for unit-variance inputs) and ONE layer (`boost_layer`) is scaled by `boost` on top — the hi/lo fp16 planes see
    weights from ~1e-5 to > 10, most of them where the lo plane is an fp16 subnormal (|v| < 2^-3), and activations that swing
    by orders of magnitude from layer to layer.  ConvLSTM gates keep their initialiser (they run in fp32)."""
    base = canonical_weights(seed)
    out = OrderedDict()
    for name, v in base.items():
        if ".Gates." in name:
            out[name] = v
            continue
        if name.endswith(".bias"):
            out[name] = np.zeros_like(v)
            continue
        r = _rng(seed + 1000, "trained." + name)
        fan_in = v.shape[1] * v.shape[2] * v.shape[3]
        gain = 10.0 ** r.uniform(-0.5, 0.3)
        top = gain / math.sqrt(0.0507 * fan_in)          # E[(10^u)^2] over u ~ U(-3, 0), times the 70 % that are non-zero
        mag = top * 10.0 ** r.uniform(-3.0, 0.0, size=v.shape)
        w = (mag * r.choice((-1.0, 1.0), size=v.shape)).astype(np.float32)
        w[r.random(v.shape) < 0.30] = 0.0
        if name == boost_layer:
            w *= np.float32(boost)
        out[name] = w
    return out


def state_dict_from_canonical(canon):
    """The 1332-key aliased state_dict (torch tensors) of a canonical-name weight table."""
    import torch
    sd = OrderedDict()
    for nm in CLSTM_NAMES:
        for p in ("weight", "bias"):
            sd[f"{nm}.Gates.{p}"] = torch.from_numpy(canon[f"{nm}.Gates.{p}"].copy())
    for alias, set_name in RDN_ALIASES.items():
        n_in = dict(RDN_SETS)[set_name]
        for local in rdn_param_shapes(n_in):
            sd[f"model.{alias}.{local}"] = torch.from_numpy(canon[f"{set_name}.{local}"].copy())
    return sd


def reference_state_dict(seed=0):
    """The 1332-key aliased state_dict of RDN_residual_interp_5_input_ConvLSTM_L
    (reference RDN.py:408-465; key naming per SURVEY.md §8b) as torch tensors."""
    import torch
    canon = canonical_weights(seed)
    sd = OrderedDict()
    for nm in CLSTM_NAMES:
        for p in ("weight", "bias"):
            sd[f"{nm}.Gates.{p}"] = torch.from_numpy(canon[f"{nm}.Gates.{p}"].copy())
    for alias, set_name in RDN_ALIASES.items():
        n_in = dict(RDN_SETS)[set_name]
        for local in rdn_param_shapes(n_in):
            sd[f"model.{alias}.{local}"] = torch.from_numpy(canon[f"{set_name}.{local}"].copy())
    return sd


def synthetic_frames(seed, n, h, w, count=6):
    """`count` seeded U[0,1) fp32 frames [n,3,h,w] (SURVEY.md §8d synthetic inputs)."""
    import torch
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    return [torch.from_numpy(rng.random((n, 3, h, w), dtype=np.float32)) for _ in range(count)]
