#!/usr/bin/env python3
"""Golden fixture g12_stitch.npz: the tile geometry of the REFERENCE's VideoBaseModel.test_stitch (Video_base_model.py:
189-280: pad to whole 320x180 tiles, 32-px replicate halo, one generator call per tile, interior copy at x4 scale).

The reference class cannot be imported as it stands — Video_base_model.py:11 imports `CharbonnierLossPlusSSIM`, which
models/loss.py does not define — so the missing NAME is injected (an empty class; the stitcher never touches a loss), cv2 /
torchvision are stubbed as in make_golden.py, and `test_stitch` is run as the unbound function on a minimal object carrying
the three attributes it reads (netG, var_L, device).  The generator is tests/stitch_cases.py::StubSR (exact torch ops).
Build container only (imports /root/reference); run:  python tests/golden/make_golden_stitch.py"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, "/root/reference")
for name in ("cv2", "torchvision", "torchvision.utils", "torchvision.models"):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.make_grid = lambda *a, **k: None
        sys.modules[name] = m
sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
sys.modules["torchvision"].models = sys.modules["torchvision.models"]

import models.loss as REF_LOSS                                     # noqa: E402
if not hasattr(REF_LOSS, "CharbonnierLossPlusSSIM"):
    REF_LOSS.CharbonnierLossPlusSSIM = type("CharbonnierLossPlusSSIM", (torch.nn.Module,), {})   # the name only
from models.Video_base_model import VideoBaseModel as RefVideoBaseModel   # noqa: E402
import stitch_cases as SC                                          # noqa: E402

torch.set_num_threads(8)
net = SC.StubSR().eval()
calls = []


class Recording(torch.nn.Module):            # the tile rectangles the reference hands its generator
    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, x):
        calls.append(tuple(x.shape))
        return self.inner(x)


me = types.SimpleNamespace(netG=Recording(net), var_L=SC.frame(), device=torch.device("cpu"))
RefVideoBaseModel.test_stitch(me)
y = me.fake_H
assert tuple(y.shape) == (1, 3, SC.LR_H * SC.SCALE, SC.LR_W * SC.SCALE), y.shape
assert len(calls) == 9 and set(calls) == {(1, SC.N_FRAMES, 3, 180 + 64, 320 + 64)}, calls
out = {k: v.numpy() for k, v in SC.sample(y).items()}
out["sha256"] = np.frombuffer(hashlib.sha256(y.numpy().tobytes()).digest(), dtype=np.uint8)
out["n_calls"] = np.int64(len(calls))
out["crop_shape"] = np.array(calls[0], dtype=np.int64)
out["mean"] = np.float64(y.double().mean())
np.savez_compressed(os.path.join(HERE, "g12_stitch.npz"), **out)
print("wrote g12_stitch.npz", {k: getattr(v, "shape", v) for k, v in out.items()}, os.path.getsize(os.path.join(HERE, "g12_stitch.npz")) // 1024, "KiB")
