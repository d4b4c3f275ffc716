#!/usr/bin/env python3
"""Golden fixture g13_videobase_step.npz: the REFERENCE's VideoBaseModel training step (Video_base_model.py:22-187 —
`__init__` with its criteria / Adam parameter groups / schedulers, `feed_data`, `optimize_parameters` :134-158,
`optimize_parameters_without_schudlue` :161-181, `set_params_lr_zero`, `get_current_log`, `test`, `update_learning_rate`) —
closing SURVEY row a17.

The reference class cannot be imported as it stands (Video_base_model.py:11 imports `CharbonnierLossPlusSSIM`, which models/loss.py
does not define): the missing NAME is injected as in make_golden_stitch.py, cv2 / torchvision are stubbed, and
`models.networks.define_G` is pointed at tests/videobase_cases.py::StubVSR — the class calls its generator with ONE [B,N,C,H,W]
tensor, which `bin_stage4` does not accept, so the reference wrapper can only train a single-tensor generator.  The whole class then
runs unmodified on the CPU: per case (videobase_cases.CASES) STEPS steps of feed_data -> step method -> update_learning_rate; stored
are each step's logged loss and learning rates, the parameters after every step, and `test()`'s output after the last (whole for
two cases, its mean for the rest).
`optimize_parameters` unpacks `loss, loss_tmp = self.cri_pix(...)`, which no criterion of models/loss.py returns: for those cases
the instance's criterion (the REFERENCE's own CharbonnierLoss) is wrapped to return `(loss, None)` after construction.
Build container only (imports /root/reference); run:  python tests/golden/make_golden_videobase_step.py"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
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
import models.networks as REF_NETWORKS                             # noqa: E402
from models.Video_base_model import VideoBaseModel as RefVideoBaseModel   # noqa: E402
import videobase_cases as VC                                       # noqa: E402

torch.set_num_threads(4)
REF_NETWORKS.define_G = lambda opt: VC.StubVSR()
out = {}
tmp = tempfile.mkdtemp()
data = VC.batch()
for case, (ft, crit, method, pair) in VC.CASES.items():
    m = RefVideoBaseModel(VC.opt(tmp, ft, crit))
    assert type(m.cri_pix).__module__ in ("models.loss", "torch.nn.modules.loss"), type(m.cri_pix)
    if pair:
        m.cri_pix = VC.PairCriterion(m.cri_pix)
    names = [n for n, _ in m.netG.module.named_parameters()]
    out[f"{case}/groups"] = np.array([len(g["params"]) for g in m.optimizer_G.param_groups], dtype=np.int64)
    for step in range(1, VC.STEPS + 1):
        m.feed_data(data)
        getattr(m, method)(step)
        out[f"{case}/s{step}/lr_used"] = np.array([g["lr"] for g in m.optimizer_G.param_groups], dtype=np.float64)
        m.update_learning_rate(step, warmup_iter=-1)
        log = m.get_current_log()
        assert list(log) == ["l_pix"], log
        out[f"{case}/s{step}/l_pix"] = np.float64(log["l_pix"])
        out[f"{case}/s{step}/lr_next"] = np.array(m.get_current_learning_rate() if isinstance(m.get_current_learning_rate(), list)
                                                  else [m.get_current_learning_rate()], dtype=np.float64)
        for n, p in m.netG.module.named_parameters():
            out[f"{case}/s{step}/{n}"] = p.detach().numpy().copy()
    m.feed_data(data, need_GT=False)
    m.test()
    assert m.netG.training                      # test() switches back (Video_base_model.py:186)
    out[f"{case}/test_mean"] = np.float64(m.fake_H.double().mean())
    if case in ("cb_pair", "l1_plain_noschedule_ft"):
        out[f"{case}/test"] = m.fake_H.numpy().copy()
    print(case, "groups", out[f"{case}/groups"], "losses", [float(out[f"{case}/s{s}/l_pix"]) for s in range(1, VC.STEPS + 1)],
          "lr", [out[f"{case}/s{s}/lr_used"].tolist() for s in range(1, VC.STEPS + 1)])
np.savez_compressed(os.path.join(HERE, "g13_videobase_step.npz"), **out)
print("wrote g13_videobase_step.npz", len(out), "arrays", os.path.getsize(os.path.join(HERE, "g13_videobase_step.npz")) // 1024, "KiB")
