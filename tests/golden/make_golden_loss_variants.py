#!/usr/bin/env python3
"""Golden fixture for the OTHER branches of the reference wrapper's loss (g11_loss_variants.npz): bin_model.py:52-60 offers
three pixel criteria ('cb', 'l1' = nn.L1Loss(reduction='sum'), 'l2' = nn.MSELoss(reduction='sum')) and get_loss
(bin_model.py:395-425) adds the three cycle terms only for nframes == 6 and version == 2; the yml the authors ship uses
('cb', version 2), which g3_train / g9_train_steps pin.  Here ONE optimize_parameters() of the reference wrapper per variant
(version 1 with 'cb', version 2 with 'l1' and 'l2', version 1 with 'l2', and pixel_weight != 1) on one 32x32 batch: loss, the
14-entry loss list, the gradient norms of all 540 parameters and two sampled post-Adam tensors.
Build container only (imports /root/reference); run:  python tests/golden/make_golden_loss_variants.py"""
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

from models import create_model                                   # noqa: E402  (reference models/__init__.py)
from bin_amd.weights import reference_state_dict                  # noqa: E402
from loss_variants import VARIANTS, SAMPLE                        # noqa: E402  (tests/loss_variants.py: shared with the tests)

torch.set_num_threads(8)
g = np.random.default_rng(29)
B, H, W = 1, 32, 32
batch = {"LQs": torch.from_numpy(g.random((B, 6, 3, H, W), dtype=np.float32)),
         "GTenh": torch.from_numpy(g.random((B, 6, 3, H, W), dtype=np.float32)),
         "GTinp": torch.from_numpy(g.random((B, 5, 3, H, W), dtype=np.float32)), "key": "x"}
out = {"seed_w": 0, "LQs": batch["LQs"].numpy(), "GTenh": batch["GTenh"].numpy(), "GTinp": batch["GTinp"].numpy()}
for tag, version, crit, weight in VARIANTS:
    opt = {
        "model": "bin", "gpu_ids": None, "is_train": True, "dist": False,
        "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": version},
        "path": {"pretrain_model_G": None, "strict_load": True, "models": "/tmp", "training_state": "/tmp"},
        "train": {"pixel_criterion": crit, "pixel_weight": weight, "weight_decay_G": 0, "ft_tsa_only": None,
                  "lr_G": 1e-4, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000],
                  "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False},
    }
    model = create_model(opt)
    model.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    model.feed_data(batch)
    model.optimize_parameters(1)
    named = dict(model.netG.module.named_parameters())
    out[tag + ".loss"] = np.float64(float(model.loss.detach()))
    out[tag + ".loss_list"] = np.array([float(l.detach()) for l in model.loss_list], dtype=np.float64)
    out[tag + ".grad_norms"] = np.array([float(p.grad.double().norm()) if p.grad is not None else 0.0
                                         for p in named.values()], dtype=np.float64)
    for n in SAMPLE:
        out[tag + ".after." + n] = named[n].detach().numpy().copy()
    print(tag, "loss", out[tag + ".loss"], "terms", len(model.loss_list))
# ft_tsa_only = 3 (bin_model.py:66-87,131-132): two parameter groups (the second one empty for bin_stage4), steps 1 and 2 run
# with group 0's rate forced to zero, step 3 trains — parameters must not move before step 3
opt["network_G"]["version"], opt["train"]["pixel_criterion"], opt["train"]["pixel_weight"] = 2, "cb", 1.0
opt["train"]["ft_tsa_only"] = 3
model = create_model(opt)
model.netG.module.load_state_dict(reference_state_dict(0), strict=True)
named = dict(model.netG.module.named_parameters())
probe = "model.model4_1.UPNet.2.weight"
before = named[probe].detach().clone()
moved = []
for step in (1, 2, 3):
    if step == 3:                                   # what a training loop does before the step it un-freezes (update_learning_rate)
        for grp in model.optimizer_G.param_groups:
            grp["lr"] = opt["train"]["lr_G"]
    model.feed_data(batch)
    model.optimize_parameters(step)
    moved.append(float((named[probe].detach() - before).abs().max()))
sd = model.optimizer_G.state_dict()
out["ft.group_sizes"] = np.array([len(gp["params"]) for gp in sd["param_groups"]])
out["ft.moved"] = np.array(moved)
out["ft.loss3"] = np.float64(float(model.loss.detach()))
out["ft.after3"] = named[probe].detach().numpy().copy()
print("ft_tsa_only: groups", out["ft.group_sizes"], "moved", moved)
out["names"] = np.array(list(named.keys()))
np.savez_compressed(os.path.join(HERE, "g11_loss_variants.npz"), **out)
