"""-m gpu: Charbonnier / l1 / l2 criteria and bin_model.get_loss's fused 17-term form against the reference wrapper's fixtures."""
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


def test_charbonnier_golden():
    from bin_amd import ops
    g = load_golden("g1_charbonnier")
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    loss = ops.charbonnier(x, y)
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    gx = ops.charbonnier_grad(x, y, torch.ones((), device="cuda"))
    assert float((gx.cpu() - torch.from_numpy(g["gx"])).abs().max()) <= 1e-9 + 1e-5 * float(np.abs(g["gx"]).max())


def _train_opt_r3(tmp_path, dist=False):
    return {"model": "bin", "gpu_ids": [0], "is_train": True, "dist": dist,
            "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2, "precision": "f16x3"},
            "path": {"pretrain_model_G": None, "strict_load": True, "models": str(tmp_path), "training_state": str(tmp_path)},
            "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                      "lr_G": 1e-4, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000],
                      "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}


def _batch(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    return {"LQs": torch.rand(B, 6, 3, S, S, generator=g), "GTenh": torch.rand(B, 6, 3, S, S, generator=g),
            "GTinp": torch.rand(B, 5, 3, S, S, generator=g)}


@pytest.mark.parametrize("tag,version,crit,weight", __import__("loss_variants").VARIANTS)
def test_loss_variants_match_reference_wrapper_on_the_gpu(tmp_path, tag, version, crit, weight):
    """g11_loss_variants (reference wrapper: version 1 / 2, 'cb' / 'l1' / 'l2', pixel_weight): one optimize_parameters() with
    the HIP network and the device-side criteria — loss, the 14 terms, all 540 gradient norms, sampled post-Adam values."""
    from bin_amd.models import create_model
    from bin_amd.weights import reference_state_dict
    from conftest import load_golden
    g = load_golden("g11_loss_variants")
    opt = _train_opt_r3(tmp_path)
    opt["network_G"]["version"] = version
    opt["train"]["pixel_criterion"], opt["train"]["pixel_weight"] = crit, weight
    m = create_model(opt)
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    m.feed_data({"LQs": torch.from_numpy(g["LQs"]), "GTenh": torch.from_numpy(g["GTenh"]), "GTinp": torch.from_numpy(g["GTinp"])})
    m.optimize_parameters(1)
    ref = float(g[tag + ".loss"])
    assert abs(float(m.loss) - ref) <= 4e-6 * max(1.0, abs(ref)), (float(m.loss), ref)
    ll = np.array([float(l) for l in m.loss_list])
    assert len(ll) == 14 and np.abs(ll - g[tag + ".loss_list"]).max() <= 4e-6 * max(1.0, np.abs(g[tag + ".loss_list"]).max())
    named = dict(m.netG.module.named_parameters())
    norms = np.array([float(p.grad.double().norm()) if p.grad is not None else 0.0 for p in named.values()])
    refn = g[tag + ".grad_norms"]
    rel = np.abs(norms - refn) / (np.abs(refn) + 1e-6 * refn.max())
    assert rel.max() <= 5e-3, (list(named.keys())[int(rel.argmax())], float(rel.max()))
    for key in g.files:
        if key.startswith(tag + ".after."):
            d = (named[key[len(tag) + 7:]].detach().cpu() - torch.from_numpy(g[key])).abs()
            assert float(d.mean()) <= 2e-6 and float((d > 5e-5).float().mean()) <= 0.01, (key, float(d.mean()), float(d.max()))


# ------------------------------------------------------------------------------------------------ l1 / l2 criteria
@pytest.mark.parametrize("kind", ["l1", "l2"])
def test_l1_l2_sum_criteria_on_the_hip_kernels(kind, tmp_path):
    """`pixel_criterion: l1 | l2` (reference bin_model.py:52-57: nn.L1Loss / nn.MSELoss with reduction='sum') run on the same
    HIP reduction kernels as Charbonnier: value and gradient against torch's own fp32 ops, and a whole training step with
    that criterion through the wrapper."""
    from bin_amd.models import create_model
    from bin_amd.models.loss import L1SumLoss, L2SumLoss
    from bin_amd.weights import reference_state_dict
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 37, 53, generator=g).cuda().requires_grad_(True)
    y = torch.rand(2, 3, 37, 53, generator=g).cuda()
    with torch.no_grad():
        y[0, 0, :5] = x[0, 0, :5]                                  # exact ties: sign(0) = 0 in the L1 gradient
    mine = (L1SumLoss() if kind == "l1" else L2SumLoss())(x, y)
    (mine * 0.37).backward()
    gx = x.grad.clone()
    x.grad = None
    ref_mod = torch.nn.L1Loss(reduction="sum") if kind == "l1" else torch.nn.MSELoss(reduction="sum")
    ref = ref_mod(x, y)
    (ref * 0.37).backward()
    assert abs(float(mine) - float(ref)) <= 2e-6 * abs(float(ref))
    assert float((gx - x.grad).abs().max()) <= 1e-6
    opt = _train_opt_r3(tmp_path)
    opt["train"]["pixel_criterion"] = kind
    m = create_model(opt)
    assert type(m.cri_pix).__name__ == ("L1SumLoss" if kind == "l1" else "L2SumLoss")
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    m.feed_data(_batch(1, 64, 3))
    before = torch.cat([p.detach().reshape(-1) for p in m.netG.module.parameters()]).clone()
    m.optimize_parameters(1)
    after = torch.cat([p.detach().reshape(-1) for p in m.netG.module.parameters()])
    assert torch.isfinite(m.loss) and float(m.loss) > 0 and torch.isfinite(after).all()
    assert float((after - before).abs().max()) > 0


@pytest.mark.parametrize("kind", ["cb", "l1", "l2"])
def test_fused_multi_term_loss_equals_the_per_term_path_bit_for_bit(kind):
    """bin_model.get_loss as one autograd node (binhip_multi_loss_fwd / _bwd: all terms + their mean in two launches, every
    gradient in one) against what it replaces — one _PixelLossFn per term, `sum(list) / len(list)` and autograd's accumulation
    in torch ops: the loss, the 17 terms and all 14 + 3 gradients agree BIT FOR BIT, for the three criteria, with tensors that sit
    in two terms (the cycle pairs) on either side."""
    from bin_amd.models.loss import CharbonnierLoss, L1SumLoss, L2SumLoss, multi_term_loss
    crit = {"cb": CharbonnierLoss, "l1": L1SumLoss, "l2": L2SumLoss}[kind]()
    g = torch.Generator().manual_seed(17)
    mk = lambda: torch.rand(2, 3, 40, 56, generator=g).cuda()
    outs = [mk().requires_grad_(True) for _ in range(14)]
    gts = [mk() for _ in range(14)]
    gts[3].requires_grad_(True)                                   # a target that wants a gradient too (sign -1)

    def pairs(o):
        return [(o[i], gts[i]) for i in range(14)] + [(o[1], o[7]), (o[5], o[9]), (o[2], o[8])]

    loss, terms = multi_term_loss(crit, pairs(outs))
    assert len(terms) == 17 and loss.grad_fn is not None
    (0.7 * loss).backward()
    fused = [o.grad.clone() for o in outs] + [gts[3].grad.clone()]
    for o in outs:
        o.grad = None
    gts[3].grad = None
    per = [crit(x, y) for x, y in pairs(outs)]
    ref = sum(per) / len(per)
    (0.7 * ref).backward()
    assert torch.equal(loss.detach(), ref.detach())
    assert all(torch.equal(a.detach(), b.detach()) for a, b in zip(terms, per))
    for i, (a, b) in enumerate(zip(fused, [o.grad for o in outs] + [gts[3].grad])):
        assert torch.equal(a, b), (i, float((a - b).abs().max()))
    # an injected (non-product) criterion takes the plain loop
    plain, pt = multi_term_loss(lambda x, y: ((x - y) ** 2).mean(), pairs([o.detach() for o in outs]))
    assert len(pt) == 17 and float(plain) > 0


def test_fused_loss_with_more_gradients_than_one_launch_holds_and_with_mixed_shapes():
    """advisor r05: (a) 13-24 pairs whose BOTH sides want a gradient need more than BINHIP_LOSS_MAX_TERMS gradient outputs: the
    backward splits them over launches instead of indexing past the descriptor; (b) pairs of equal numel but different shapes (or
    of another device) are not fused — they take the per-term path, which is where the reference's criterion would see them."""
    from bin_amd import _lib as L
    from bin_amd.models.loss import CharbonnierLoss, multi_term_loss
    crit = CharbonnierLoss()
    g = torch.Generator().manual_seed(23)
    mk = lambda *s: torch.rand(*s, generator=g).cuda().requires_grad_(True)
    xs, ys = [mk(1, 3, 24, 40) for _ in range(14)], [mk(1, 3, 24, 40) for _ in range(14)]
    assert 2 * len(xs) > L.LOSS_MAX_TERMS
    loss, terms = multi_term_loss(crit, list(zip(xs, ys)))
    loss.backward()
    fused = [t.grad.clone() for t in xs + ys]
    for t in xs + ys:
        t.grad = None
    per = [crit(x, y) for x, y in zip(xs, ys)]
    (sum(per) / len(per)).backward()
    assert torch.equal(loss.detach(), (sum(per) / len(per)).detach())
    for a, t in zip(fused, xs + ys):
        assert torch.equal(a, t.grad)
    # (b) same numel, different shapes: must not be summed as if they were one size
    a, b = mk(1, 3, 24, 40), mk(1, 3, 40, 24)
    loss2, terms2 = multi_term_loss(crit, [(xs[0], ys[0]), (a, b.reshape(1, 3, 24, 40))])
    assert len(terms2) == 2 and loss2.grad_fn is not None
    mixed = [(xs[0], ys[0]), (a.reshape(3, 24, 40), a.detach().reshape(3, 24, 40) * 0.5)]       # second pair: another shape
    loss3, terms3 = multi_term_loss(crit, mixed)
    ref3 = [crit(x, y) for x, y in mixed]
    assert torch.equal(loss3.detach(), (sum(ref3) / 2).detach())
