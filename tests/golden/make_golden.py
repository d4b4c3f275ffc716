#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (build container
only: /root/reference is not present on the GPU box) and, in the same run, assert that
oracle/rdn_oracle.py reproduces the reference on every fixture.

Run:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)

Fixtures are data only (seeded inputs -> reference outputs); weights are never stored, they come
from bin_amd/weights.py (seed recorded in each file).  cv2 / torchvision are absent in this image;
the reference's wrapper imports them for PNG IO only, so two empty stub modules are injected (IO
symbols are never called here).
"""
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

for name in ("cv2", "torchvision", "torchvision.utils", "torchvision.models"):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.make_grid = lambda *a, **k: None
        sys.modules[name] = m
sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
sys.modules["torchvision"].models = sys.modules["torchvision.models"]

import models.archs.RDN as REF_RDN            # noqa: E402  (the reference)
import models.module_util as REF_MU           # noqa: E402
from models.loss import CharbonnierLoss as RefCharbonnier  # noqa: E402

from bin_amd.weights import reference_state_dict, canonical_weights, synthetic_frames  # noqa: E402
from oracle import rdn_oracle as O             # noqa: E402

SEED_W = 0
torch.set_num_threads(8)


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"wrote {path} ({os.path.getsize(path)/1024:.0f} KiB)")


def close(a, b, tol, what):
    d = float((a - b).abs().max())
    assert d <= tol, f"oracle != reference on {what}: {d}"
    return d


def main():
    canon = {k: t(v) for k, v in canonical_weights(SEED_W).items()}
    sd = reference_state_dict(SEED_W)

    # ---------------- G1: per-op ----------------
    g = rng(101)
    x = t(g.standard_normal((2, 6, 8, 12), dtype=np.float32))
    y = REF_RDN.pixel_reshuffle(x, 2)
    assert torch.equal(y, O.pixel_reshuffle(x, 2))
    assert torch.equal(y, torch.nn.functional.pixel_unshuffle(x, 2))
    save("g1_pixel_reshuffle", x=x, y=y)

    # conv shapes on the live path, each with grads
    convs = OrderedDict([
        ("k2_sfe1_24", ("model1.SFENet1", 5, 24, 96)),
        ("k2_sfe1_36", ("model2.SFENet1", 5, 36, 96)),
        ("k2_sfe1_60", ("model3.SFENet1", 5, 60, 96)),
        ("k3_sfe2", ("model1.SFENet2", 3, 96, 96)),
        ("k4_rdbconv0", ("model1.RDBs.0.convs.0.conv.0", 3, 96, 32)),
        ("k4_rdbconv1", ("model1.RDBs.0.convs.1.conv.0", 3, 128, 32)),
        ("k4_rdbconv2", ("model1.RDBs.0.convs.2.conv.0", 3, 160, 32)),
        ("k4_rdbconv3", ("model1.RDBs.0.convs.3.conv.0", 3, 192, 32)),
        ("k5_lff", ("model1.RDBs.0.LFF", 1, 224, 96)),
        ("k6_gff0", ("model1.GFF.0", 1, 1152, 96)),
        ("k8_up0", ("model1.UPNet.0", 3, 96, 256)),
        ("k9_up2", ("model1.UPNet.2", 3, 64, 3)),
    ])
    arrs = {}
    for key, (wname, ks, cin, cout) in convs.items():
        w = canon[wname + ".weight"].clone().requires_grad_(True)
        b = canon[wname + ".bias"].clone().requires_grad_(True)
        xin = t(g.standard_normal((1, cin, 12, 20), dtype=np.float32)).requires_grad_(True)
        conv = torch.nn.Conv2d(cin, cout, ks, padding=(ks - 1) // 2, stride=1)   # as RDN.py:141/187
        with torch.no_grad():
            conv.weight.copy_(w)
            conv.bias.copy_(b)
        yy = conv(xin)
        gy = t(g.standard_normal(tuple(yy.shape), dtype=np.float32))
        yy.backward(gy)
        arrs[key + ".x"] = xin.detach()
        arrs[key + ".y"] = yy.detach()
        arrs[key + ".gy"] = gy
        arrs[key + ".gx"] = xin.grad
        arrs[key + ".gw"] = conv.weight.grad
        arrs[key + ".gb"] = conv.bias.grad
    save("g1_convs", seed_w=SEED_W, **arrs)

    # Charbonnier (loss.py:130-141)
    a = t(g.random((2, 3, 20, 28), dtype=np.float32)).requires_grad_(True)
    bb = t(g.random((2, 3, 20, 28), dtype=np.float32))
    l = RefCharbonnier()(a, bb)
    l.backward()
    close(l.detach(), O.charbonnier(a.detach(), bb), 0, "charbonnier")
    save("g1_charbonnier", x=a.detach(), y=bb, loss=l.detach(), gx=a.grad)

    # ---------------- G2: per-block ----------------
    ref = REF_RDN.bin_stage4_lstm()
    ref.load_state_dict(sd, strict=True)
    ref.eval()

    # RDB
    rdb_mod = ref.model.model1_1.RDBs[0]
    xr = t(g.standard_normal((1, 96, 12, 16), dtype=np.float32)).requires_grad_(True)
    yr = rdb_mod(xr)
    gyr = t(g.standard_normal(tuple(yr.shape), dtype=np.float32))
    for p in rdb_mod.parameters():
        p.grad = None
    yr.backward(gyr)
    close(yr.detach(), O.rdb(xr.detach(), canon, "model1.RDBs.0"), 1e-6, "rdb")
    rdb_arrs = dict(x=xr.detach(), y=yr.detach(), gy=gyr, gx=xr.grad)
    for n_, p in rdb_mod.named_parameters():
        rdb_arrs["g." + n_] = p.grad.clone()
    save("g2_rdb", seed_w=SEED_W, **rdb_arrs)

    # the three RDN variants
    for set_name, mod, k in (("model1", ref.model.model1_1, 2), ("model2", ref.model.model2_1, 3),
                             ("model3", ref.model.model3_1, 5), ("model4", ref.model.model4_1, 5)):
        ins = [t(g.random((1, 3, 32, 32), dtype=np.float32)) for _ in range(k)]
        with torch.no_grad():
            out = mod(*ins)
            d = close(out, O.rdn(ins, canon, set_name), 1e-5, f"rdn {set_name}")
        print(f"rdn {set_name}: oracle-vs-ref max diff {d:.2e}")
        save(f"g2_rdn_{set_name}", seed_w=SEED_W, y=out, **{f"in{i}": v for i, v in enumerate(ins)})

    # ConvLSTM with None and non-None state
    cell = ref.clstm_6_prime
    xl = t(g.random((2, 3, 20, 24), dtype=np.float32))
    with torch.no_grad():
        h1, st1 = cell(xl, None)
        xl2 = t(g.random((2, 3, 20, 24), dtype=np.float32))
        h2, st2 = cell(xl2, st1)
        oh1, ost1 = O.convlstm_cell(xl, None, canon["clstm_6_prime.Gates.weight"], canon["clstm_6_prime.Gates.bias"])
        oh2, ost2 = O.convlstm_cell(xl2, ost1, canon["clstm_6_prime.Gates.weight"], canon["clstm_6_prime.Gates.bias"])
    close(h1, oh1, 1e-6, "clstm1"); close(h2, oh2, 1e-6, "clstm2"); close(st2[0], ost2[0], 1e-6, "clstm c")
    save("g2_convlstm", seed_w=SEED_W, x1=xl, x2=xl2, h1=h1, c1=st1[0], h2=h2, c2=st2[0])

    # ---------------- G5: ResidualBlock_noBN (dead code coverage) ----------------
    torch.manual_seed(5)
    rb = REF_MU.ResidualBlock_noBN(64)
    xb = t(g.standard_normal((1, 64, 16, 16), dtype=np.float32))
    with torch.no_grad():
        yb = rb(xb)
        close(yb, O.residual_block_nobn(xb, rb.conv1.weight, rb.conv1.bias, rb.conv2.weight, rb.conv2.bias), 1e-6, "resblock")
    save("g5_resblock", x=xb, y=yb, w1=rb.conv1.weight.detach(), b1=rb.conv1.bias.detach(),
         w2=rb.conv2.weight.detach(), b2=rb.conv2.bias.detach())

    # ---------------- G3: whole net ----------------
    for tag, (n, h, w, seed_x) in (("a", (1, 32, 32, 1234)), ("b", (2, 64, 48, 4321))):
        frames = synthetic_frames(seed_x, n, h, w, 6)
        with torch.no_grad():
            out = ref(*frames)
            oo = O.bin_stage4_forward(frames, canon)
        dmax = max(close(a_, b_, 1e-5, f"whole net {tag}") for a_, b_ in zip(out, oo))
        print(f"whole net {tag}: oracle-vs-ref max diff {dmax:.2e}")
        save(f"g3_net_{tag}", seed_w=SEED_W, seed_x=seed_x, shape=np.array([n, 3, h, w]),
             out=torch.stack(out, 0))

    # ---------------- G3t: one optimize_parameters through the reference wrapper ----------------
    from models import create_model                               # reference models/__init__.py
    opt = {
        "model": "bin", "gpu_ids": None, "is_train": True, "dist": False,
        "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2},
        "path": {"pretrain_model_G": None, "strict_load": True, "models": "/tmp", "training_state": "/tmp"},
        "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                  "lr_G": 1e-4, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR",
                  "lr_steps": [100000], "restarts": None, "restart_weights": None, "lr_gamma": 0.5,
                  "clear_state": False},
    }
    model = create_model(opt)
    model.netG.module.load_state_dict(sd, strict=True)
    gg = rng(7)
    B, Hh, Ww = 1, 32, 32
    batch = {"LQs": t(gg.random((B, 6, 3, Hh, Ww), dtype=np.float32)),
             "GTenh": t(gg.random((B, 6, 3, Hh, Ww), dtype=np.float32)),
             "GTinp": t(gg.random((B, 5, 3, Hh, Ww), dtype=np.float32)), "key": "x"}
    model.feed_data(batch)
    before = {k: v.detach().clone() for k, v in model.netG.module.named_parameters()}
    model.optimize_parameters(1)
    loss = model.loss.detach()
    loss_list = torch.stack([l_.detach() for l_ in model.loss_list])
    named = OrderedDict(model.netG.module.named_parameters())
    # oracle loss
    frames = [batch["LQs"][:, i] for i in range(6)]
    with torch.no_grad():
        oo = O.bin_stage4_forward(frames, canon)
        I = {2 * i + 2: batch["GTinp"][:, i] for i in range(5)}
        I.update({2 * i + 1: batch["GTenh"][:, i] for i in range(6)})
        ol, oll = O.bin_loss(oo, I)
    close(loss, ol, 1e-6, "train loss")
    close(loss_list, torch.stack(oll), 1e-6, "train loss_list")
    sample = ["clstm_4_prime.Gates.weight", "clstm_6_prime_prime_prime.Gates.bias",
              "model.model1_1.SFENet1.weight", "model.model1_1.RDBs.5.convs.2.conv.0.weight",
              "model.model2_1.RDBs.11.LFF.weight", "model.model3_1.GFF.0.weight",
              "model.model4_1.UPNet.0.bias", "model.model4_1.UPNet.2.weight"]
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None))
    tr = dict(loss=loss, loss_list=loss_list, grad_norm=gn.float(), LQs=batch["LQs"], GTenh=batch["GTenh"],
              GTinp=batch["GTinp"])
    for nm in sample:
        tr["grad." + nm] = named[nm].grad.clone()
        tr["after." + nm] = named[nm].detach().clone()
        tr["before." + nm] = before[nm]
    # every parameter's gradient L2 norm (cheap, 540 floats) pins the whole backward
    tr["all_grad_norms"] = torch.stack([(p.grad.double().norm().float() if p.grad is not None else torch.zeros(()))
                                        for p in named.values()])
    save("g3_train", seed_w=SEED_W, names=np.array(list(named.keys())), **tr)

    # ---------------- G4: harness ----------------
    import utils.util as REF_UTIL            # reference utils/util.py (cv2 stubbed)
    tt = t(g.random((3, 24, 40), dtype=np.float32)) * 1.4 - 0.2
    tt2 = t(g.random((3, 24, 40), dtype=np.float32))
    i1 = REF_UTIL.tensor2img(tt.clone()); i2 = REF_UTIL.tensor2img(tt2.clone())
    assert np.array_equal(i1, O.tensor2img(tt))
    ps = REF_UTIL.calculate_psnr(i1, i2)
    assert ps == O.calculate_psnr(i1, i2)
    pads = {}
    for (hh, ww) in ((720, 1280), (256, 256), (352, 640), (256, 448), (64, 96), (128, 128)):
        # reference rule restated inline from test.py:348-366 (test.py itself cannot be imported: cv2, cuda)
        def one(n):
            if n != ((n >> 7) << 7):
                p_ = (((n >> 7) + 1) << 7); a_ = int((p_ - n) / 2); return a_, p_ - n - a_
            return 32, 32
        pads[f"{hh}x{ww}"] = np.array(one(ww) + one(hh))
        assert tuple(pads[f"{hh}x{ww}"]) == O.pad_sizes(hh, ww)
    save("g4_harness", t1=tt, t2=tt2, img1=i1, img2=i2, psnr=np.float64(ps),
         **{"pad." + k: v for k, v in pads.items()})
    lr_goldens()
    print("all fixtures written; oracle == reference on every fixture")


def lr_goldens():
    """G6: LR curves of the reference's restartable schedulers (models/lr_scheduler.py:10-66)."""
    import models.lr_scheduler as REF_LRS
    curves = {}
    for tag, mk in (("multistep", lambda o: REF_LRS.MultiStepLR_Restart(o, [5, 12, 20], restarts=[15], weights=[0.5],
                                                                        gamma=0.5, clear_state=False)),
                    ("cosine", lambda o: REF_LRS.CosineAnnealingLR_Restart(o, [10, 10, 10], restarts=[10, 20],
                                                                           weights=[1, 0.5], eta_min=1e-7))):
        p = torch.nn.Parameter(torch.zeros(1))
        o = torch.optim.Adam([p], lr=1e-4)
        s = mk(o)
        lrs = []
        for _ in range(30):
            o.step()
            s.step()
            lrs.append(o.param_groups[0]["lr"])
        curves[tag] = np.array(lrs, dtype=np.float64)
    save("g6_lr", **curves)


if __name__ == "__main__":
    if "--lr-only" in sys.argv:
        lr_goldens()
    else:
        main()
