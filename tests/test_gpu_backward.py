"""-m gpu: HIP backward (dgrad on the conv kernel, MFMA wgrad, ConvLSTM bwd, whole training step) vs the
reference's own autograd results (golden fixtures) and vs torch autograd of the oracle."""
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
    "k5_lff": ("model1.RDBs.0.LFF", 1), "k6_gff0": ("model1.GFF.0", 1), "k8_up0": ("model1.UPNet.0", 3),
    "k9_up2": ("model1.UPNet.2", 3),
}
TOL = {1: 3e-3, 3: 3e-5}


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("key", sorted(CONVS))
def test_conv_dgrad_wgrad_golden(key, nterms, canon_gpu):
    """dX, dW, db of every live conv shape vs the reference autograd (g1_convs)."""
    from bin_amd import ops
    g = load_golden("g1_convs")
    wname, ks = CONVS[key]
    w = canon_gpu[wname + ".weight"]
    cout, cin = w.shape[0], w.shape[1]
    x = torch.from_numpy(g[key + ".x"]).cuda()
    gy = torch.from_numpy(g[key + ".gy"]).cuda()
    gyp = ops.nchw_to_planes(gy, nterms)
    gx = ops.planes_to_nchw(ops.conv2d_bwd_data(gyp, ops.DgradWeights(w, nterms)), cin)
    assert _rel(gx, torch.from_numpy(g[key + ".gx"]).cuda()) <= TOL[nterms], "dgrad"
    dw, db = ops.conv2d_bwd_weight(ops.nchw_to_planes(x, nterms), gyp, cout, cin, ks, nterms)
    assert _rel(dw, torch.from_numpy(g[key + ".gw"]).cuda()) <= TOL[nterms], "wgrad"
    assert _rel(db, torch.from_numpy(g[key + ".gb"]).cuda()) <= TOL[nterms], "dbias"


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("shape", [(1, 5, 7), (2, 19, 45), (1, 8, 32)])
def test_wgrad_ragged(nterms, shape):
    from bin_amd import ops
    n, h, w = shape
    gen = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn(n, 40, h, w, generator=gen, dtype=torch.float64)
    gy = torch.randn(n, 35, h, w, generator=gen, dtype=torch.float64)
    wt = torch.zeros(35, 40, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(35, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x, wt, b, padding=1).backward(gy)
    dw, db = ops.conv2d_bwd_weight(ops.nchw_to_planes(x.float().cuda(), nterms),
                                   ops.nchw_to_planes(gy.float().cuda(), nterms), 35, 40, 3, nterms)
    assert _rel(dw.cpu().double(), wt.grad) <= TOL[nterms]
    assert _rel(db.cpu().double(), b.grad) <= TOL[nterms]


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("cfg", [(2, 33, 70, 192, 32), (1, 16, 32, 96, 96), (3, 17, 40, 224, 35), (1, 40, 33, 16, 32),
                                 (2, 64, 64, 160, 32), (1, 130, 31, 128, 64)])
def test_wgrad_3x3_shapes(nterms, cfg):
    """the 3x3 weight-gradient kernel (eight waves, two LDS stages) over 1-7 channel pairs, 1-3 output tiles, tiles that hang
    over the right / bottom edge, more workgroups than tiles, several images (reference: autograd of F.conv2d(padding=1),
    RDN.py:141,187-207).  The same shapes validated the rolling-row experiment of the tuning build."""
    from bin_amd import ops
    n, h, w, cin, cout = cfg
    gen = torch.Generator().manual_seed(h * 1000 + w + cin)
    x = torch.randn(n, cin, h, w, generator=gen, dtype=torch.float64)
    gy = torch.randn(n, cout, h, w, generator=gen, dtype=torch.float64)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x, wt, b, padding=1).backward(gy)
    xp, gp = ops.nchw_to_planes(x.float().cuda(), nterms), ops.nchw_to_planes(gy.float().cuda(), nterms)
    dw, db = ops.conv2d_bwd_weight(xp, gp, cout, cin, 3, nterms)
    assert _rel(dw.cpu().double(), wt.grad) <= TOL[nterms]
    assert _rel(db.cpu().double(), b.grad) <= TOL[nterms]
    dw2, db2 = ops.conv2d_bwd_weight(xp, gp, cout, cin, 3, nterms)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "fixed summation order"


@pytest.mark.parametrize("nterms", [3, 1])
@pytest.mark.parametrize("cfg", [(1, 5, 7, 40, 35), (2, 19, 45, 224, 96), (1, 8, 32, 16, 96), (1, 33, 70, 600, 64),
                                 (3, 6, 40, 272, 96), (1, 130, 64, 1152, 96)])
def test_wgrad_1x1_ragged(nterms, cfg):
    """the streaming 1x1 kernel: one / two channel pairs per wave, 1-3 workgroup columns, odd chunk counts, strips that hang
    over the right and bottom edges, more workgroups than strips (reference: autograd of F.conv2d, RDN.py:141,162)."""
    from bin_amd import ops
    n, h, w, cin, cout = cfg
    gen = torch.Generator().manual_seed(h * 100 + w + cin)
    x = torch.randn(n, cin, h, w, generator=gen, dtype=torch.float64)
    gy = torch.randn(n, cout, h, w, generator=gen, dtype=torch.float64)
    wt = torch.zeros(cout, cin, 1, 1, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x, wt, b).backward(gy)
    dw, db = ops.conv2d_bwd_weight(ops.nchw_to_planes(x.float().cuda(), nterms),
                                   ops.nchw_to_planes(gy.float().cuda(), nterms), cout, cin, 1, nterms)
    assert _rel(dw.cpu().double(), wt.grad) <= TOL[nterms]
    assert _rel(db.cpu().double(), b.grad) <= TOL[nterms]
    dw2, db2 = ops.conv2d_bwd_weight(ops.nchw_to_planes(x.float().cuda(), nterms),
                                     ops.nchw_to_planes(gy.float().cuda(), nterms), cout, cin, 1, nterms)
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "fixed summation order"


def test_convlstm_backward_vs_autograd(canon_cpu, canon_gpu):
    from bin_amd.autograd import convlstm_apply
    from oracle import rdn_oracle as O
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 18, 70, generator=gen)
    c0, h0 = torch.rand(2, 3, 18, 70, generator=gen), torch.rand(2, 3, 18, 70, generator=gen)
    for with_state in (False, True):
        xr = x.clone().requires_grad_(True)
        w = canon_cpu["clstm_6_prime.Gates.weight"].clone().requires_grad_(True)
        b = (canon_cpu["clstm_6_prime.Gates.bias"] + 0.1).clone().requires_grad_(True)
        st = [c0.clone().requires_grad_(True), h0.clone().requires_grad_(True)] if with_state else None
        h, (c, _) = O.convlstm_cell(xr, st, w, b)
        gh, gc = torch.rand_like(h), torch.rand_like(c)
        (h * gh).sum().backward(retain_graph=True) if False else ((h * gh).sum() + (c * gc).sum()).backward()
        xg = x.cuda().requires_grad_(True)
        wg = w.detach().cuda().requires_grad_(True)
        bg = b.detach().cuda().requires_grad_(True)
        stg = [c0.cuda().requires_grad_(True), h0.cuda().requires_grad_(True)] if with_state else None
        hh, (cc, _) = convlstm_apply(xg, stg, wg, bg, 1.0)
        ((hh * gh.cuda()).sum() + (cc * gc.cuda()).sum()).backward()
        assert _rel(hh.detach().cpu(), h.detach()) <= 1e-5
        assert _rel(xg.grad.cpu(), xr.grad) <= 2e-5
        assert _rel(wg.grad.cpu(), w.grad) <= 2e-4
        assert _rel(bg.grad.cpu(), b.grad) <= 2e-4
        if with_state:
            assert _rel(stg[0].grad.cpu(), st[0].grad) <= 2e-5
            assert _rel(stg[1].grad.cpu(), st[1].grad) <= 2e-5


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
