"""CPU: host logic of the drop-in boundary — library symbols, module/state_dict contract, wrapper API
(bin_model) driven by an injected CPU generator, checkpoint IO, LR schedules, harness helpers."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import load_golden, REPO


def T(a):
    return torch.from_numpy(np.asarray(a))


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_header_symbol():
    from bin_amd import _lib
    hdr = open(os.path.join(REPO, "include", "binhip.h")).read()
    declared = set(re.findall(r"\b(binhip_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"libbinhip.so does not export {sym}"
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    assert _lib.lib().binhip_version() >= 400
    # the product library has no process-global switches (SURVEY §8b "no globals except immutable tables"): the
    # tuning / ablation setters exist only in BINHIP_TUNING side builds and are not declared in the public header
    for sym in _lib._TUNING_SIGNATURES:
        assert not hasattr(lib, sym), f"product libbinhip.so must not export {sym}"
        assert sym not in declared
    assert "binhip_profile_begin" not in declared and "set_variant" not in hdr


def test_library_exports_nothing_but_the_abi():
    """-fvisibility=hidden + a version script generated from the header: the dynamic symbol table is EXACTLY the entry
    points of include/binhip.h — no bh_* launch helpers, no kernel handles / device stubs."""
    import subprocess
    from bin_amd import _lib, build
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    dyn = {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.strip()}
    assert dyn == set(build.abi_symbols()) == set(_lib.exported_symbols()), dyn ^ set(build.abi_symbols())


def test_product_library_contains_only_dispatched_weight_gradient_kernels():
    """The experimental weight-gradient kernels of rounds 1-3 (lean single-stage, wave = gY row, rolling rows) live in
    tools/experiments/wgrad_experiments.inc and are compiled by `--tuning` side builds only: neither their host stubs nor
    their device symbols are in the product .so, and the product source stays under 1 500 lines."""
    from bin_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"wgrad3x3_db_kernel", b"wgrad3x3_roll", b"wgrad_mfma_sb_kernel"):
        assert name not in blob, name
    for name in (b"wgrad3x3_xrow_kernel", b"wgrad1x1_kernel", b"wgrad_mfma_kernel", b"wgrad_reduce_kernel"):
        assert name in blob, name
    src = os.path.join(REPO, "bin_amd", "csrc", "binhip_wgrad.hip")
    assert sum(1 for _ in open(src)) < 1500
    assert os.path.exists(os.path.join(REPO, "tools", "experiments", "wgrad_experiments.inc"))


def test_bare_rdn_subnetwork_offers_the_direct_gradient_context():
    """A bare RDN sub-network can be a wrapper's netG (VideoBaseModel's netG= injection): `direct_param_grads` is a METHOD on
    it as on the whole network (round 3 shadowed it with a bool attribute there), and the wrapper's helper flips its flag."""
    from bin_amd.models.archs.RDN import RDN_residual_interp_2_input, bin_stage4_lstm
    from bin_amd.models.base_model import _direct_param_grads
    for net in (RDN_residual_interp_2_input(96, 2, 4, 32), bin_stage4_lstm()):
        mods = [net] if not hasattr(net, "rdn_modules") else net.rdn_modules()
        assert not any(m._direct_grads for m in mods)
        with _direct_param_grads(net):
            assert all(m._direct_grads for m in mods)
        assert not any(m._direct_grads for m in mods)


def test_header_version_and_export_count_match_the_library():
    """include/binhip.h carries BINHIP_VERSION (what binhip_version() returns) and BINHIP_ABI_EXPORTS (the number of
    BINHIP_API declarations): a binder can check both at compile / load time."""
    import re
    from bin_amd import _lib, build
    hdr = open(os.path.join(REPO, "include", "binhip.h")).read()
    ver = int(re.search(r"#define\s+BINHIP_VERSION\s+(\d+)", hdr).group(1))
    n = int(re.search(r"#define\s+BINHIP_ABI_EXPORTS\s+(\d+)", hdr).group(1))
    assert _lib.lib().binhip_version() == ver
    assert n == len(build.abi_symbols()) == len(set(build.abi_symbols()))
    assert "BINHIP_VERSION" not in open(os.path.join(REPO, "bin_amd", "csrc", "binhip_internal.h")).read()


def test_integration_doc_names_exist_in_the_header():
    """Every BINHIP_* constant and binhip_* entry point INTEGRATION.md mentions is declared in include/binhip.h (round 3's
    document promised an error code that did not exist)."""
    import re
    from bin_amd import build
    hdr = open(os.path.join(REPO, "include", "binhip.h")).read()
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    macros = set(re.findall(r"#define\s+(BINHIP_\w+)", hdr)) | set(re.findall(r"\b(BINHIP_\w+)\s*=", hdr))
    macros |= set(re.findall(r"\b(BINHIP_\w+)\b", hdr))                      # enum members, too
    side_build = {"BINHIP_TUNING", "BINHIP_E_"}                                # the -D switch of tools/ builds; a prefix in prose
    for name in set(re.findall(r"\b(BINHIP_\w+)", doc)) - side_build:
        assert name in macros, f"INTEGRATION.md mentions {name}, which include/binhip.h does not define"
    entries = set(build.abi_symbols())
    for name in set(re.findall(r"\b(binhip_[a-z0-9_]+)\b", doc)):
        if name.endswith(("_fwd/bwd", "_")) or name in entries:
            continue
        # prose shorthands like `binhip_pixel_loss_fwd/bwd` are split by the regex into existing prefixes
        assert any(e.startswith(name) for e in entries), f"INTEGRATION.md mentions {name}(), not in the ABI"


def test_integration_doc_states_the_version_and_export_count_once_and_right():
    """INTEGRATION.md carried 400 / 43 in one section and 500 / 45 in another: every number the document puts next to
    BINHIP_VERSION / BINHIP_ABI_EXPORTS / "entry points" must be the header's, and the header's count must be the ABI's."""
    import re
    from bin_amd import build
    hdr = open(os.path.join(REPO, "include", "binhip.h")).read()
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    ver = int(re.search(r"#define\s+BINHIP_VERSION\s+(\d+)", hdr).group(1))
    n = int(re.search(r"#define\s+BINHIP_ABI_EXPORTS\s+(\d+)", hdr).group(1))
    assert n == len(build.abi_symbols())
    near_ver = re.findall(r"BINHIP_VERSION`?[^.\n]{0,60}?\*{0,2}(\d{3})\*{0,2}", doc)
    near_cnt = re.findall(r"BINHIP_ABI_EXPORTS`?[^.\n]{0,80}?\*{0,2}(\d{2})\*{0,2}", doc) + re.findall(r"(\d{2}) entry points", doc)
    assert near_ver and all(int(v) == ver for v in near_ver), (near_ver, ver)
    assert near_cnt and all(int(c) == n for c in near_cnt), (near_cnt, n)
    design = open(os.path.join(REPO, "DESIGN.md")).read()
    assert all(int(v) == ver for v in re.findall(r"BINHIP_VERSION`? (\d{3})", design))
    assert all(int(c) == n for c in re.findall(r"(\d{2}) entry points", design))


def test_several_gpu_ids_without_dist_raise_with_the_launch_recipe(tmp_path):
    """The reference's default multi-GPU mode is nn.DataParallel over gpu_ids (bin_model.py:41-42; the shipped yml has gpu_ids:
    [2, 3], dist: false).  One process per GPU cannot honour that silently on one device (VERDICT r05 missing 4)."""
    from bin_amd.models.Video_base_model import VideoBaseModel
    from bin_amd.models.bin_model import bin_model
    opt = _opt(tmp_path)
    opt["gpu_ids"] = [2, 3]
    for cls in (bin_model, VideoBaseModel):
        with pytest.raises(RuntimeError, match=r"torch\.distributed\.run --nnodes=1 --nproc-per-node=2"):
            cls(opt, netG=torch.nn.Conv2d(3, 3, 1))
    from bin_amd.models.base_model import BaseModel
    opt["gpu_ids"] = [0]
    assert BaseModel(opt).device.type == "cuda"          # one id: fine; several ids under dist: each rank has its own
    opt["gpu_ids"], opt["dist"] = [0, 1], True
    BaseModel(opt)


def test_module_util_mirror_is_importable_and_shaped_like_the_reference():
    """reference models/module_util.py:7-52 — names, constructor arguments, state_dict keys; the forward has no CPU path."""
    from bin_amd.models import module_util as MU
    blk = MU.ResidualBlock_noBN(nf=16)
    assert sorted(blk.state_dict()) == ["conv1.bias", "conv1.weight", "conv2.bias", "conv2.weight"]
    assert blk.conv1.weight.shape == (16, 16, 3, 3) and float(blk.conv2.bias.abs().max()) == 0.0
    assert float(blk.conv1.weight.std()) < 0.05           # kaiming-normal x 0.1 (module_util.py:47)
    seq = MU.make_layer(lambda: MU.ResidualBlock_noBN(8), 3)
    assert len(seq) == 3 and len({id(m) for m in seq}) == 3
    lin = torch.nn.Linear(4, 4)
    bn = torch.nn.BatchNorm2d(4)
    MU.initialize_weights([lin, bn], scale=2.0)
    assert float(lin.bias.abs().max()) == 0.0 and float(bn.weight.min()) == 1.0
    with pytest.raises(RuntimeError, match="no CPU path"):
        blk(torch.zeros(1, 16, 4, 4))


def test_bench_step_spread_and_cycle_fields():
    import bench
    sp = bench.step_spread([10.0, 10.2, 9.9, 31.0, 10.1])
    assert sp == {"ms_min": 9.9, "ms_median": 10.1, "ms_max": 31.0, "max_over_median": round(31.0 / 10.1, 4), "steps": 5}
    assert bench.step_spread([]) is None and bench.step_spread([2.0, 4.0])["ms_median"] == 3.0
    cf = bench.clock_fields(70.0, {"xcd_clock_mhz": {"mean": 1800.0}, "clock_mhz": {"mean": 1850.0}})
    assert cf["cycles_per_step_M"] == 126.0 and cf["xcd_clock_mhz_mean"] == 1800.0 and "per-XCD" in cf["cycles_clock"]
    cf = bench.clock_fields(70.0, {"clock_mhz": {"mean": 2000.0}})
    assert cf["cycles_per_step_M"] == 140.0 and "XCD 0" in cf["cycles_clock"]
    assert bench.clock_fields(70.0, None)["cycles_per_step_M"] is None


def test_trained_like_weights_have_the_advertised_distribution():
    from bin_amd.weights import canonical_weights, state_dict_from_canonical, reference_state_dict, trained_like_weights
    w = trained_like_weights(0)
    base = canonical_weights(0)
    assert list(w) == list(base) and all(w[k].shape == base[k].shape and w[k].dtype == np.float32 for k in w)
    conv = np.concatenate([np.abs(v).ravel() for k, v in w.items() if k.endswith(".weight") and ".Gates." not in k])
    nz = conv[conv > 0]
    assert 0.29 < 1 - nz.size / conv.size < 0.31
    assert nz.min() < 1e-4 and nz.max() > 1.0 and (nz < 0.125).mean() > 0.9
    assert all(float(np.abs(v).max()) == 0.0 for k, v in w.items() if k.endswith(".bias") and ".Gates." not in k)
    a, b = state_dict_from_canonical(base), reference_state_dict(0)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)


def test_library_host_queries():
    from bin_amd import _lib
    lib = _lib.lib()
    assert lib.binhip_conv_cout_block(3, 32, 1) == 32
    assert lib.binhip_conv_cout_block(3, 256, 1) == 128 and lib.binhip_conv_cout_block(3, 256, 3) == 32
    assert lib.binhip_conv_cout_block(3, 96, 3) == 32 and lib.binhip_conv_cout_block(3, 96, 1) == 96
    assert lib.binhip_weights_bytes(32, 6, 3) == 32 * 6 * 9 * 32
    assert lib.binhip_rdn_workspace_bytes(1, 33, 32, 2, 1, None) == 0          # odd height rejected
    assert lib.binhip_rdn_workspace_bytes(1, 64, 64, 4, 1, None) == 0          # 4 inputs do not exist
    b1, b3 = lib.binhip_rdn_workspace_bytes(1, 768, 1344, 2, 1, None), lib.binhip_rdn_workspace_bytes(1, 768, 1344, 2, 3, None)
    assert 0 < b1 < b3 < (8 << 30)
    # network shapes (BinRdnShape): NULL and all-zero mean bin_stage4's (96, 12, 4, 32); others are sized accordingly
    def shp(*v):
        s_ = _lib.BinRdnShape()
        s_.G0, s_.D, s_.C, s_.G = v
        return ctypes.byref(s_)
    assert lib.binhip_rdn_workspace_bytes(1, 768, 1344, 2, 1, shp(0, 0, 0, 0)) == b1
    assert lib.binhip_rdn_workspace_bytes(1, 768, 1344, 2, 1, shp(96, 12, 4, 32)) == b1
    small = lib.binhip_rdn_workspace_bytes(1, 768, 1344, 2, 1, shp(64, 6, 4, 32))
    assert 0 < small < b1
    for bad in ((48, 6, 4, 32), (64, 6, 4, 16), (64, 6, 8, 32), (64, 21, 4, 32), (64, 0, 4, 32)):
        assert lib.binhip_rdn_workspace_bytes(1, 64, 64, 2, 1, shp(*bad)) == 0, bad
        assert lib.binhip_rdn_backward_workspace_bytes(1, 64, 64, 2, 1, shp(*bad)) == 0, bad
    from bin_amd.rdn_plan import check_shape, layer_names
    assert len(layer_names()) == 66 and len(layer_names((64, 6, 4, 32))) == 36 and len(layer_names((32, 2, 3, 64))) == 14
    with pytest.raises(NotImplementedError):
        check_shape((48, 6, 4, 32))


def test_abi_error_codes_without_a_gpu():
    """Argument / shape validation happens before any HIP call, so the error behaviour of the C ABI (0 ok, < 0
    BINHIP_E_*, never an exception) can be pinned on a GPU-less box: every call below must return, not launch."""
    from bin_amd import _lib as L
    lib = L.lib()
    E_ARG, E_SHAPE, E_WS = -1, -2, -3
    null = ctypes.c_void_p(0)
    d = L.BinConvDesc()
    d.N, d.H, d.W, d.ksize, d.cin_chunks, d.cout, d.cout_pad, d.nterms, d.epilogue = 1, 8, 8, 3, 2, 32, 32, 1, 0
    # null pointers
    assert lib.binhip_conv2d_fwd(None, *([null] * 10), None, null) == E_ARG
    assert lib.binhip_conv2d_fwd(ctypes.byref(d), *([null] * 10), None, null) == E_ARG
    assert lib.binhip_rdn_forward(None, None, null, null, 0, null) == E_ARG
    assert lib.binhip_weights_relayout(null, null, 32, 32, 3, 32, 2, 32, 0, null, null, null, null) == E_ARG
    four = (ctypes.c_void_p * 4)()
    assert lib.binhip_weights_relayout_rdb_gather(four, 7, 32, null, null, null, null) == E_ARG       # bad group
    # shapes the kernels do not have
    one = ctypes.c_void_p(16)                      # any non-null value: validation must fail before it is touched
    assert lib.binhip_weights_relayout(one, null, 32, 32, 3, 48, 2, 32, 0, one, null, one, null) == E_SHAPE   # cout_pad % 32
    assert lib.binhip_weights_relayout(one, null, 40, 32, 3, 32, 2, 32, 0, one, null, one, null) == E_SHAPE   # cout > cout_pad
    assert lib.binhip_weights_relayout_dgrad(one, 32, 96, 3, 96, 2, 40, 0, one, null, one, null) == E_SHAPE   # bad cout_block
    assert lib.binhip_dgrad_rows_pad(1, 224) == 224 and lib.binhip_dgrad_rows_pad(3, 96) == 96
    assert lib.binhip_conv_cout_block(1, 224, 3) == 224 and lib.binhip_conv_cout_block(1, 96, 1) == 96
    plan = L.BinRdnPlan()
    plan.N, plan.H, plan.W, plan.n_inputs, plan.nterms = 1, 33, 32, 2, 1
    arr = (ctypes.c_void_p * 2)(16, 16)
    assert lib.binhip_rdn_forward(ctypes.byref(plan), arr, one, one, 1 << 20, null) in (E_SHAPE, E_ARG)      # odd height
    assert lib.binhip_rdn_backward_workspace_bytes(1, 64, 64, 4, 1, None) == 0
    assert lib.binhip_wgrad_workspace_bytes(3, 0, 8, 8, 2, 32) == 0
    h = ctypes.c_void_p(0)
    assert lib.binhip_profiler_create(3, 32, 0, 0, ctypes.byref(h)) == E_ARG and not h.value     # max_launches <= 0
    assert lib.binhip_profiler_create(3, 32, 0, 4, None) == E_ARG
    assert lib.binhip_profiler_read(None, None, None) == E_ARG
    assert E_WS == -3


def test_product_has_no_cpu_path():
    from bin_amd import ops
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.models.loss import CharbonnierLoss
    with pytest.raises(RuntimeError):
        bin_stage4_lstm()(*[torch.rand(1, 3, 32, 32) for _ in range(6)])
    with pytest.raises(RuntimeError):
        ops.nchw_to_planes(torch.rand(1, 3, 8, 8))
    with pytest.raises(RuntimeError):
        CharbonnierLoss()(torch.rand(4), torch.rand(4))


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "bin_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(root, f)


# ------------------------------------------------------------------ module contract
def test_state_dict_contract():
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.weights import reference_state_dict
    net = bin_stage4_lstm()
    sd = reference_state_dict(0)
    assert len(sd) == 1332 and list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in net.parameters()) == 11441668
    assert len(list(net.named_parameters())) == 540
    assert net.model.model1_2 is net.model.model1_1 and net.model.model3_2 is net.model.model3_1
    assert net.model.model4_1 is not net.model.model3_1
    # ConvLSTM init (RDN.py:26-38): zero bias, xavier-uniform bound
    fresh = bin_stage4_lstm()
    assert float(fresh.clstm_4_prime.Gates.bias.abs().max()) == 0.0
    assert float(fresh.clstm_4_prime.Gates.weight.abs().max()) <= (6.0 / (54 + 108)) ** 0.5


def test_define_G_and_create_model_errors():
    from bin_amd.models import create_model, networks
    with pytest.raises(NotImplementedError):
        networks.define_G({"network_G": {"which_model_G": "nope"}})
    with pytest.raises(NotImplementedError):
        create_model({"model": "sr"})
    net = networks.define_G({"network_G": {"which_model_G": "bin_stage4", "precision": "f16"}})
    assert net.precision == "f16" and net.model.model2_1.precision == "f16"
    assert not any(m.allow_f16_training for m in net.rdn_modules())                 # f16 TRAINING stays gated by default ...
    net = networks.define_G({"network_G": {"which_model_G": "bin_stage4", "precision": "f16", "allow_f16_training": True}})
    assert all(m.allow_f16_training for m in net.rdn_modules())                     # ... and opens only on an explicit option


# ------------------------------------------------------------------ wrapper
def _opt(tmp, is_train=True, dist=False):
    return {"model": "bin", "gpu_ids": None, "is_train": is_train, "dist": dist,
            "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2},
            "path": {"pretrain_model_G": None, "strict_load": True, "models": str(tmp), "training_state": str(tmp)},
            "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                      "lr_G": 1e-4, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000],
                      "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}


class _Cb(torch.nn.Module):
    def forward(self, x, y):
        from oracle import rdn_oracle as O
        return O.charbonnier(x, y)


def _cpu_model(tmp, **kw):
    from bin_amd.models.bin_model import bin_model
    from bin_amd.weights import reference_state_dict
    from oracle_net import OracleNet
    net = OracleNet()
    net.load_state_dict(reference_state_dict(0), strict=True)
    return bin_model(_opt(tmp, **kw), netG=net, cri_pix=_Cb())


def test_training_step_matches_reference_golden(tmp_path):
    """One optimize_parameters() through OUR wrapper (injected CPU generator) reproduces the reference
    wrapper's loss, 14-entry loss_list, all 540 gradient norms, sampled gradients and post-Adam values."""
    g = load_golden("g3_train")
    m = _cpu_model(tmp_path)
    m.feed_data({"LQs": T(g["LQs"]), "GTenh": T(g["GTenh"]), "GTinp": T(g["GTinp"]), "key": "x"})
    assert (m.batch, m.channel, m.height, m.width) == (1, 3, 32, 32)
    m.optimize_parameters(1)
    assert abs(float(m.loss) - float(g["loss"])) <= 1e-6
    assert len(m.loss_list) == 14
    assert float((torch.stack([l.detach() for l in m.loss_list]) - T(g["loss_list"])).abs().max()) <= 1e-6
    named = dict(m.netG.module.named_parameters())
    names = [str(n) for n in g["names"]]
    assert names == list(named.keys())
    norms = torch.stack([named[n].grad.double().norm().float() if named[n].grad is not None else torch.zeros(())
                         for n in names])
    ref = T(g["all_grad_norms"])
    assert float(((norms - ref).abs() / (ref.abs() + 1e-8)).max()) <= 2e-3
    for key in g.files:
        if key.startswith("grad."):
            n = key[5:]
            assert float((named[n].grad - T(g[key])).abs().max()) <= 1e-6 + 1e-3 * float(np.abs(g[key]).max())
            assert float((named[n].detach() - T(g["after." + n])).abs().max()) <= 2e-6
    # recurrent half of the ConvLSTM gates never receives gradient in the 2-window topology
    assert float(named["clstm_4_prime.Gates.weight"].grad[:, 3:].abs().max()) == 0.0


def test_three_training_steps_match_reference_golden(tmp_path):
    """Three consecutive optimize_parameters() through OUR wrapper (CPU generator) vs the reference wrapper
    (g9_train_steps, tests/golden/make_golden_steps.py): the loss before each update and sampled parameters after the third —
    steps 2 and 3 see the weights Adam changed."""
    g = load_golden("g9_train_steps")
    m = _cpu_model(tmp_path)
    batch = {"LQs": T(g["LQs"]), "GTenh": T(g["GTenh"]), "GTinp": T(g["GTinp"]), "key": "x"}
    got = []
    for step in (1, 2, 3):
        m.feed_data(batch)
        m.optimize_parameters(step)
        got.append(float(m.loss))
    assert np.abs(np.array(got) - g["losses"]).max() <= 2e-6, (got, g["losses"])
    assert abs(got[1] - got[0]) > 1e-3, "the second step must see updated weights"
    named = dict(m.netG.module.named_parameters())
    for key in g.files:
        if key.startswith("after3."):
            assert float((named[key[7:]].detach() - T(g[key])).abs().max()) <= 5e-6, key


def _crit(kind):
    """torch's own criteria, as the reference builds them (bin_model.py:52-60) — for the CPU run of OUR wrapper."""
    return {"cb": _Cb(), "l1": torch.nn.L1Loss(reduction="sum"), "l2": torch.nn.MSELoss(reduction="sum")}[kind]


@pytest.mark.parametrize("tag,version,crit,weight", __import__("loss_variants").VARIANTS)
def test_loss_variants_match_reference_wrapper(tmp_path, tag, version, crit, weight):
    """The other branches of the reference's loss (g11_loss_variants, tests/golden/make_golden_loss_variants.py): no cycle
    terms unless version == 2, the 'l1' / 'l2' sum criteria, pixel_weight — one optimize_parameters() of OUR wrapper (CPU
    generator) against the reference wrapper's."""
    from bin_amd.models.bin_model import bin_model
    from bin_amd.weights import reference_state_dict
    from oracle_net import OracleNet
    g = load_golden("g11_loss_variants")
    opt = _opt(tmp_path)
    opt["network_G"]["version"] = version
    opt["train"]["pixel_criterion"], opt["train"]["pixel_weight"] = crit, weight
    net = OracleNet()
    net.load_state_dict(reference_state_dict(0), strict=True)
    m = bin_model(opt, netG=net, cri_pix=_crit(crit))
    m.feed_data({"LQs": T(g["LQs"]), "GTenh": T(g["GTenh"]), "GTinp": T(g["GTinp"]), "key": "x"})
    m.optimize_parameters(1)
    ref = float(g[tag + ".loss"])
    assert abs(float(m.loss) - ref) <= 2e-6 * max(1.0, abs(ref)), (float(m.loss), ref)
    assert len(m.loss_list) == 14
    ll = np.array([float(l) for l in m.loss_list])
    assert np.abs(ll - g[tag + ".loss_list"]).max() <= 2e-6 * max(1.0, np.abs(g[tag + ".loss_list"]).max())
    named = dict(m.netG.module.named_parameters())
    assert [str(n) for n in g["names"]] == list(named.keys())
    norms = np.array([float(p.grad.double().norm()) if p.grad is not None else 0.0 for p in named.values()])
    refn = g[tag + ".grad_norms"]
    assert (np.abs(norms - refn) / (np.abs(refn) + 1e-8 * refn.max())).max() <= 2e-3
    for key in g.files:
        if key.startswith(tag + ".after."):
            assert float((named[key[len(tag) + 7:]].detach() - T(g[key])).abs().max()) <= 2e-6, key


def test_ft_tsa_only_freezes_group_zero_like_the_reference(tmp_path):
    """train.ft_tsa_only = 3 (bin_model.py:66-87,131-132): the optimizer gets the reference's TWO parameter groups (all 540
    tensors, then the empty 'tsa_fusion' group — the layout `.state` files carry), steps 1 and 2 run with group 0's rate at
    zero, step 3 trains with Adam moments that already saw three gradients.  Fixture g11_loss_variants ('ft.*')."""
    from bin_amd.models.bin_model import bin_model
    from bin_amd.weights import reference_state_dict
    from oracle_net import OracleNet
    g = load_golden("g11_loss_variants")
    opt = _opt(tmp_path)
    opt["train"]["ft_tsa_only"] = 3
    net = OracleNet()
    net.load_state_dict(reference_state_dict(0), strict=True)
    m = bin_model(opt, netG=net, cri_pix=_Cb())
    sd = m.optimizer_G.state_dict()
    assert [len(gp["params"]) for gp in sd["param_groups"]] == list(g["ft.group_sizes"]) == [540, 0]
    named = dict(m.netG.module.named_parameters())
    probe = "model.model4_1.UPNet.2.weight"
    before = named[probe].detach().clone()
    batch = {"LQs": T(g["LQs"]), "GTenh": T(g["GTenh"]), "GTinp": T(g["GTinp"]), "key": "x"}
    for step in (1, 2, 3):
        if step == 3:
            for grp in m.optimizer_G.param_groups:
                grp["lr"] = opt["train"]["lr_G"]
        m.feed_data(batch)
        m.optimize_parameters(step)
        moved = float((named[probe].detach() - before).abs().max())
        assert (moved == 0.0) == (step < 3) and abs(moved - float(g["ft.moved"][step - 1])) <= 1e-6
    assert abs(float(m.loss) - float(g["ft.loss3"])) <= 2e-6
    assert float((named[probe].detach() - T(g["ft.after3"])).abs().max()) <= 2e-6


def test_wrapper_api_surface(tmp_path):
    m = _cpu_model(tmp_path)
    for name in ("feed_data", "test_set_input", "test", "forward", "test_forward", "optimize_parameters", "get_loss",
                 "get_info", "get_current_log", "get_current_visuals", "save", "load", "reset_state",
                 "train_AverageMeter", "train_AverageMeter_update", "val_AverageMeter_para", "update_learning_rate",
                 "save_network", "load_network", "save_training_state", "resume_training", "get_current_learning_rate",
                 "compute_current_psnr_ssim", "print_network", "get_lr", "set_params_lr_zero", "test_sharp_forward"):
        assert callable(getattr(m, name)), name
    frames = [torch.rand(1, 3, 32, 32) for _ in range(6)]
    m.test_set_input(frames + [torch.tensor([0])])
    out = m.test()
    assert len(out) == 14 and out[13].shape == (1, 3, 32, 32) and not out[13].requires_grad
    m.test_forward()
    assert len(m.Ft_p) == 14
    num, gts, lqs = 14, None, None
    g = load_golden("g3_train")
    m.feed_data({"LQs": T(g["LQs"]), "GTenh": T(g["GTenh"]), "GTinp": T(g["GTinp"])})
    num, gt_list, lq_list = m.get_info(mode=2)
    assert num == 14 and len(lq_list) == 6
    order = [2, 4, 6, 8, 3, 5, 7, 4, 6, 5, 10, 9, 8, 7]           # bin_model.py:530-534
    for gt, k in zip(gt_list, order):
        assert gt is getattr(m, f"I{k}")
    m.test()
    psnr, ssim = m.compute_current_psnr_ssim()
    assert len(psnr) == 14 and all(np.isfinite(psnr)) and all(-1 <= s <= 1 for s in ssim)
    m.train_AverageMeter(); m.get_loss(); m.train_AverageMeter_update()
    inst, avg = m.get_current_log("train")
    assert set(inst) == {str(i) for i in range(14)} and "Al" in avg


def test_checkpoint_roundtrip(tmp_path):
    m = _cpu_model(tmp_path)
    m.save("7")
    path = os.path.join(str(tmp_path), "7_G.pth")
    sd = torch.load(path)
    assert len(sd) == 1332 and all(v.device.type == "cpu" for v in sd.values())
    with torch.no_grad():
        for p in m.netG.parameters():
            p.add_(1.0)
    m.load_network(path, m.netG, strict=True)
    from bin_amd.weights import reference_state_dict
    ref = reference_state_dict(0)
    for k, v in m.netG.module.state_dict().items():
        assert torch.equal(v, ref[k])
    # DataParallel-style 'module.' prefixes are stripped (base_model.py:94-96)
    torch.save({"module." + k: v for k, v in sd.items()}, path)
    m.load_network(path, m.netG, strict=True)
    m.save_training_state(3, 11)
    st = torch.load(os.path.join(str(tmp_path), "11.state"))
    assert st["epoch"] == 3 and st["iter"] == 11 and len(st["optimizers"]) == 1 and len(st["schedulers"]) == 1
    m.resume_training(st)


def test_lr_schedulers_match_reference():
    from bin_amd.models import lr_scheduler as LRS
    g = load_golden("g6_lr")
    for tag, mk in (("multistep", lambda o: LRS.MultiStepLR_Restart(o, [5, 12, 20], restarts=[15], weights=[0.5],
                                                                    gamma=0.5, clear_state=False)),
                    ("cosine", lambda o: LRS.CosineAnnealingLR_Restart(o, [10, 10, 10], restarts=[10, 20],
                                                                       weights=[1, 0.5], eta_min=1e-7))):
        p = torch.nn.Parameter(torch.zeros(1))
        o = torch.optim.Adam([p], lr=1e-4)
        s = mk(o)
        lrs = []
        for _ in range(30):
            o.step(); s.step(); lrs.append(o.param_groups[0]["lr"])
        assert np.allclose(np.array(lrs), g[tag], rtol=1e-12, atol=0), tag


def test_lr_schedulers_in_the_training_loop_match_reference():
    """g6b_lr (tests/golden/make_golden_lr.py): two parameter groups, warm-up overriding the rates between scheduler steps,
    gamma = 0.1, restarts listed out of order, cleared optimizer state, and a state_dict round trip mid-curve."""
    from bin_amd.models import lr_scheduler as LRS
    import lr_cases
    g = load_golden("g6b_lr")
    for tag, case in lr_cases.CASES.items():
        kind, kw, warm, rescale = lr_cases.unpack(case)     # (cosine_bottom_rescale: rates rescaled ON the cosine's floor)
        for resume in (False, True):
            got = lr_cases.drive(LRS, kind, kw, warm, resume, rescale)
            assert np.allclose(got, g[tag], rtol=1e-12, atol=0), (tag, resume, np.abs(got / g[tag] - 1).max())


def test_lr_closed_form_equals_the_stepped_curve():
    """Without outside interference the stepped rates ARE the closed form initial_lr * shape(e) (floor-corrected)."""
    from bin_amd.models import lr_scheduler as LRS
    for mk in (lambda o: LRS.MultiStepLR_Restart(o, [5, 12, 20, 20], restarts=[15], weights=[0.5], gamma=0.1),
               lambda o: LRS.CosineAnnealingLR_Restart(o, [10, 7, 10], restarts=[10, 24], weights=[1, 0.5], eta_min=1e-7)):
        p = torch.nn.Parameter(torch.zeros(1))
        o = torch.optim.Adam([p], lr=1e-4)
        s = mk(o)
        for e in range(1, 35):          # (past the floor of a WEIGHTED cycle the reference re-enters from the unweighted rate)
            o.step(); s.step()
            cyc, start, wgt = s._cycle(e)
            want = s.floor + (1e-4 * wgt - s.floor) * s.curve(cyc, e - start, start)
            assert abs(o.param_groups[0]["lr"] / want - 1) < 1e-9, (type(s).__name__, e)


def test_warmup_lr(tmp_path):
    m = _cpu_model(tmp_path)
    m.optimizer_G.step()
    m.update_learning_rate(2, warmup_iter=10)
    assert abs(m.get_current_learning_rate()[0] - 1e-4 * 2 / 10) < 1e-12


def test_util_helpers_match_reference():
    from bin_amd.utils import util
    g = load_golden("g4_harness")
    assert np.array_equal(util.tensor2img(T(g["t1"])), g["img1"])
    assert util.calculate_psnr(g["img1"], g["img2"]) == float(g["psnr"])
    for key in g.files:
        if key.startswith("pad."):
            h, w = [int(v) for v in key[4:].split("x")]
            assert util.pad_sizes(h, w) == tuple(int(v) for v in g[key])
    x = torch.arange(12.0).view(1, 1, 3, 4)
    y = util.replicate_pad(x, (1, 2, 1, 0))
    assert y.shape == (1, 1, 4, 7) and float(y[0, 0, 0, 0]) == 0.0 and float(y[0, 0, 3, 6]) == 11.0
    s = util.calculate_ssim(g["img1"], g["img1"])
    assert abs(s - 1.0) < 1e-12


def test_entry_points_parse_their_arguments():
    """bench.py / bin_amd.test / bin_amd.train import and build their CLIs on a GPU-less box (guards against syntax and
    import errors in files the CPU suite does not otherwise execute)."""
    import subprocess
    import sys
    for cmd in (["bench.py", "--help"], ["-m", "bin_amd.test", "--help"], ["-m", "bin_amd.train", "--help"]):
        r = subprocess.run([sys.executable] + cmd, cwd=REPO, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (cmd, r.stderr[-500:])
        assert "usage" in r.stdout.lower()


def test_harness_rules_properties():
    """Property checks (hypothesis) of the host rules taken from test.py: padding (348-366), window frames (257-261),
    window sharding."""
    from hypothesis import given, settings, strategies as st
    from bin_amd import harness
    from bin_amd.utils import util

    @settings(max_examples=200, deadline=None)
    @given(st.integers(2, 2000), st.integers(2, 3000))
    def pads(h, w):
        l, r, t, b = util.pad_sizes(h, w)
        for size, lo, hi in ((w, l, r), (h, t, b)):
            if size % 128 == 0:
                assert (lo, hi) == (32, 32)                       # already a multiple: 32 px of context per side
            else:
                assert (size + lo + hi) % 128 == 0 and 0 <= hi - lo <= 1 and lo + hi < 128
    pads()

    @settings(max_examples=200, deadline=None)
    @given(st.integers(1, 500), st.integers(0, 499))
    def frames(n, i):
        i = i % n
        ids = harness.window_frame_ids(i, n)
        assert len(ids) == 6 and ids == sorted(ids) and ids[0] >= 0 and ids[-1] <= n - 1
        assert ids[2] == i and all(b - a in (0, 1) for a, b in zip(ids, ids[1:]))
    frames()

    @settings(max_examples=200, deadline=None)
    @given(st.integers(0, 5000), st.integers(1, 64))
    def shards(n, world):
        spans = [harness.shard_windows(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [e - s for s, e in spans]
        assert max(sizes) - min(sizes) <= 1
    shards()


def test_power_bound_reading_is_derived_from_its_numbers():
    """bench.power_bound_reading: three outcomes from the numbers, decided by the device's limiter residency when it is there."""
    import bench

    def summ(clock, power, ppt=None, cap=1400.0, mem=None):
        d = {"clock_mhz": {"mean": clock}, "power_w": {"mean": power, "max": power + 30}, "power_cap_w": cap,
             "limiter": {"source": "unavailable"} if ppt is None else
             {"source": "x", "active_frac": {"ppt_power": ppt, "hbm_thermal": 0.0}, "dominant": "ppt_power" if ppt > 0.05 else None}}
        if mem:
            d["mem_clock_mhz"] = {"mean": mem}
        return d
    zero = summ(2366.0, 1050.0, 0.0)
    r = bench.power_bound_reading(69.2, 53.4, summ(1810.0, 1296.0, 0.98), zero)        # r04 builder box: same cycles
    assert r["reading"].startswith("clock-explained") and r["at_cap"] is True and abs(r["cycle_ratio"] - 1.0) <= 0.03
    r = bench.power_bound_reading(73.9, 53.4, summ(1812.7, 1296.0, 0.98, mem=1800.0), dict(zero, mem_clock_mhz={"mean": 2000.0}))
    assert r["reading"].startswith("not clock-explained") and r["cycle_ratio"] == pytest.approx(1.0603, abs=2e-3)   # r04 driver box
    assert "mem_clock_mhz 1800 vs 2000" in r["reading"]
    r = bench.power_bound_reading(60.0, 53.4, summ(2300.0, 1100.0, 0.02), zero)
    assert r["reading"].startswith("not at the cap") and r["at_cap"] is False
    # without limiter data the watts decide: 1296 W of a 1400 W cap is "at the cap" (>= 0.9), 1100 W is not
    assert bench.power_bound_reading(69.2, 53.4, summ(1810.0, 1296.0), zero)["at_cap"] is True
    assert bench.power_bound_reading(69.2, 53.4, summ(1810.0, 1100.0), zero)["at_cap"] is False
    assert bench.power_bound_reading(69.2, 53.4, {"clock_mhz": None}, zero)["reading"].startswith("undetermined")
    # with the per-XCD table the cycles are counted at the device-wide MEAN clock (amdsmi's GFX clk is the fastest XCD):
    # the r05 box — 71.67 ms at clk 1791 / XCD mean 1733 vs 53.32 ms at 2364 / 2352
    xd = {"xcds": 8, "mean": 1732.9, "slowest_xcd_mean": 1675.0, "fastest_xcd_mean": 1791.2}
    xz = {"xcds": 8, "mean": 2351.9, "slowest_xcd_mean": 2338.2, "fastest_xcd_mean": 2364.1}
    r = bench.power_bound_reading(71.669, 53.322, dict(summ(1791.2, 1308.7, 0.956), xcd_clock_mhz=xd), dict(summ(2363.8, 1210.7, 0.28), xcd_clock_mhz=xz))
    assert r["cycles_clock"].startswith("per-XCD") and r["cycle_ratio"] == pytest.approx(0.9903, abs=1e-3)
    assert r["reading"].startswith("clock-explained") and r["other_domains"]["xcd_clock_mhz"]["cycle_ratio_at_gfx_clk"] == pytest.approx(1.0185, abs=1e-3)
    # fewer cycles on real data (training: a quarter of the step is HBM-bound): the implied clock-independent share is reported
    r = bench.power_bound_reading(132.72, 114.29, dict(summ(1797.5, 1335.0, 0.92), xcd_clock_mhz=dict(xd, mean=1754.6)),
                                  dict(summ(2193.9, 1231.0, 0.48), xcd_clock_mhz=dict(xz, mean=2178.9)))
    assert r["reading"].startswith("not clock-explained") and 0.25 < r["clock_independent_share_implied"] < 0.40


def test_smi_device_index_accepts_every_device_spelling():
    """advisor r04: a str device used to resolve to the bound method `str.index` and the sampler silently reported nothing."""
    from bin_amd.utils import smi
    assert [smi.device_index(d) for d in (None, 0, 3, "cuda", "cuda:0", "cuda:2", "1", torch.device("cuda", 1), torch.device("cuda"))] \
        == [0, 0, 3, 0, 0, 2, 1, 1, 0]
    with pytest.raises(ValueError):
        smi.device_index("cpu")
    s = smi.Sampler("cuda:0")                    # no GPU in the build container: reports the error, never raises
    assert "samples" in s.summary()


def test_bench_live_parity_object():
    """bench.py's cpu_baseline leg also CHECKS the timed window against the oracle it times there (`cpu_baseline.hip_vs_oracle`): identical
    outputs read as max_abs 0 / PSNR null, a perturbed output shows up in every field; the object is JSON-serialisable."""
    import json
    import bench
    from bin_amd.utils import util
    from oracle import rdn_oracle as O
    g = torch.Generator().manual_seed(0)
    ref = [torch.rand(1, 3, 768, 1344, generator=g) for _ in range(2)]
    same = bench.compare_with_oracle([r.clone() for r in ref], ref, O, util)
    assert same["max_abs"] == 0.0 and same["psnr_db_float"] is None and same["u8_values_differing"] == 0 and same["psnr_db_u8_worst"] is None
    assert same["u8_values"] == 2 * 3 * 720 * 1280                                   # the crop test.py writes, not the padded frame
    hip = [r.clone() for r in ref]
    hip[1][0, 0, 24 + 100, 32 + 100] += 0.25                                         # inside the crop (pads: 24 rows, 32 columns)
    hip[0][0, 0, 0, 0] += 0.5                                                        # in the padding: counts for max_abs only
    d = bench.compare_with_oracle(hip, ref, O, util)
    assert abs(d["max_abs"] - 0.5) < 1e-6 and d["u8_values_differing"] == 1 and d["psnr_db_u8_worst"] is not None and d["psnr_db_float"] > 60
    json.dumps(d)


def test_fused_upnet_weights_reproduce_the_two_layers():
    """rdn_plan.fused_upnet_weights: UPNet's two convolutions around the PixelShuffle (RDN.py:203-207, no activation) as one 5x5
    operator on 12 sub-pixel channels + the nine border variants — in float64 the composition equals F.conv2d / F.pixel_shuffle /
    F.conv2d to rounding on every pixel, including 1 x 1 and 2 x 3 inputs where everything is border; the interior operator alone
    is right everywhere but on the one-pixel full-resolution ring."""
    import torch.nn.functional as F
    from bin_amd.rdn_plan import fused_upnet_reference, fused_upnet_weights
    g = torch.Generator().manual_seed(0)
    g0 = 32
    w0, b0 = torch.randn(256, g0, 3, 3, generator=g).double() * 0.1, torch.randn(256, generator=g).double() * 0.1
    w2, b2 = torch.randn(3, 64, 3, 3, generator=g).double() * 0.1, torch.randn(3, generator=g).double()
    W, B = fused_upnet_weights(w0, b0, w2, b2)
    assert W.shape == (9, 12, g0, 5, 5) and B.shape == (9, 12)
    for h, w in ((7, 9), (1, 1), (2, 3), (16, 5)):
        x = torch.randn(2, g0, h, w, generator=g).double()
        ref = F.conv2d(F.pixel_shuffle(F.conv2d(x, w0, b0, padding=1), 2), w2, b2, padding=1)
        assert float((fused_upnet_reference(x, W, B) - ref).abs().max()) <= 1e-12
        if h > 1 and w > 1:
            inter = F.pixel_shuffle(F.conv2d(x, W[4], B[4], padding=2), 2)
            d = (inter - ref).abs()
            assert float(d[..., 1:-1, 1:-1].max()) <= 1e-12 and float(d.max()) > 1e-3      # only the ring needs its own operators
