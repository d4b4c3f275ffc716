"""GPU tests added in round 4:
  * VideoBaseModel.test_stitch on the device against the REFERENCE stitcher's fixture (g12_stitch),
  * the bench line's new objects (power sampled outside the timed region, the zero-operand `power_bound` control, the
    roofline's bound / traffic source, the PNG-in -> PNG-out `harness` leg),
  * tools/fp16_headroom.py --checkpoint (a holder of adobe_bin.pth checks the fp16 range in one command),
  * the general ConvLSTM gates convolution refuses a weight that changed between forward and backward.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden

pytestmark = pytest.mark.gpu


def test_stitch_on_the_device_matches_the_reference_fixture(tmp_path):
    """tests/test_cpu_data.py pins the stitcher bit for bit on the CPU; here the same geometry runs on cuda:0 (the stand-in
    generator's float ops may round differently there, the copies may not: 1e-6)."""
    import stitch_cases as SC
    from bin_amd.models.Video_base_model import VideoBaseModel
    g = load_golden("g12_stitch")
    opt = {"model": "video_base", "gpu_ids": [0], "is_train": False, "dist": False,
           "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2},
           "path": {"pretrain_model_G": None, "strict_load": True, "models": str(tmp_path), "training_state": str(tmp_path)}}
    m = VideoBaseModel(opt, netG=SC.StubSR().eval())
    m.feed_data({"LQs": SC.frame()}, need_GT=False)
    with torch.no_grad():
        m.test_stitch(tile_hw=SC.TILE_HW, halo=SC.HALO, scale=SC.SCALE)
    y = m.fake_H
    assert y.is_cuda and tuple(y.shape) == (1, 3, SC.LR_H * SC.SCALE, SC.LR_W * SC.SCALE)
    for k, v in SC.sample(y.cpu()).items():
        assert np.abs(v.numpy() - g[k]).max() <= 1e-6, k
    assert abs(float(y.double().mean()) - float(g["mean"])) <= 1e-7


def test_bench_line_carries_power_bound_harness_and_labelled_roofline():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--harness-frames", "7"], cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and 0 < roof["frac"] <= 1.0
    assert roof["bound"] == ("mfma" if roof["arithmetic_intensity_flop_per_byte"] >= roof["ridge_flop_per_byte"] else "hbm")
    assert roof["regime"].startswith("mfma@power-cap")              # f16x3: 115 FLOP/B on a ridge of 104
    assert roof["hbm"]["frac"] <= 1.0 and roof["mfma"]["frac"] <= 1.0
    assert (roof["traffic"] is None) == (roof["traffic_source"] is None)
    if roof["traffic"] is not None:
        assert os.path.exists(os.path.join(REPO, roof["traffic_source"]))
    assert d["tolerance_mode"]["roofline"]["bound"] == "hbm"        # f16: 230 FLOP/B below a ridge of 312
    # power: sampled in its own pass, never inside the timed region
    assert "separate pass" in d["power"]["sampled"] and d["power"]["repetitions"] >= 1
    pb = d["power_bound"]
    assert pb["ms_zero"] > 0 and pb["ratio"] == pytest.approx(d["ms_per_step"] / pb["ms_zero"], rel=1e-3)
    assert 0.8 < pb["ratio"] < 2.5                                   # all-zero operands are never slower by much, nor 2.5x faster
    tb = d["train"]["power_bound"]
    assert tb["ms_zero"] > 0 and 0.8 < tb["ratio"] < 2.5
    assert np.isfinite(d["train"]["loss"]) and d["train"]["loss"] > 0          # the loss of the REAL data, not of the control
    h = d["harness"]
    assert "error" not in h, h
    assert h["windows"] == 6 and h["png_files_written"] == 6 + 1 + 5           # interp per window, first deblur once, second x5
    assert h["frames_per_s"] > 0 and h["gpu_only_frames_per_s"] > 0 and 0 < h["io_overlap_frac"] < 1.5


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


def test_general_convlstm_conv_refuses_a_weight_changed_before_backward():
    from bin_amd.autograd import _ConvFn
    x = torch.rand(1, 8, 16, 16, device="cuda", requires_grad=True)
    w = (torch.rand(12, 8, 3, 3, device="cuda") - 0.5).requires_grad_()
    b = torch.zeros(12, device="cuda", requires_grad=True)
    y = _ConvFn.apply(x, w, b)
    y.sum().backward()                                             # untouched weight: fine
    assert w.grad is not None and torch.isfinite(w.grad).all()
    y = _ConvFn.apply(x, w, b)
    with torch.no_grad():
        w.mul_(0.5)                                                # what an optimizer step does
    with pytest.raises(RuntimeError, match="modified in place"):
        y.sum().backward()


@pytest.mark.parametrize("prec", ["f16x3", "f16"])
def test_streaming_windows_reuse_every_lstm_free_call(prec, monkeypatch):
    """N3, widened in round 4: the next window's first sub-window repeats 4 stage-1, 2 stage-2 and 1 stage-3 call of this one
    (none sees ConvLSTM state), so a streaming caller's cache brings a window from 17 to 10 RDN calls — sliding forward AND
    backward — and every one of the 14 outputs stays bit-identical to an independent forward of the same six frames."""
    from bin_amd import rdn_plan
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.weights import reference_state_dict, synthetic_frames
    net = bin_stage4_lstm()
    net.load_state_dict(reference_state_dict(0), strict=True)
    net = net.cuda().eval().set_precision(prec)
    net.four_calls_infer = "0"                       # (small test frames would otherwise take the batched four-call schedule)
    clip = [f.cuda() for f in synthetic_frames(91, 1, 64, 96, 11)]
    calls = []
    real = rdn_plan.rdn_forward

    def counting(*a, **k):
        calls.append(1)
        return real(*a, **k)
    with torch.no_grad():
        want = {i: [o.clone() for o in net(*clip[i:i + 6])] for i in range(6)}
        torch.cuda.synchronize()
        monkeypatch.setattr(rdn_plan, "rdn_forward", counting)
        cache, per_window = {}, []
        order = [0, 1, 2, 3, 4, 5, 4, 3, 2]
        for i in order:
            calls.clear()
            got = net(*clip[i:i + 6], stage1_cache=cache)
            per_window.append(len(calls))
            torch.cuda.synchronize()
            for x, y in zip(got, want[i]):
                assert torch.equal(x, y), (i, prec)
    assert per_window == [17] + [10] * (len(order) - 1), per_window
    assert len(cache) <= 11                          # only what the last forward touched is kept


def test_4k_window_runs_whole_in_both_precisions():
    """The reference tiles 4K frames because they do not fit its GPU (Video_base_model.py:189-194); with 288 GB a 6-frame
    2160x3840 window (padded by the test.py rule to 2176x3904, 8.5 M pixels per frame, ~30 GB of workspace in f16x3) goes
    through the network WHOLE.  No CPU oracle finishes at this size in a test's time, so the check is the agreement of the two
    independent kernel families (f16x3: plane-split three-product kernels; f16: the generic single-product kernels + VALU
    UPNet.2) to the f16 mode's own bar, plus finiteness, the status word and the largest 32-bit buffer offsets in use."""
    from bin_amd import ops
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.utils import util
    from bin_amd.weights import reference_state_dict, synthetic_frames
    H, W = 2160, 3840
    pads = util.pad_sizes(H, W)
    frames = [util.replicate_pad(f, pads).cuda() for f in synthetic_frames(4242, 1, H, W, 6)]
    assert frames[0].shape[2] % 32 == 0 and frames[0].shape[3] % 32 == 0
    net = bin_stage4_lstm()
    net.load_state_dict(reference_state_dict(0), strict=True)
    net = net.cuda().eval()
    outs = {}
    with torch.no_grad():
        for prec in ("f16x3", "f16"):
            net.set_precision(prec)
            out = net(*frames)
            torch.cuda.synchronize()
            ops.check_status()
            assert all(torch.isfinite(o).all() for o in out)
            outs[prec] = [o[..., pads[2]:pads[2] + H, pads[0]:pads[0] + W].cpu() for o in (out[13], out[8], out[12])]
            del out
    for a, b in zip(outs["f16x3"], outs["f16"]):
        assert tuple(a.shape) == (1, 3, H, W)
        assert float((a - b).abs().max()) <= 1e-3
    from bin_amd.rdn_plan import release_workspaces
    release_workspaces()
    torch.cuda.empty_cache()
