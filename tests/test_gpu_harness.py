"""-m gpu: device-side harness glue (u8 <-> frame kernels, clip interpolation, hipGraph replay, test_stitch), the bench line's objects, folder evaluation through bin_amd.test."""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden
from host_fixtures import OPTION_YML, make_adobe_tree

pytestmark = pytest.mark.gpu


def _net(prec, reuse=True):
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.weights import reference_state_dict
    net = bin_stage4_lstm()
    net.load_state_dict(reference_state_dict(0), strict=True)
    net = net.cuda().eval().set_precision(prec)
    net.reuse_schedule = reuse
    return net


def test_harness_glue_kernels_match_reference_helpers():
    """N1: u8->frame (read_image + ReplicationPad2d) and frame->u8 (tensor2img + crop) on the device are bit-exact
    against the reference helpers' restatement (oracle), including round-half-even and out-of-range values."""
    from bin_amd import ops
    from bin_amd.utils import util
    from oracle import rdn_oracle as O
    g = torch.Generator().manual_seed(4)
    img = torch.randint(0, 256, (37, 53, 3), generator=g, dtype=torch.uint8)
    pads = (3, 5, 2, 7)
    got = ops.u8_to_frame(img.cuda(), pads).cpu()
    ref = torch.from_numpy(img.numpy().astype("float32") / 255.0)[:, :, [2, 1, 0]].permute(2, 0, 1).unsqueeze(0)
    ref = O.replicate_pad(ref, pads)
    assert torch.equal(got, ref)
    x = torch.rand(1, 3, 40, 60, generator=g) * 1.5 - 0.25
    x[0, :, 0, :8] = torch.tensor([0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255, 0.0, 1.0, -1.0, 2.0])
    top, left, h, w = 4, 6, 30, 50
    got8 = ops.frame_to_u8(x.cuda(), top, left, h, w).cpu().numpy()
    ref8 = O.tensor2img(x[0])[top:top + h, left:left + w, :]
    assert (got8 == ref8).all()
    assert (ops.frame_to_u8(x.cuda(), 0, 0, 40, 60).cpu().numpy() == util.tensor2img(x[0])).all()


def test_interpolate_clip_u8_sharded():
    """test.py-style loop on a synthetic u8 clip: window sharding over 2 'ranks' covers every window once and
    equals the unsharded run; u8 path == fp32 path."""
    from bin_amd.harness import interpolate_clip
    g = torch.Generator().manual_seed(9)
    clip = torch.randint(0, 256, (5, 40, 72, 3), generator=g, dtype=torch.uint8)
    net = _net("f16")
    full = interpolate_clip(net, clip)
    a = interpolate_clip(net, clip, rank=0, world=2)
    b = interpolate_clip(net, clip, rank=1, world=2)
    assert sorted(full) == [0, 1, 2, 3] and sorted(list(a) + list(b)) == [0, 1, 2, 3]
    for k, v in {**a, **b}.items():
        for x, y in zip(v, full[k]):
            assert (x == y).all() and x.shape == (40, 72, 3) and x.dtype.name == "uint8"
    nocache = interpolate_clip(net, clip, reuse_stage1=False)           # N3: the stage-1 cache changes no bit
    for k in full:
        for x, y in zip(nocache[k], full[k]):
            assert (x == y).all()
    for bsz in (2, 3, 8):                                                # windows batched along N: no bit changes
        batched = interpolate_clip(net, clip, batch=bsz)
        assert sorted(batched) == sorted(full)
        for k in full:
            for x, y in zip(batched[k], full[k]):
                assert (x == y).all()
    clip_f = (clip.float() / 255.0)[:, :, :, [2, 1, 0]].permute(0, 3, 1, 2).contiguous()
    ff = interpolate_clip(net, clip_f)
    for k in full:
        for x, y in zip(ff[k], full[k]):
            assert (x == y).all()


@pytest.mark.parametrize("multi", [False, True], ids=["one_graph_serial", "per_call_graphs_3_streams"])
def test_hipgraph_replay_matches_eager(multi):
    """hipGraph replay (bin_amd.harness.GraphedNet) is bit-identical to eager launches: the whole serial forward as one
    graph, and the 3-stream schedule as 23 per-call graphs joined by eager events."""
    from bin_amd.harness import GraphedNet
    from bin_amd.weights import synthetic_frames
    net = _net("f16")
    f1 = [f.cuda() for f in synthetic_frames(31, 1, 64, 64, 6)]
    f2 = [f.cuda() for f in synthetic_frames(32, 1, 64, 64, 6)]
    with torch.no_grad():
        e1 = [o.clone() for o in net(*f1)]
        e2 = [o.clone() for o in net(*f2)]
    g = GraphedNet(net, f1, multi_stream=multi)
    assert (len(g.call_graphs) == 23) if multi else hasattr(g, "graph")
    for frames, ref in ((f2, e2), (f1, e1), (f2, e2)):
        out = g(*frames)
        torch.cuda.synchronize()
        for x, y in zip(out, ref):
            assert torch.equal(x, y)


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


def test_bench_power_bound_reading_is_computed_from_the_numbers_beside_it():
    """`power_bound` (bench.py): the device's own limiter residency, cycles = ms x clock for the real-data and the all-zero pass,
    and a `reading` that is one of four outcomes DERIVED from them — never a constant string."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                        "--no-extras"], cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    pw, pb = d["power"], d["power_bound"]
    assert pw["samples"] >= 5 and pw["clock_mhz"]["mean"] > 500
    assert "limiter" in pw and "source" in pw["limiter"]
    kind = pb["reading"].split(":")[0]
    assert kind in ("clock-explained", "not clock-explained", "not at the cap", "undetermined")
    assert kind != "undetermined", pb                                  # a GPU box has an smi source
    x = (pb["other_domains"] or {}).get("xcd_clock_mhz")
    fd, fz = (x["data"]["mean"], x["zero"]["mean"]) if x else (pb["clock_mhz"]["data"], pb["clock_mhz"]["zero"])
    assert ("per-XCD" in pb["cycles_clock"]) == bool(x)
    assert pb["cycles_data_M"] == pytest.approx(pb["ms"] * fd * 1e-3, rel=1e-3)
    assert pb["cycles_zero_M"] == pytest.approx(pb["ms_zero"] * fz * 1e-3, rel=1e-3)
    assert pb["cycle_ratio"] == pytest.approx(pb["cycles_data_M"] / pb["cycles_zero_M"], rel=1e-3)
    assert pb["cycle_ratio"] == pytest.approx(pb["ratio"] / pb["clock_ratio_used"], rel=2e-3)
    if x:                          # amdsmi's GFX clk lies within the XCDs' range (it tracks the fastest one; the two are read by
        xd = x["data"]             # separate calls a moment apart, so only the range is asserted — a box whose XCDs run level
        assert xd["slowest_xcd_mean"] <= xd["mean"] <= xd["fastest_xcd_mean"]          # put clk 0.25 % under the mean once)
        assert 0.98 * xd["slowest_xcd_mean"] <= pb["clock_mhz"]["data"] <= 1.02 * xd["fastest_xcd_mean"]
    if kind == "not at the cap":
        assert pb["at_cap"] is False
    else:
        assert pb["at_cap"] is True
        assert (abs(pb["cycle_ratio"] - 1.0) <= 0.03) == (kind == "clock-explained")
    fr = pb["limiter"]["active_frac"]
    if fr and "ppt_power" in fr:                                       # the device's own word decides "at the cap"
        assert pb["at_cap"] == (fr["ppt_power"] >= 0.5) and pb["at_cap_rule"].startswith("device:")
        assert all(0.0 <= v <= 1.0001 for v in fr.values())
    assert pb["power_cap_observed_w"] >= pb["power_w"]["data"] - 1e-6


def _blur_tree(root, clips=(("c0", 0, 5), ("c1", 40, 4)), hw=(72, 100)):
    """test_blur/<clip>/NNNNN.png + test/<clip>/NNNNN.png (sharp at +4 and +8 offsets) of tiny seeded frames."""
    from PIL import Image
    g = np.random.Generator(np.random.PCG64(9))
    for clip, first, n in clips:
        os.makedirs(os.path.join(root, "test_blur", clip))
        os.makedirs(os.path.join(root, "test", clip))
        for k in range(n):
            idx = first + 8 * k
            Image.fromarray(g.integers(0, 256, hw + (3,), dtype=np.uint8)).save(
                os.path.join(root, "test_blur", clip, f"{idx:05d}.png"))
        for idx in range(first, first + 8 * n + 8, 4):
            Image.fromarray(g.integers(0, 256, hw + (3,), dtype=np.uint8)).save(
                os.path.join(root, "test", clip, f"{idx:05d}.png"))
    return root


def _yml(tmp, weights):
    y = OPTION_YML.replace("/tmp/bin_amd_runs", str(tmp)).replace("~/w/adobe_bin.pth", weights)
    y = y.replace("name: debug_host", "name: adobe_stage4")
    p = os.path.join(str(tmp), "opt.yml")
    open(p, "w").write(y)
    return p


def test_folder_evaluation_matches_harness(tmp_path):
    from bin_amd import harness
    from bin_amd import test as run_test
    from bin_amd.data import util as du
    from bin_amd.models import networks
    from bin_amd.weights import reference_state_dict
    root = _blur_tree(str(tmp_path / "data"))
    weights = str(tmp_path / "w.pth")
    torch.save(reference_state_dict(0), weights)
    out = str(tmp_path / "out")
    rc = run_test.main(["--input_path", os.path.join(root, "test_blur"), "--gt_path", os.path.join(root, "test"),
                        "--output_path", out, "--opt", _yml(tmp_path, weights), "--precision", "f16x3",
                        "--io_threads", "4", "--ssim"])
    assert rc == 0
    res = os.path.join(out, "60fps_test_results", "adobe_stage4")
    net = networks.define_G({"network_G": {"which_model_G": "bin_stage4", "precision": "f16x3"}}).cuda().eval()
    net.load_state_dict(reference_state_dict(0), strict=True)
    for clip, first, n in (("c0", 0, 5), ("c1", 40, 4)):
        names = sorted(os.listdir(os.path.join(root, "test_blur", clip)))
        frames = np.stack([du.imread_u8(os.path.join(root, "test_blur", clip, f)) for f in names])
        want = harness.interpolate_clip(net, torch.from_numpy(frames))
        written = sorted(os.listdir(os.path.join(res, clip)))
        # per window: <num+8> interpolated, <num+4> deblurred, <num+12> deblurred (not for the last window)
        expect = set()
        for i in range(n - 1):
            num = first + 8 * i
            expect |= {f"{num + 8:05d}.png", f"{num + 4:05d}.png"} | ({f"{num + 12:05d}.png"} if i < n - 2 else set())
        assert set(written) == expect
        read = lambda k: du.imread_u8(os.path.join(res, clip, f"{k:05d}.png"))
        for i in range(n - 1):
            num = first + 8 * i
            interp, d0, d1 = want[i]
            assert np.array_equal(read(num + 8), interp)
            if i == 0:
                assert np.array_equal(read(num + 4), d0)          # only window 0 owns its first deblurred frame
            if i < n - 2:
                assert np.array_equal(read(num + 12), d1)         # later ones come from the previous window's Ft_p[12]
    logs = [f for f in os.listdir(res) if f.endswith(".log")]
    text = open(os.path.join(res, logs[0])).read()
    assert "Avg. testset" in text and "interp_psnr" in text and "interpolated frames/s" in text
    # windows batched along N (--batch 3, no cross-window stage-1 reuse): the same files, bit for bit
    out_b = str(tmp_path / "out_b")
    assert run_test.main(["--input_path", os.path.join(root, "test_blur"), "--output_path", out_b,
                          "--opt", _yml(tmp_path, weights), "--precision", "f16x3", "--batch", "3"]) == 0
    res_b = os.path.join(out_b, "60fps_test_results", "adobe_stage4")
    for clip in ("c0", "c1"):
        assert sorted(os.listdir(os.path.join(res_b, clip))) == sorted(os.listdir(os.path.join(res, clip)))
        for f in os.listdir(os.path.join(res, clip)):
            assert np.array_equal(du.imread_u8(os.path.join(res_b, clip, f)), du.imread_u8(os.path.join(res, clip, f)))
    # second run: everything exists -> nothing is rewritten (mtime unchanged), still scores
    before = {f: os.path.getmtime(os.path.join(res, "c0", f)) for f in os.listdir(os.path.join(res, "c0"))}
    assert run_test.main(["--input_path", os.path.join(root, "test_blur"), "--output_path", out,
                          "--opt", _yml(tmp_path, weights), "--precision", "f16x3"]) == 0
    after = {f: os.path.getmtime(os.path.join(res, "c0", f)) for f in os.listdir(os.path.join(res, "c0"))}
    assert before == after
