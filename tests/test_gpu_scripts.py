"""GPU: the folder-level entry points (bin_amd.test = the reference's test.py / demo.py flow with overlapped PNG IO)
against the in-memory harness, and one real bin_amd.train run on the device."""
import os
import random

import numpy as np
import pytest
import torch

from host_fixtures import OPTION_YML, make_adobe_tree

pytestmark = pytest.mark.gpu


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
