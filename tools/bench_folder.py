"""End-to-end folder run of `python -m bin_amd.test` on a synthetic 720p clip (PNG decode -> device -> net -> PNG encode),
to measure what the overlapped IO leaves of the in-HBM rate.  Frames are smooth seeded images so
the PNG codec sees realistic entropy."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def make_clip(root, n_frames, h, w):
    from PIL import Image
    g = np.random.Generator(np.random.PCG64(1))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    os.makedirs(os.path.join(root, "test_blur", "clip0"))
    for k in range(n_frames):
        img = np.stack([127 + 100 * np.sin((xx + 9 * k) / 37.0 + c) * np.cos((yy - 5 * k) / 53.0 - c) for c in range(3)], -1)
        img = (img + g.normal(0, 2.0, img.shape)).clip(0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, "test_blur", "clip0", f"{8 * k:05d}.png"), compress_level=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=33)
    ap.add_argument("--io_threads", type=int, default=12)
    ap.add_argument("--precision", default="f16")
    args = ap.parse_args()
    from host_fixtures import OPTION_YML
    from bin_amd import test as run_test
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.time()
        make_clip(tmp, args.frames, 720, 1280)
        print(f"made {args.frames} frames in {time.time() - t0:.1f} s", flush=True)
        yml = os.path.join(tmp, "o.yml")
        open(yml, "w").write(OPTION_YML.replace("/tmp/bin_amd_runs", tmp).replace("~/w/adobe_bin.pth", "~")
                             .replace("pretrain_model_G: ~", "pretrain_model_G: ~").replace("name: debug_host", "name: e2e"))
        for rep in range(2):                       # second pass: warm page cache / allocator, fresh output dir
            out = os.path.join(tmp, f"out{rep}")
            run_test.main(["--input_path", os.path.join(tmp, "test_blur"), "--output_path", out, "--opt", yml,
                           "--precision", args.precision, "--io_threads", str(args.io_threads)])


if __name__ == "__main__":
    main()
