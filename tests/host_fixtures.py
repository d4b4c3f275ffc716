"""Synthetic inputs shared by the host-logic tests and tests/golden/make_golden_host.py (so the goldens and the
tests are built from the same bytes): an Adobe240-shaped directory tree of tiny seeded PNG frames and an option
file."""
import os

import numpy as np

OPTION_YML = """\
name: debug_host
use_tb_logger: false
model: bin
distortion: blur
scale: 4
gpu_ids: [0]
datasets:
  train:
    name: train
    mode: BIN
    dataroot_GT: ~/data/adobe
    dataroot_LQ: ~/data/adobe
    n_workers: 0
    batch_size: 2
    LQ_size: [3, 32, 32]
  val:
    name: test
    mode: BIN_mc
    dataroot_GT: /data/val.lmdb
    dataroot_LQ: /data/val
    LQ_size: [3, 32, 32]
network_G:
  which_model_G: bin_stage4
  nframes: 6
  version: 2
path:
  pretrain_model_G: ~/w/adobe_bin.pth
  save_path: /tmp/bin_amd_runs
  strict_load: true
  resume_state: ~
train:
  lr_G: !!float 1e-4
  lr_scheme: MultiStepLR
  lr_steps: [4, 8]
  lr_gamma: 0.5
  beta1: 0.9
  beta2: 0.99
  niter: 6
  pixel_criterion: cb
  pixel_weight: 1.0
  val_freq: !!float 5e3
  manual_seed: 0
logger:
  print_freq: 100
  save_checkpoint_freq: !!float 5000
"""


def make_adobe_tree(root, mode="train", clips=(("clipA", 16, 9), ("clipB", 0, 7)), hw=(352, 640), seed=5):
    """<root>/<mode>/<clip>/NNNNN.png (sharp, every index), <mode>_blur/<clip>/NNNNN.png (every 8th) and
    <mode>_list/<clip>_im_list.txt.  clips: (name, first index, number of blurry frames); clipA's list omits its
    last blurry frame so one window is filtered out.  Frames are smooth seeded gradients + noise, 3 x u8."""
    from PIL import Image
    g = np.random.Generator(np.random.PCG64(seed))
    h, w = hw
    yy, xx = np.mgrid[0:h, 0:w]
    for ci, (clip, first, n_blur) in enumerate(clips):
        sharp_dir = os.path.join(root, mode, clip)
        blur_dir = os.path.join(root, mode + "_blur", clip)
        os.makedirs(sharp_dir), os.makedirs(blur_dir)
        os.makedirs(os.path.join(root, mode + "_list"), exist_ok=True)
        names = []
        for k in range(n_blur):
            idx = first + 8 * k
            base = ((xx * (k + 1) + yy * (ci + 2) + idx) % 256).astype(np.uint8)
            img = np.stack([base, base[::-1], base[:, ::-1]], -1)
            img = (img.astype(np.int16) + g.integers(-3, 4, img.shape)).clip(0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(blur_dir, f"{idx:05d}.png"), compress_level=1)
            names.append(f"{idx:05d}.png")
            for off in (0, 4):                               # the sharp frames a window can ask for
                s = (img.astype(np.int16) + off).clip(0, 255).astype(np.uint8)
                Image.fromarray(s).save(os.path.join(sharp_dir, f"{idx + off:05d}.png"), compress_level=1)
        listed = names[:-1] if ci == 0 else names
        with open(os.path.join(root, mode + "_list", clip + "_im_list.txt"), "w") as f:
            f.write("\n".join(listed))
    return root
