"""Adobe240 blurry-interpolation training/validation dataset (reference data/BIN_dataset.py:12-287).

Directory contract (made by the reference's data_scripts/adobe240fps/create_dataset_blur_N_frames_average.py):
  <root>/<mode>/<clip>/NNNNN.png        sharp 240-fps frames
  <root>/<mode>_blur/<clip>/NNNNN.png   blurry 30-fps frames (every 8th index)
  <root>/<mode>_list/<clip>_im_list.txt names of the usable blurry frames
One sample = 6 blurry frames B1,B3,..,B11 (8 apart), the 6 sharp frames at the same indices (I1..I11) and the 5
sharp frames half-way (I2..I10); windows slide by one blurry frame.

NOTE the reference's `_make_dataset_deep_long_` falls off its end without returning (BIN_dataset.py:283-287), so
`BINDataset(opt)` raises there; this class implements what that code computes up to that point (the window list,
shuffled, `split` % kept) and the loader's crop/flip/reverse draws in the same order, which the goldens pin."""
import math
import os
import random

import numpy as np
import torch
import torch.utils.data as data

from . import util

NUM_WIN_PER_BUNCH = 4
BLUR_STEP = 8                    # sharp frames per blurry frame
SRC_H, SRC_W = 352, 640          # frame size of the prepared dataset (crop offsets are drawn against it)


def make_window_list(root, mode="train", split=100, shuffle=True):
    """[[6 blurry paths], [6 sharp paths], [5 in-between sharp paths], key] per window, as the reference builds
    them; windows whose blurry frames are not all named in the clip's im_list are dropped."""
    sharp_root = os.path.join(root, mode)
    blur_root = os.path.join(root, mode + "_blur")
    list_root = os.path.join(root, mode + "_list")
    windows = []
    for clip in os.listdir(blur_root):
        blur_dir, sharp_dir = os.path.join(blur_root, clip), os.path.join(sharp_root, clip)
        blur_pics = sorted(os.listdir(blur_dir))
        with open(os.path.join(list_root, clip + "_im_list.txt")) as f:
            usable = set(f.read().split("\n"))
        first = int(blur_pics[0][:-4])
        for win in range(len(blur_pics) - NUM_WIN_PER_BUNCH - 1):
            base = first + BLUR_STEP * win
            name = lambda i: str(i).zfill(5) + ".png"
            blurry = [name(base + BLUR_STEP * k) for k in range(6)]
            if not all(b in usable for b in blurry):
                continue
            windows.append([[os.path.join(blur_dir, b) for b in blurry],
                            [os.path.join(sharp_dir, b) for b in blurry],
                            [os.path.join(sharp_dir, name(base + BLUR_STEP * k + BLUR_STEP // 2)) for k in range(5)],
                            clip + "_" + blurry[0][:-4]])
    if shuffle:
        random.shuffle(windows)
    keep = int(math.floor(len(windows) * split / 100.0))
    return windows[:keep], windows[keep:]


def load_window(window, input_frame_size=(3, 128, 256), data_aug=True):
    """Read the 17 frames of one window, with the reference's augmentation draws in its order: temporal order
    (randint: 1 keeps it, 0 reverses; no aug => reversed, as in the reference), crop offsets (choice, choice),
    horizontal flip (randint).  Returns ([B1..B11], [I1..I11], [I2..I10], key) as float32 HWC BGR crops."""
    blurry, sharp, mid, key = window
    if not (data_aug and random.randint(0, 1)):
        blurry, sharp, mid = blurry[::-1], sharp[::-1], mid[::-1]
    frames = [util.read_img(p) for p in list(blurry) + list(sharp) + list(mid)]
    _, ch, cw = input_frame_size
    y0 = random.choice(range(SRC_H - ch + 1))
    x0 = random.choice(range(SRC_W - cw + 1))
    frames = [f[y0:y0 + ch, x0:x0 + cw, :] for f in frames]
    if data_aug and random.randint(0, 1):
        frames = [np.fliplr(f) for f in frames]
    return frames[:6], frames[6:12], frames[12:], key


class BINDataset(data.Dataset):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.GT_root, self.LQ_root = opt["dataroot_GT"], opt["dataroot_LQ"]
        self.data_type = opt.get("data_type", "img")
        self.input_frame_size = tuple(opt["LQ_size"])
        self.all_paths, _ = make_window_list(self.LQ_root, mode=opt["name"])

    def __len__(self):
        return len(self.all_paths)

    @staticmethod
    def Adobe_BIN_loader(im_path_pair, input_frame_size=(3, 128, 256), data_aug=True, transform=None):
        return load_window(im_path_pair, input_frame_size, data_aug)

    @staticmethod
    def _to_tensor(frames):
        a = np.stack(frames, axis=0)[:, :, :, [2, 1, 0]]                    # T H W C, BGR -> RGB
        return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 3, 1, 2))).float()

    def __getitem__(self, index):
        LQs, GTenh, GTinp, key = load_window(self.all_paths[index], self.input_frame_size)
        return {"LQs": self._to_tensor(LQs), "GTenh": self._to_tensor(GTenh), "GTinp": self._to_tensor(GTinp),
                "key": key}
