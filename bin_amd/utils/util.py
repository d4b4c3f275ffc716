"""Host helpers of the evaluation harness (counterparts of reference utils/util.py:113-137, 201-231,
utils/AverageMeter.py and the padding rule of test.py:348-366).  Pure numpy/torch host code."""
import logging
import math
import os

import numpy as np
import torch


class AverageMeter:
    """reference utils/AverageMeter.py:1-16."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def tensor2img(tensor, out_type=np.uint8, min_max=(0, 1)):
    """3-D (C,H,W) RGB tensor or 2-D tensor -> HWC BGR uint8 (reference utils/util.py:113-137:
    clamp, scale to [0,1], x255, round, RGB->BGR).  4-D grids (torchvision.make_grid) are not built."""
    t = tensor.detach().squeeze().float().cpu().clamp(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    if t.dim() == 3:
        img = np.transpose(t.numpy()[[2, 1, 0], :, :], (1, 2, 0))
    elif t.dim() == 2:
        img = t.numpy()
    else:
        raise TypeError(f"Only support 3D and 2D tensor. But received with dimension: {t.dim()}")
    if out_type == np.uint8:
        img = (img * 255.0).round()
    return img.astype(out_type)


def calculate_psnr(img1, img2):
    """reference utils/util.py:201-208 (inputs in [0,255])."""
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))


def _gauss_window(size=11, sigma=1.5):
    ax = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    k = np.exp(-(ax ** 2) / (2 * sigma ** 2))
    k /= k.sum()
    return np.outer(k, k)


def _filter_valid(img, win):
    from numpy.lib.stride_tricks import sliding_window_view
    v = sliding_window_view(img, win.shape)
    return np.einsum("ijkl,kl->ij", v, win)


def ssim(img1, img2):
    """reference utils/util.py:211-231 ('valid' 11x11 Gaussian, sigma 1.5), numpy only."""
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    win = _gauss_window()
    mu1, mu2 = _filter_valid(a, win), _filter_valid(b, win)
    s1 = _filter_valid(a * a, win) - mu1 ** 2
    s2 = _filter_valid(b * b, win) - mu2 ** 2
    s12 = _filter_valid(a * b, win) - mu1 * mu2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))
    return m.mean()


def calculate_ssim(img1, img2):
    """reference utils/util.py:234-251: per-channel mean for HWC images."""
    if img1.shape != img2.shape:
        raise ValueError("Input images must have the same dimensions.")
    if img1.ndim == 2:
        return ssim(img1, img2)
    if img1.ndim == 3:
        if img1.shape[2] == 3:
            return float(np.mean([ssim(img1[..., i], img2[..., i]) for i in range(3)]))
        if img1.shape[2] == 1:
            return ssim(np.squeeze(img1), np.squeeze(img2))
    raise ValueError("Wrong input image dimensions.")


def pad_sizes(h, w):
    """test.py:348-366 -> (left, right, top, bottom): pad up to the next multiple of 128 (centred), or
    32+32 when already a multiple."""
    def one(n):
        if n != ((n >> 7) << 7):
            padded = ((n >> 7) + 1) << 7
            a = int((padded - n) / 2)
            return a, padded - n - a
        return 32, 32
    l, r = one(w)
    t, b = one(h)
    return l, r, t, b


def replicate_pad(x, pads):
    """torch.nn.ReplicationPad2d([l, r, t, b]) (test.py:368-371)."""
    return torch.nn.functional.pad(x, list(pads), mode="replicate")


# ---------------------------------------------------------------------------------------------------------------
# run bookkeeping used by the train / test scripts (reference utils/util.py:43-111, 140-142)
def get_timestamp():
    from datetime import datetime
    return datetime.now().strftime("%y%m%d-%H%M%S")


def mkdir(path):
    os.makedirs(path, exist_ok=True)


def mkdirs(paths):
    for p in ([paths] if isinstance(paths, str) else paths):
        mkdir(p)


def mkdir_and_rename(path):
    """A fresh experiment directory; an existing one is kept under `<path>_archived_<timestamp>`."""
    if os.path.exists(path):
        archived = f"{path}_archived_{get_timestamp()}"
        logging.getLogger("base").info("Path already exists. Rename it to [%s]", archived)
        os.rename(path, archived)
    os.makedirs(path)


def set_random_seed(seed):
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def setup_logger(logger_name, root, phase, level=logging.INFO, screen=False, tofile=False):
    """Named logger with the reference's line format; `<root>/<phase>_<timestamp>.log` when tofile."""
    lg = logging.getLogger(logger_name)
    fmt = logging.Formatter("%(asctime)s.%(msecs)03d - %(levelname)s: %(message)s", datefmt="%y-%m-%d %H:%M:%S")
    lg.setLevel(level)
    for old in list(lg.handlers):            # a set-up call defines the handlers: a second run in one process does
        lg.removeHandler(old)                # not log twice / into the previous run's file
        old.close()
    if tofile:
        fh = logging.FileHandler(os.path.join(root, f"{phase}_{get_timestamp()}.log"), mode="w")
        fh.setFormatter(fmt)
        lg.addHandler(fh)
    if screen:
        sh = logging.StreamHandler()
        sh.setFormatter(fmt)
        lg.addHandler(sh)
    return lg


def crop_border(img_list, border):
    """Drop `border` pixels from each side of every HWC image."""
    if border == 0:
        return img_list
    return [v[border:-border, border:-border] for v in img_list]


def save_img(img, img_path, mode="RGB"):
    """Write an HWC uint8 image in cv2's channel order (BGR), as cv2.imwrite would (PNG/JPEG by extension)."""
    from PIL import Image
    img = np.asarray(img)
    if img.ndim == 3 and img.shape[2] == 3:
        img = img[:, :, ::-1]
    elif img.ndim == 3 and img.shape[2] == 1:
        img = img[:, :, 0]
    Image.fromarray(np.ascontiguousarray(img)).save(img_path, compress_level=1)


def png_bytes_striped(img_bgr, pool=None, strips=4, level=1):
    """An 8-bit BGR (HWC, 3 channels) image as the bytes of a standard PNG file, with the DEFLATE work cut into `strips` horizontal
    bands compressed independently — on `pool` (a concurrent.futures executor; zlib releases the GIL) when given — and stitched into
    ONE zlib stream the way pigz does: every band but the last ends in a sync flush (byte-aligned, not final), the last one
    finishes the stream, and the Adler-32 of the whole filtered image closes it.  Rows use PNG filter 2 ("Up": the difference to the
    row above, which costs one vectorised subtraction and compresses smooth frames about as well as an adaptive choice).  Any PNG
    reader decodes it to the same pixels `save_img` would have written; what differs is latency: the LAST window of a clip waits for
    one band's deflate instead of a whole image's (bin_amd/test.py)."""
    import struct
    import zlib
    a = np.asarray(img_bgr)
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
        raise ValueError("png_bytes_striped: HWC uint8 image with 3 channels")
    a = np.ascontiguousarray(a[:, :, ::-1])                                    # RGB, HWC
    h, w, _ = a.shape
    rows = a.reshape(h, w * 3)
    filt = np.empty((h, 1 + w * 3), dtype=np.uint8)
    filt[:, 0] = 2                                                             # filter type Up
    filt[0, 1:] = rows[0]
    np.subtract(rows[1:], rows[:-1], out=filt[1:, 1:])                         # uint8 arithmetic wraps = the PNG definition
    strips = max(1, min(int(strips), h))
    edges = [h * i // strips for i in range(strips + 1)]

    def deflate(i):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        data = filt[edges[i]:edges[i + 1]]
        out = c.compress(data)                                                 # (the buffer protocol: no copy)
        return out + (c.flush(zlib.Z_FINISH) if i == strips - 1 else c.flush(zlib.Z_SYNC_FLUSH))
    parts = list(pool.map(deflate, range(strips))) if (pool is not None and strips > 1) else [deflate(i) for i in range(strips)]
    adler = zlib.adler32(filt)
    idat = b"\x78\x01" + b"".join(parts) + struct.pack(">I", adler & 0xFFFFFFFF)

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", idat)
            + chunk(b"IEND", b""))


def save_png_striped(img_bgr, img_path, pool=None, strips=4):
    with open(img_path, "wb") as f:
        f.write(png_bytes_striped(img_bgr, pool, strips))


class ProgressBar:
    """Two-line console progress bar (reference utils/util.py:255-302): bar + counts, then a message."""

    def __init__(self, task_num=0, bar_width=50, start=True):
        import shutil
        cols = shutil.get_terminal_size().columns
        self.task_num = task_num
        self.bar_width = max(10, min(bar_width, min(int(cols * 0.6), cols - 50)))
        self.completed = 0
        if start:
            self.start()

    def start(self):
        import sys
        import time
        if self.task_num > 0:
            sys.stdout.write("[{}] 0/{}, elapsed: 0s, ETA:\nStart...\n".format(" " * self.bar_width, self.task_num))
        else:
            sys.stdout.write("completed: 0, elapsed: 0s")
        sys.stdout.flush()
        self.start_time = time.time()

    def update(self, msg="In progress..."):
        import sys
        import time
        self.completed += 1
        elapsed = max(time.time() - self.start_time, 1e-9)
        rate = self.completed / elapsed
        if self.task_num > 0:
            frac = self.completed / float(self.task_num)
            eta = int(elapsed * (1 - frac) / frac + 0.5)
            done = int(self.bar_width * frac)
            sys.stdout.write("\033[2F\033[J[{}] {}/{}, {:.1f} task/s, elapsed: {}s, ETA: {:5}s\n{}\n".format(
                ">" * done + "-" * (self.bar_width - done), self.completed, self.task_num, rate, int(elapsed + 0.5),
                eta, msg))
        else:
            sys.stdout.write("completed: {}, elapsed: {}s, {:.1f} tasks/s".format(self.completed, int(elapsed + 0.5),
                                                                                 rate))
        sys.stdout.flush()
