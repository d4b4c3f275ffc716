"""Evaluation harness pieces of the reference's test.py that touch the hot path: the sliding 6-frame
window rule (test.py:257-261), the padding rule (test.py:348-366), which outputs are consumed
(test.py:380-382) and window-sharded multi-GPU inference (shard by WINDOW index, not by
clip — the 8 Adobe240 clips have 63-418 frames, so per-clip sharding caps the 8-GPU speed-up at 3.1x)."""
import torch

from .utils import util


def shard_windows(n_windows, rank, world):
    """Contiguous, balanced [begin, end) range of the flattened (clip, frame) window list for `rank`."""
    base, rem = divmod(n_windows, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def window_frame_ids(index, n_frames):
    """Indices of the 6 blurry frames of the window centred on frame `index` (test.py:257-261):
    index + [-2, -1, 0, 1, 2, 3], clamped to the clip."""
    return [min(max(index + d, 0), n_frames - 1) for d in (-2, -1, 0, 1, 2, 3)]


@torch.no_grad()
def interpolate_clip(netG, clip, rank=0, world=1):
    """Run the test.py inner loop over `clip` ([T,3,H,W] fp32 in [0,1], any device) for this rank's share
    of the T-1 windows.  Returns {window index: (interp, deblur_first, deblur_second)} as cropped HWC BGR
    uint8 images (what test.py writes with cv2.imwrite), all computed on `netG`'s device."""
    T, _, h, w = clip.shape
    dev = next(netG.parameters()).device
    pads = util.pad_sizes(h, w)
    l, r, t, b = pads
    begin, end = shard_windows(T - 1, rank, world)
    out = {}
    for index in range(begin, end):
        ids = window_frame_ids(index, T)
        frames = [util.replicate_pad(clip[i:i + 1].to(dev), pads) for i in ids]
        Ft_p = netG(*frames)
        imgs = []
        for k in (13, 8, 12):
            imgs.append(util.tensor2img(Ft_p[k][0])[t:t + h, l:l + w, :])
        out[index] = tuple(imgs)
    return out
