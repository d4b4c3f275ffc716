"""Synthetic clips with the STRUCTURE of the reference's training data, for runs without the Adobe240 frames (the dataset is
a Drive link: /root/reference/Adobe_240fps_dataset/download_*.txt).

The reference's sample (data/BIN_dataset.py:95-183) is eleven consecutive instants of one scene: six blurry frames B1, B3 … B11
(each the average of the sharp 240-fps frames inside its exposure), the six sharp frames at the same instants (`GTenh`) and the
five sharp frames in between (`GTinp`).  `moving_texture_batch` renders exactly that from a band-limited random texture in
uniform motion (translation + slow zoom, velocities of a few pixels per instant): sharp frame t = the texture sampled at
position(t), blurry frame t = the mean of `taps` renders across the exposure interval [t - e/2, t + e/2].  A network can learn the task —
the loss of `bin_model.optimize_parameters` falls within tens of steps — which makes it usable for convergence tests and for
producing weights that went through real optimisation (tests/test_gpu_train.py), unlike white noise, where the best answer is
the mean of the inputs.  Deterministic in (seed, sample index); plain torch on the CPU.
"""
import math

import torch
import torch.nn.functional as F


def _texture(gen, size):
    """Band-limited RGB texture in [0, 1], `size` x `size`: three octaves of bicubically upsampled noise."""
    tex = torch.zeros(1, 3, size, size)
    for cells, amp in ((4, 0.5), (12, 0.3), (36, 0.2)):
        n = torch.rand(1, 3, cells, cells, generator=gen)
        tex += amp * F.interpolate(n, size=(size, size), mode="bicubic", align_corners=False)
    lo, hi = tex.amin(), tex.amax()
    return ((tex - lo) / (hi - lo).clamp_min(1e-6)).clamp_(0.0, 1.0)


def _render(tex, cx, cy, zoom, out):
    """`out` x `out` crop of `tex` centred at (cx, cy) (texture pixels, fractional), magnified by `zoom`; bilinear."""
    size = tex.shape[-1]
    lin = (torch.arange(out, dtype=torch.float32) - (out - 1) / 2.0) / zoom
    xs = (cx + lin) / (size - 1) * 2.0 - 1.0
    ys = (cy + lin) / (size - 1) * 2.0 - 1.0
    grid = torch.stack((xs[None, :].expand(out, out), ys[:, None].expand(out, out)), -1)[None]
    return F.grid_sample(tex, grid, mode="bilinear", padding_mode="reflection", align_corners=True)[0]


def moving_texture_clip(seed, index, size, max_speed=3.0, exposure=1.6, taps=7):
    """One sample: (LQs [6,3,S,S], GTenh [6,3,S,S], GTinp [5,3,S,S]) as in BIN_dataset.__getitem__ (BIN_dataset.py:170-183)."""
    gen = torch.Generator().manual_seed((int(seed) * 1000003 + int(index)) & 0x7FFFFFFF)
    canvas = 2 * size + int(2 * 12 * max_speed) + 8
    tex = _texture(gen, canvas)
    ang = float(torch.rand((), generator=gen)) * 2.0 * math.pi
    speed = (0.3 + 0.7 * float(torch.rand((), generator=gen))) * max_speed
    vx, vy = speed * math.cos(ang), speed * math.sin(ang)
    zrate = (float(torch.rand((), generator=gen)) - 0.5) * 0.01            # relative zoom per instant
    c0 = (canvas - 1) / 2.0

    def sharp(t):
        return _render(tex, c0 + vx * (t - 6.0), c0 + vy * (t - 6.0), 1.0 + zrate * (t - 6.0), size)

    def blurry(t):
        acc = 0.0
        for k in range(taps):
            acc = acc + sharp(t + exposure * (k / (taps - 1.0) - 0.5))
        return acc / taps

    lq = torch.stack([blurry(float(t)) for t in (1, 3, 5, 7, 9, 11)])
    enh = torch.stack([sharp(float(t)) for t in (1, 3, 5, 7, 9, 11)])
    inp = torch.stack([sharp(float(t)) for t in (2, 4, 6, 8, 10)])
    return lq, enh, inp


def moving_texture_batch(seed, first_index, batch, size, **kw):
    """The dict `bin_model.feed_data` takes (bin_model.py:147-202): LQs [B,6,3,S,S], GTenh [B,6,3,S,S], GTinp [B,5,3,S,S]."""
    clips = [moving_texture_clip(seed, first_index + b, size, **kw) for b in range(batch)]
    return {"LQs": torch.stack([c[0] for c in clips]), "GTenh": torch.stack([c[1] for c in clips]),
            "GTinp": torch.stack([c[2] for c in clips])}


class SyntheticTextureDataset(torch.utils.data.Dataset):
    """`datasets.<phase>.mode: synthetic_texture` — the moving-texture clips behind the `BINDataset` item contract
    (BIN_dataset.py:170-183: LQs / GTenh / GTinp / key), so that `python -m bin_amd.train -opt options/bin_stage4_synthetic.yml` trains
    and validates with no frames on disk.  Options: `LQ_size: [3, S, S]` (crop), `num_windows` (default 1000), `seed` (default 0; the val
    phase draws from a different stream), `max_speed` (pixels per instant, default 3)."""

    def __init__(self, opt):
        size = opt["LQ_size"] or [3, 256, 256]
        if size[-1] != size[-2]:
            raise ValueError("synthetic_texture renders square crops: LQ_size must be [3, S, S]")
        self.size = int(size[-1])
        self.n = int(opt["num_windows"] or 1000)
        self.seed = int(opt["seed"] or 0) + (100003 if opt["phase"] != "train" else 0)
        self.max_speed = float(opt["max_speed"] or 3.0)
        # (the trainer's probe reads `all_paths[i][3]`, the window key, as it does for BINDataset)
        self.all_paths = [(None, None, None, "synthetic/%06d" % i) for i in range(self.n)]

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        lq, enh, inp = moving_texture_clip(self.seed, index, self.size, max_speed=self.max_speed)
        return {"LQs": lq, "GTenh": enh, "GTinp": inp, "key": self.all_paths[index][3]}
