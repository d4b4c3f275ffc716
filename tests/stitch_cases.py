"""The stand-in generator and the frame of fixture g12_stitch: shared by its generator (tests/golden/make_golden_stitch.py,
which drives the REFERENCE's VideoBaseModel.test_stitch with them) and by the tests (which drive bin_amd's).

The reference stitcher (Video_base_model.py:189-280) is hard-wired to a single-tensor x4 generator: [1,N,C,h,w] ->
[1,C,4h,4w], a 960x540 frame, 320x180 tiles with a 32-px halo.  `StubSR` is such a generator made of exact torch ops, built so
that every geometry mistake shows: a blur (the halo's content reaches the interior), a ramp over the CROP's own coordinates
(a wrong interior offset moves it), frame weights (the order along N matters)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

N_FRAMES, LR_H, LR_W, SCALE, TILE_HW, HALO = 2, 540, 960, 4, (180, 320), 32


class StubSR(nn.Module):
    takes_stacked_frames = True          # bin_amd's wrapper hands such a generator var_L as ONE tensor, like the reference does

    def __init__(self):
        super().__init__()
        self.gain = nn.Parameter(torch.tensor(0.75))           # (a wrapper wants at least one parameter)

    def forward(self, x):                                      # [B, N, C, h, w]
        b, n, c, h, w = x.shape
        wts = torch.arange(1, n + 1, dtype=x.dtype, device=x.device) / float(n * (n + 1) // 2)
        m = (x * wts.view(1, n, 1, 1, 1)).sum(1)
        blur = F.avg_pool2d(m, 5, stride=1, padding=2, count_include_pad=True)
        up = F.interpolate(self.gain * m + (1 - self.gain) * blur, scale_factor=SCALE, mode="nearest")
        yy = torch.arange(SCALE * h, dtype=x.dtype, device=x.device).view(1, 1, -1, 1) / float(SCALE * h)
        xx = torch.arange(SCALE * w, dtype=x.dtype, device=x.device).view(1, 1, 1, -1) / float(SCALE * w)
        return up + 0.125 * yy * xx


def frame():
    g = np.random.Generator(np.random.PCG64(404))
    return torch.from_numpy(g.random((1, N_FRAMES, 3, LR_H, LR_W), dtype=np.float32))


def sample(y):
    """What the fixture keeps of the [1,3,2160,3840] result: a strided grid, and full-resolution bands across the first
    horizontal and vertical tile seams."""
    th, tw = TILE_HW[0] * SCALE, TILE_HW[1] * SCALE
    return {"grid": y[0, :, ::37, ::41].contiguous(), "seam_rows": y[0, :, th - 4:th + 4, ::7].contiguous(),
            "seam_cols": y[0, :, ::7, tw - 4:tw + 4].contiguous()}
