"""Small-frame throughput (BASELINE configs 0 and 4: 256x256 -> 320x320, Vimeo 448x256 -> 320x512): a small frame
does not fill 256 CUs (50-80 tiles per launch), so windows are batched along N — every kernel takes N > 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bin_amd.models.archs.RDN import bin_stage4_lstm
from bin_amd.utils import util
from bin_amd.weights import reference_state_dict, synthetic_frames

net = bin_stage4_lstm(); net.load_state_dict(reference_state_dict(0)); net = net.cuda().eval().set_precision("f16")


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


with torch.no_grad():
    for (h, w) in ((256, 256), (256, 448)):
        pads = util.pad_sizes(h, w)
        for b in (1, 2, 4, 8, 16):
            frames = [util.replicate_pad(f, pads).cuda() for f in synthetic_frames(1234, b, h, w, 6)]
            for ns in (1, 3):
                net.n_streams = ns
                ms = timeit(lambda: net(*frames))
                print(f"{h}x{w} batch {b:2d} streams {ns}: {ms:7.2f} ms per forward = {b / ms * 1e3:7.1f} windows/s", flush=True)
