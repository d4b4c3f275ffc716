"""Small-frame latency / throughput (BASELINE configs 0 and 4: 256x256 -> 320x320, Vimeo 448x256 -> 320x512): a small
frame does not fill 256 CUs (50-80 tiles per launch).  Two remedies: the four-call schedule (stage s of both windows as
one batch along N: 17 -> 4 RDN calls, `BIN_AMD_INFER_FOUR`), and batching consecutive windows along N."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bin_amd.models.archs.RDN import bin_stage4_lstm
from bin_amd.utils import util
from bin_amd.weights import reference_state_dict, synthetic_frames


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


net = bin_stage4_lstm(); net.load_state_dict(reference_state_dict(0)); net = net.cuda().eval()
with torch.no_grad():
    for prec in ("f16x3", "f16"):
        net.set_precision(prec)
        for (h, w) in ((256, 256), (256, 448)):
            pads = util.pad_sizes(h, w)
            for b in (1, 8):
                frames = [util.replicate_pad(f, pads).cuda() for f in synthetic_frames(1234, b, h, w, 6)]
                ref = None
                for four, ns in (("0", 1), ("0", 3), ("1", 1)):
                    net.four_calls_infer, net.n_streams = four, ns
                    out = net(*frames)
                    same = True if ref is None else all(torch.equal(a, c) for a, c in zip(out, ref))
                    ref = ref or out
                    ms = timeit(lambda: net(*frames))
                    print(f"{prec} {h}x{w} batch {b:2d} four_calls {four} streams {ns}: {ms:7.2f} ms per forward = "
                          f"{b / ms * 1e3:7.1f} windows/s  bit-identical {same}", flush=True)
