"""BASELINE config 1 shape on the GPU: 6 x [1,3,256,256] demo window padded to 320x320, eager launches vs hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bin_amd.harness import GraphedNet
from bin_amd.models.archs.RDN import bin_stage4_lstm
from bin_amd.utils import util
from bin_amd.weights import reference_state_dict, synthetic_frames

net = bin_stage4_lstm(); net.load_state_dict(reference_state_dict(0)); net = net.cuda().eval().set_precision("f16")
pads = util.pad_sizes(256, 256)
frames = [util.replicate_pad(f, pads).cuda() for f in synthetic_frames(1234, 1, 256, 256, 6)]
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    for ns in (1, 3):
        net.n_streams = ns
        print(f"eager  streams={ns}: {timeit(lambda: net(*frames)):.2f} ms per 320x320 window")
    if len(sys.argv) > 1 and sys.argv[1] == "multi":
        net.n_streams = 3
        g = GraphedNet(net, frames, multi_stream=True)
        ref = net(*frames)
        out = g(*frames)
        torch.cuda.synchronize()
        print("3-stream hipGraph == eager:", all(torch.equal(a, b) for a, b in zip(ref, out)))
        print(f"hipGraph replay (3-stream capture): {timeit(lambda: g(*frames)):.2f} ms per 320x320 window")
    else:
        g = GraphedNet(net, frames)
        print(f"hipGraph replay    : {timeit(lambda: g(*frames)):.2f} ms per 320x320 window")
