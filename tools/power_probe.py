"""Package power of ONE kernel under sustained load: loops the dense-block conv (f16x3, Cin = 160, 720p working size) for a
few seconds while rocm-smi is sampled (tools/smi_watch.sh).  Run it against the product library and against
-DBINHIP_ABLATE=1|2|4 side builds (BIN_AMD_LIB) to split the power into matrix / LDS-read / LDS-DMA shares.
usage: power_probe.py [seconds] [zero]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bin_amd import ops
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
zero = len(sys.argv) > 2 and sys.argv[2] == "zero"
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
n, h, w, nt, cin = 1, 384, 672, 3, 160
x = torch.zeros(n, 224, h, w) if zero else (torch.rand(n, 224, h, w, generator=g) - 0.3)
x = ops.nchw_to_planes(x.to(dev), nt)
wt = torch.zeros(32, cin, 3, 3) if zero else (torch.rand(32, cin, 3, 3, generator=g) - 0.5) / 38
cw = ops.ConvWeights(wt.to(dev), torch.zeros(32).to(dev), nterms=nt)
out = ops.CP.empty(2, n, h, w, nt, dev)
f = lambda: ops.conv2d(x, cw, relu=True, out=out, cin_chunks=cin // 16)
for _ in range(20):
    f()
torch.cuda.synchronize()
t0 = time.time(); k = 0
while time.time() - t0 < secs:
    for _ in range(200):
        f()
    torch.cuda.synchronize(); k += 200
dt = time.time() - t0
print(f"{os.environ.get('BIN_AMD_LIB', 'product')}: {k} launches in {dt:.2f} s = {dt / k * 1e6:.1f} us per launch", flush=True)
