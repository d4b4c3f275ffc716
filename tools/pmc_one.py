"""Run ONE conv layer class a few times (for rocprofv3 --pmc passes on a single kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bin_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "rdb"
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
n, h, w = 1, 384, 672
x = ops.nchw_to_planes((torch.rand(n, 224, h, w, generator=g) - 0.3).to(dev), 1)
if which == "rdb":
    cw = ops.ConvWeights(((torch.rand(32, 160, 3, 3, generator=g) - 0.5) / 38).to(dev), torch.zeros(32).to(dev), nterms=1)
    out = ops.CP.empty(2, n, h, w, 1, dev)
    f = lambda: ops.conv2d(x, cw, relu=True, out=out, cin_chunks=10)
else:
    raise SystemExit("unknown")
for _ in range(10):
    f()
torch.cuda.synchronize()
