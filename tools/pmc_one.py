"""Run ONE conv layer class a few times (for rocprofv3 --pmc / --kernel-trace passes on a single kernel).
usage: pmc_one.py [rdb|tail|wgrad] [nterms] [cin] [variant]   (variant needs BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bin_amd import _lib as L, ops
which = sys.argv[1] if len(sys.argv) > 1 else "rdb"
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cin = int(sys.argv[3]) if len(sys.argv) > 3 else 160
variant = int(sys.argv[4]) if len(sys.argv) > 4 else -1
lib = L.lib()
if variant != -1:
    lib.binhip_set_variant(0, variant)
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
n, h, w = 1, 384, 672
x = ops.nchw_to_planes((torch.rand(n, 224, h, w, generator=g) - 0.3).to(dev), nt)
if which == "rdb":
    cw = ops.ConvWeights(((torch.rand(32, cin, 3, 3, generator=g) - 0.5) / 38).to(dev), torch.zeros(32).to(dev), nterms=nt)
    out = ops.CP.empty(2, n, h, w, nt, dev)
    f = lambda: ops.conv2d(x, cw, relu=True, out=out, cin_chunks=cin // 16)
elif which == "tail":
    cw3 = ops.ConvWeights(((torch.rand(32, 192, 3, 3, generator=g) - 0.5) / 41).to(dev), torch.zeros(32).to(dev), nterms=nt)
    cwl = ops.ConvWeights(((torch.rand(96, 224, 1, 1, generator=g) - 0.5) / 15).to(dev), torch.zeros(96).to(dev), nterms=nt)
    y = ops.CP.empty(6, n, h, w, nt, dev)
    p = lambda t: t.data_ptr() if t is not None else None
    f = lambda: L.check(lib.binhip_rdb_tail_fwd(n, h, w, nt, p(x.hi), p(x.lo), p(cw3.w_hi), p(cw3.w_lo), p(cw3.bias),
                                                p(cwl.w_hi), p(cwl.w_lo), p(cwl.bias), p(y.hi), p(y.lo), 0, None,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tail")
elif which == "wgrad":          # 3x3 weight gradient cin -> 32 on the training working size (40 x 128 x 128)
    xs = ops.nchw_to_planes((torch.rand(40, cin, 128, 128, generator=g) - 0.3).to(dev), nt)
    gy = ops.nchw_to_planes((torch.rand(40, 32, 128, 128, generator=g) - 0.5).to(dev), nt)
    f = lambda: ops.conv2d_bwd_weight(xs, gy, 32, cin, 3, nt)
else:
    raise SystemExit("unknown")
for _ in range(10):
    f()
torch.cuda.synchronize()
