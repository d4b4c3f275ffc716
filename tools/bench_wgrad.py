"""Isolated timing of the weight-gradient kernels with the tuning build's ablation switches (BIN_AMD_LIB=tools/_abl/
libbinhip_tuning.so): dbg 0 = product path, 1 = no LDS-DMA, 2 = no LDS reads / MFMAs, 3 = neither (launch + reduction only).
usage: bench_wgrad.py [n h w]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bin_amd import ops, _lib as L
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (40, 128, 128)
DBGS = [int(v, 0) for v in os.environ.get("WG_DBGS", "0,1,2,3").split(",")]
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
lib = L.lib()
LAYERS = [tuple(int(v) for v in l.split(",")) for l in os.environ.get("WG_LAYERS", "3,96,32;3,160,32;3,192,32;3,96,96").split(";")]
for ks, cin, cout in LAYERS:
    zero = 0.0 if os.environ.get("WG_ZERO") == "1" else 1.0          # all-zero operands: same instruction stream, idle datapaths
    x = ops.nchw_to_planes(((torch.rand(n, cin, h, w, generator=g) - 0.3) * zero).to(dev), 3)
    gy = ops.nchw_to_planes(((torch.rand(n, cout, h, w, generator=g) - 0.5) * zero).to(dev), 3)
    row = []
    for dbg in DBGS:
        if hasattr(lib, "binhip_wgrad_set_debug"):
            lib.binhip_wgrad_set_debug(dbg)
        elif dbg:
            break
        f = lambda: ops.conv2d_bwd_weight(x, gy, cout, cin, ks, 3)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        row.append((time.time() - t0) / 20 * 1e6)
    flops = 2.0 * n * h * w * cin * cout * ks * ks * 3
    print(f"ks {ks} cin {cin:4d} cout {cout:3d}: " + "  ".join(f"dbg{d:#x} {t:7.1f} us" for d, t in zip(DBGS, row)) +
          f"   product = {flops / row[0] / 1e6:.0f} TFLOP/s", flush=True)
if hasattr(lib, "binhip_wgrad_set_debug"):
    lib.binhip_wgrad_set_debug(0)
