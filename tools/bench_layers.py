#!/usr/bin/env python3
"""Per-layer microbenchmark at the 720p working size (half-res 384x672): times each distinct layer class of the
RDN forward with events on the launch stream and prints us / actual fp16 GB/s / TFLOP/s.  With the BINHIP_TUNING
side build (`python -m bin_amd.build --tuning`, then BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so) it also sweeps the
kernel variants (binhip_set_variant) and reports the largest difference to the default variant's result."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bin_amd import _lib as L, ops


def time_fn(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nterms", type=int, default=1)
    ap.add_argument("--h", type=int, default=384)
    ap.add_argument("--w", type=int, default=672)
    ap.add_argument("--n", type=int, default=1)
    ap.add_argument("--classes", default="0,1,2,3,4,5")
    args = ap.parse_args()
    nt, n, h, w = args.nterms, args.n, args.h, args.w
    lib = L.lib()
    tuning = hasattr(lib, "binhip_set_variant")
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(0)
    px = n * h * w
    bpe = 2 if nt == 1 else 4

    def mk(c, hh=h, ww=w):
        return ops.nchw_to_planes(torch.rand(n, c, hh, ww, generator=g).to(dev) - 0.3, nt)

    def wts(cout, cin, ks, shuffle=False):
        wt = (torch.rand(cout, cin, ks, ks, generator=g) - 0.5) / (cin * ks * ks) ** 0.5
        return ops.ConvWeights(wt.to(dev), (torch.rand(cout, generator=g) - 0.5).to(dev), nterms=nt, shuffle=shuffle)

    cases = []   # (class id, name, variants, fn factory, flops, bytes)
    x224 = mk(224)
    res96 = mk(96)
    for cin in (96, 192):
        cw = wts(32, cin, 3)
        out = ops.CP.empty(2, n, h, w, nt, dev)
        cases.append((0, f"RDB conv3x3 {cin}->32 +ReLU", [-1, 0, 12] if nt == 1 else [-1, 1],
                      (lambda cw=cw, cin=cin, out=out: ops.conv2d(x224, cw, relu=True, out=out, cin_chunks=cin // 16)),
                      2 * 9 * cin * 32 * px, (cin + 32) * bpe * px, out))
    cw = wts(96, 224, 1)
    out96 = ops.CP.empty(6, n, h, w, nt, dev)
    cases.append((1, "LFF 1x1 224->96 +res", [-1],
                  (lambda cw=cw: ops.conv2d(x224, cw, residual=res96, out=out96)), 2 * 224 * 96 * px,
                  (224 + 96 + 96) * bpe * px, out96))
    x1152 = mk(1152)
    cwg = wts(96, 1152, 1)
    outg = ops.CP.empty(6, n, h, w, nt, dev)
    cases.append((1, "GFF.0 1x1 1152->96", [-1],
                  (lambda: ops.conv2d(x1152, cwg, out=outg)), 2 * 1152 * 96 * px, (1152 + 96) * bpe * px, outg))
    cw3 = wts(96, 96, 3)
    out3 = ops.CP.empty(6, n, h, w, nt, dev)
    cases.append((2, "3x3 96->96 +res", [-1],
                  (lambda: ops.conv2d(res96, cw3, residual=x224, out=out3)), 2 * 9 * 96 * 96 * px,
                  (96 + 96 + 96) * bpe * px, out3))
    cwu = wts(256, 96, 3, shuffle=True)
    outu = ops.CP.empty(4, n, 2 * h, 2 * w, nt, dev)
    cases.append((3, "UPNet.0 3x3 96->256 +shuffle", [-1],
                  (lambda: ops.conv2d(res96, cwu, out=outu, epilogue=L.EPI_SHUFFLE)), 2 * 9 * 96 * 256 * px,
                  (96 + 256) * bpe * px, outu))
    xu = mk(64, 2 * h, 2 * w)
    cwf = wts(3, 64, 3)
    imgs = [torch.rand(n, 3, 2 * h, 2 * w, generator=g).to(dev) for _ in range(2)]
    cases.append((4, "UPNet.2 3x3 64->3 +mean (final)", [-1] if nt == 1 else [-1, 1],
                  (lambda: ops.conv2d(xu, cwf, epilogue=L.EPI_FINAL, images=imgs)), 2 * 9 * 64 * 3 * 4 * px,
                  (64 * bpe + 3 * 4 * 3) * 4 * px, None))
    x48 = mk(48)
    cw5 = wts(96, 36, 5)
    out5 = ops.CP.empty(6, n, h, w, nt, dev)
    cases.append((5, "SFENet1 5x5 36->96", [-1], (lambda: ops.conv2d(x48, cw5, out=out5, cin_chunks=3)),
                  2 * 25 * 36 * 96 * px, (48 + 96) * bpe * px, out5))

    # fused conv#3 + LFF vs the two separate kernels
    import ctypes as C
    blk = mk(224)
    cw3b = wts(32, 192, 3)
    cwlb = wts(96, 224, 1)
    ynext = ops.CP.empty(6, n, h, w, nt, dev)

    def fused(store=0):
        L.check(lib.binhip_rdb_tail_fwd(n, h, w, nt, blk.hi.data_ptr(), blk.lo.data_ptr() if blk.lo is not None else None,
                                        cw3b.w_hi.data_ptr(), cw3b.w_lo.data_ptr() if cw3b.w_lo is not None else None,
                                        cw3b.bias.data_ptr(), cwlb.w_hi.data_ptr(),
                                        cwlb.w_lo.data_ptr() if cwlb.w_lo is not None else None, cwlb.bias.data_ptr(),
                                        ynext.hi.data_ptr(), ynext.lo.data_ptr() if ynext.lo is not None else None, store,
                                        None, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tail")

    def unfused():
        ops.conv2d(blk, cw3b, relu=True, out=blk.sub(12, 2), cin_chunks=12)
        ops.conv2d(blk, cwlb, residual=blk.sub(0, 6), out=ynext)

    fl = (2 * 9 * 192 * 32 + 2 * 224 * 96) * px
    us = time_fn(fused)
    print(f"fused conv3+LFF:              {us:8.1f} us  {(192 + 96) * bpe * px / us / 1e3:7.0f} GB/s(actual)  {fl / us / 1e6:7.0f} TF/s")
    us = time_fn(lambda: fused(1))
    print(f"fused conv3+LFF (+store o3):  {us:8.1f} us")
    us = time_fn(unfused)
    print(f"unfused conv3 ; LFF:          {us:8.1f} us")
    if os.environ.get("WGRAD") and tuning:
        # weight-gradient kernels at the training working size (8 x 128 x 128 half-res pixels)
        nn_, hh_, ww_ = 8, 128, 128
        gg = torch.Generator().manual_seed(1)
        for (ks, cin, cout) in ((3, 96, 32), (3, 192, 32), (1, 224, 96), (3, 96, 96), (1, 1152, 96)):
            xx = ops.nchw_to_planes(torch.rand(nn_, cin, hh_, ww_, generator=gg).to(dev), nt)
            gy = ops.nchw_to_planes(torch.rand(nn_, cout, hh_, ww_, generator=gg).to(dev) - 0.5, nt)
            f = (lambda xx=xx, gy=gy, ks=ks, cin=cin, cout=cout: ops.conv2d_bwd_weight(xx, gy, cout, cin, ks, nt))
            for dbg, nm in ((0, "full"), (16, "full, 1 wg/CU double-buffered"), (1, "no DMA"), (2, "no MFMA"), (4, "no reduce/store"), (7, "nothing")):
                lib.binhip_wgrad_set_debug(dbg)
                us = time_fn(f, iters=10, warm=2)
                print(f"wgrad k{ks} {cin}->{cout} nt={nt} {nm:16s}: {us:8.1f} us   {2*ks*ks*cin*cout*nn_*hh_*ww_*(3 if nt==3 else 1)/us/1e6:7.0f} TF-eq/s")
            lib.binhip_wgrad_set_debug(0)
    want = {int(c) for c in args.classes.split(",")}
    print(f"nterms={nt} N={n} {h}x{w}")
    for cls, name, variants, fn, flops, nbytes, outbuf in cases:
        if cls not in want:
            continue
        ref = None
        for v in (variants if tuning else [-1]):
            if tuning:
                lib.binhip_set_variant(cls, v)
            r = fn()
            torch.cuda.synchronize()
            if outbuf is not None:
                cur = outbuf.hi.float() + (outbuf.lo.float() if outbuf.lo is not None else 0)
            else:
                cur = r.clone()
            if ref is None:
                ref = cur
            diff = float((ref - cur).abs().max()) / max(float(ref.abs().max()), 1e-30)
            us = time_fn(fn)
            print(f"{name:34s} v{v}: {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s(actual)  "
                  f"{flops / us / 1e6:7.0f} TF/s  rel.diff to default {diff:.2e}")
        if tuning:
            lib.binhip_set_variant(cls, -1)


if __name__ == "__main__":
    main()
