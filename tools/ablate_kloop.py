"""K-loop ablation of the dominant RDB conv kernel: builds side libraries with -DBINHIP_ABLATE=n (timing-only
variants, see binhip_conv.hip) into tools/_abl/ and times each with per-launch hipEvent pairs.

  python tools/ablate_kloop.py --build          (build container: hipcc cross-compiles)
  python tools/ablate_kloop.py                  (GPU box: run every variant in a subprocess)
"""
import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OUT = os.path.join(HERE, "_abl")
VARIANTS = {0: "full", 1: "no MFMA", 2: "no fragment loads", 3: "no stage barrier", 4: "no DMA instructions",
            5: "3 convs in 1 launch (no sync)"}


def build():
    sys.path.insert(0, REPO)
    from bin_amd.build import CSRC, SOURCES
    os.makedirs(OUT, exist_ok=True)
    for v in VARIANTS:
        objs = []
        for src in SOURCES:
            obj = os.path.join(OUT, f"{src[:-4]}_{v}.o")
            if src != "binhip_conv.hip" and v != 0:
                obj = os.path.join(OUT, f"{src[:-4]}_0.o")
            else:
                subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                       f"-DBINHIP_ABLATE={v}", "-c", os.path.join(CSRC, src), "-o", obj])
            objs.append(obj)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"libbinhip_abl{v}.so")] + objs)
    print("built", sorted(os.listdir(OUT)))


def run_one(nterms):
    import ctypes as C
    import torch
    sys.path.insert(0, REPO)
    from bin_amd import _lib as L, ops
    lib = L.lib()
    dev = torch.device("cuda")
    n, h, w = 1, 384, 672
    g = torch.Generator().manual_seed(0)
    x = ops.nchw_to_planes(torch.rand(n, 224, h, w, generator=g).to(dev), nterms=nterms)
    tag = os.environ.get("ABL_TAG", "?")
    if tag.startswith("3 convs"):
        wt = torch.randn(32, 192, 3, 3, generator=g) * 0.05
        cw = ops.ConvWeights(wt.to(dev), torch.zeros(32, device=dev), nterms=nterms)
        out = ops.CP.empty(8, n, h, w, nterms, dev)
        for dbg, nm, chunks in ((0, "conv0 + conv1 + conv2 as 3 launches", (6, 8, 10)), (32, "the same as 3 phases of 1 launch", (6,))):
            lib.binhip_set_variant(-2, dbg)
            f = lambda: [ops.conv2d(x, cw, relu=True, out=out, cin_chunks=c) for c in chunks]
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                f()
            e1.record()
            torch.cuda.synchronize()
            print(f"nt={nterms} {tag:30s} {nm:40s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per dense-block front", flush=True)
        lib.binhip_set_variant(-2, 0)
        return
    for cin in (96, 192):
        wt = torch.randn(32, cin, 3, 3, generator=g) * 0.05
        cw = ops.ConvWeights(wt.to(dev), torch.zeros(32, device=dev), nterms=nterms)
        out = ops.CP.empty(2, n, h, w, nterms, dev)
        f = lambda: ops.conv2d(x, cw, relu=True, out=out, cin_chunks=cin // 16)
        for dbg, nm in ((0, ""), (8, "+ no epilogue"), (3, "+ DMA to nowhere"), (11, "+ DMA to nowhere, no epilogue")):
            lib.binhip_set_variant(-2, dbg)
            for _ in range(5):
                f()
            L.check(lib.binhip_profile_begin(3, 32, 0, 64), "profile_begin")
            for _ in range(64):
                f()
            torch.cuda.synchronize()
            ms, cnt = C.c_double(0), C.c_int(0)
            L.check(lib.binhip_profile_end(C.byref(ms), C.byref(cnt)), "profile_end")
            print(f"nt={nterms} {cin:3d}->32  {tag:22s} {nm:32s} {ms.value / max(cnt.value, 1) * 1e3:7.1f} us", flush=True)
        lib.binhip_set_variant(-2, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--one", action="store_true")
    ap.add_argument("--nterms", type=int, default=1)
    args = ap.parse_args()
    if args.build:
        return build()
    if args.one:
        return run_one(args.nterms)
    only = os.environ.get("ABL_ONLY")
    for v, name in VARIANTS.items():
        if only and str(v) not in only.split(","):
            continue
        env = dict(os.environ, BIN_AMD_LIB=os.path.join(OUT, f"libbinhip_abl{v}.so"), ABL_TAG=name)
        subprocess.call([sys.executable, os.path.abspath(__file__), "--one", "--nterms", str(args.nterms)], env=env)


if __name__ == "__main__":
    main()
