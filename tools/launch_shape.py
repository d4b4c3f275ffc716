"""Launch-shape experiment: time PER TILE of ONE kernel against the number of "rounds" of tiles its grid
is, one stream, nothing else resident, sustained for a few seconds at the package limit while the device's own clocks / watts /
limiter residency are sampled (bin_amd/utils/smi.py).

    python tools/launch_shape.py [--secs 4] [--th 16] [--kernels rdb,tail] [--rounds 0.5,1.0,1.3,1.5,2.0,3.0] [--zero]

A tile is TH x 32 half-resolution pixels (dense-block conv: TH = 16 = 8 waves x 2 rows, BINHIP_X3_WN side builds change it;
fused tail: TH = 8); the CU holds two workgroups of either kernel, so one round = 512 tiles.  Only H varies (W = 672 = 21 tile
columns, the 720p working width), so `rounds` is met to within one tile row.  Columns: launch time, the same per tile and per
round-equivalent (512 tiles), shader clock (device mean and slowest XCD), socket watts, the share of firmware samples with the PPT
limiter active, joules per launch from the energy counter.  `--zero` repeats every row on all-zero operands (same instruction
stream, idle datapaths: the cycle-bound time without the power limit)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bin_amd import _lib as L, ops  # noqa: E402
from bin_amd.utils.smi import Sampler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--secs", type=float, default=4.0)
ap.add_argument("--th", type=int, default=16, help="tile rows of the dense-block conv in the loaded library (BINHIP_X3_R * BINHIP_X3_WN)")
ap.add_argument("--kernels", default="rdb,tail")
ap.add_argument("--rounds", default="0.5,1.0,1.3,1.5,2.0,3.0")
ap.add_argument("--zero", action="store_true")
ap.add_argument("--json", default=None)
args = ap.parse_args()

dev = torch.device("cuda")
lib = L.lib()
W, NT, SLOTS = 672, 3, 512
g = torch.Generator().manual_seed(0)


def build(kind, h, zero):
    mk = (lambda *s: torch.zeros(*s)) if zero else (lambda *s: torch.rand(*s, generator=g))
    x = ops.nchw_to_planes((mk(1, 224, h, W) - (0.0 if zero else 0.3)).to(dev), NT)
    if kind == "rdb":
        cin = 160
        wt = mk(32, cin, 3, 3) if zero else (mk(32, cin, 3, 3) - 0.5) / 38
        cw = ops.ConvWeights(wt.to(dev), torch.zeros(32).to(dev), nterms=NT)
        out = ops.CP.empty(2, 1, h, W, NT, dev)
        return lambda: ops.conv2d(x, cw, relu=True, out=out, cin_chunks=cin // 16), (x, cw, out)
    w3 = mk(32, 192, 3, 3) if zero else (mk(32, 192, 3, 3) - 0.5) / 41
    wl = mk(96, 224, 1, 1) if zero else (mk(96, 224, 1, 1) - 0.5) / 15
    cw3 = ops.ConvWeights(w3.to(dev), torch.zeros(32).to(dev), nterms=NT)
    cwl = ops.ConvWeights(wl.to(dev), torch.zeros(96).to(dev), nterms=NT)
    y = ops.CP.empty(6, 1, h, W, NT, dev)
    p = lambda t: t.data_ptr() if t is not None else None
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: L.check(lib.binhip_rdb_tail_fwd(1, h, W, NT, p(x.hi), p(x.lo), p(cw3.w_hi), p(cw3.w_lo), p(cw3.bias),
                                                p(cwl.w_hi), p(cwl.w_lo), p(cwl.bias), p(y.hi), p(y.lo), 0, None, stream), "tail")
    return f, (x, cw3, cwl, y)


rows = []
print(f"library: {os.environ.get('BIN_AMD_LIB', 'product')}   secs per row: {args.secs}")
hdr = ("kernel", "data", "H", "tiles", "rounds", "us/launch", "us/tile", "us/512tiles", "MHz", "slowXCD", "memMHz", "W", "ppt", "mJ/launch", "uJ/tile")
print(" ".join(f"{h:>11}" for h in hdr))
for kind in args.kernels.split(","):
    th = args.th if kind == "rdb" else 8
    for r in [float(v) for v in args.rounds.split(",")]:
        k = max(1, round(r * SLOTS / 21))
        h = th * k
        tiles = 21 * k
        for zero in ([False, True] if args.zero else [False]):
            f, keep = build(kind, h, zero)
            for _ in range(50):
                f()
            torch.cuda.synchronize()
            smp = Sampler(dev, 0.02).start()
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < args.secs:
                for _ in range(100):
                    f()
                torch.cuda.synchronize()
                n += 100
            dt = time.perf_counter() - t0
            s = smp.stop()
            us = dt / n * 1e6
            lim = (s.get("limiter") or {}).get("active_frac") or {}
            xc = s.get("xcd_clock_mhz") or {}
            row = {"kernel": kind, "data": "zero" if zero else "real", "H": h, "tiles": tiles, "rounds": round(tiles / SLOTS, 3),
                   "us_per_launch": round(us, 2), "us_per_tile": round(us / tiles, 4), "us_per_512_tiles": round(us / tiles * SLOTS, 2),
                   "clock_mhz": (s.get("clock_mhz") or {}).get("mean"), "slowest_xcd_mhz": xc.get("slowest_xcd_mean"),
                   "xcd_mean_mhz": xc.get("mean"), "mem_clock_mhz": (s.get("mem_clock_mhz") or {}).get("mean"),
                   "power_w": (s.get("power_w") or {}).get("mean"), "power_from_energy_w": s.get("power_from_energy_w"),
                   "ppt_frac": lim.get("ppt_power"), "limiter": lim,
                   "mj_per_launch": None if not s.get("energy_j") else round(s["energy_j"] / n * 1e3, 3),
                   "uj_per_tile": None if not s.get("energy_j") else round(s["energy_j"] / n / tiles * 1e6, 2),
                   "mcycles_per_512_tiles": None if not (s.get("clock_mhz") or {}).get("mean") else
                   round(us / tiles * SLOTS * s["clock_mhz"]["mean"] * 1e-6, 4)}
            rows.append(row)
            vals = (kind, row["data"], h, tiles, row["rounds"], row["us_per_launch"], row["us_per_tile"], row["us_per_512_tiles"],
                    row["clock_mhz"], row["slowest_xcd_mhz"], row["mem_clock_mhz"], row["power_w"], row["ppt_frac"],
                    row["mj_per_launch"], row["uj_per_tile"])
            print(" ".join(f"{str(v):>11}" for v in vals), flush=True)
            del f, keep
            torch.cuda.empty_cache()
if args.json:
    with open(args.json, "w") as fh:
        json.dump({"library": os.environ.get("BIN_AMD_LIB", "product"), "secs": args.secs, "rows": rows}, fh, indent=1)
