"""Per-workgroup timeline of the fp32-class dense-block kernels on MI355X.

Needs the BINHIP_TIMELINE side build:
    python -c "from bin_amd.build import build_library as b; b(force=True, defines=('BINHIP_TIMELINE=1',), out='tools/_abl/libbinhip_timeline.so')"
    BIN_AMD_LIB=tools/_abl/libbinhip_timeline.so python tools/wg_timeline.py [--plan-flags 4] [--out gpurun_out/tl]

Wave 0 of every workgroup of `conv_x3_kernel` / `rdb_tail_x3_kernel` stamps s_memrealtime (100 MHz, one time base for the
whole package) at entry / first MFMA / last MFMA / last store issued / stores drained, and s_memtime (shader cycles of ITS XCD)
at entry and exit (bin_amd/csrc/binhip_conv_common.h, BhTl).  This script runs the 720p window at the package limit (real
operands, warm), records ONE window (17 RDN calls), and reports per kernel kind:
  * launch span (first entry -> last drain), gap to the next instrumented launch of the stream;
  * dispatch stagger (entry - first entry), prologue (entry -> first MFMA), K loop, epilogue issue, store drain, workgroup life;
  * per-XCD finish times (skew), per-XCD shader clock over the workgroup's life;
  * the share of launch-span x slots in which no workgroup is alive, split into ramp / tail / boundary.
Output: <out>.json (summary) and <out>.npz (raw records)."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

REC_WORDS = 20          # 80-byte records as uint32
KIND = {1: "rdb_conv", 2: "tail", 3: "other_x3", 4: "rdbs (convs 0-2 in one launch)"}


def parse(raw):
    """raw: uint32 [n, 20] -> dict of arrays"""
    u64 = raw[:, 4:18].copy().view(np.uint64).reshape(-1, 7)
    return {"kind": raw[:, 0] >> 24, "launch": raw[:, 0] & 0xFFFFFF, "bid": raw[:, 1], "hwid": raw[:, 2], "xcc": raw[:, 3] & 0xF,
            "rt": u64[:, :5].astype(np.int64), "clk": u64[:, 5:7].astype(np.int64)}


def pct(a, qs=(0, 10, 50, 90, 100)):
    a = np.asarray(a, dtype=np.float64)
    return {f"p{q}": round(float(np.percentile(a, q)), 3) for q in qs} | {"mean": round(float(a.mean()), 3)}


def analyse(rec, slots_per_cu=2, cus=256):
    """Per-kind summary.  Times in microseconds (100 MHz ticks / 100)."""
    out = {}
    order = np.argsort(rec["launch"], kind="stable")
    launches = np.unique(rec["launch"])
    per_launch = []
    for ln in launches:
        m = rec["launch"] == ln
        rt = rec["rt"][m] / 100.0
        kind = min(int(rec["kind"][m].min()), 4)       # kinds 4, 5, 6 = the phases of ONE persistent launch
        t0, t1 = rt[:, 0].min(), rt[:, 4].max()
        per_launch.append({"launch": int(ln), "kind": kind, "n": int(m.sum()), "t0": t0, "t1": t1, "mask": m})
    per_launch.sort(key=lambda d: d["t0"])
    for i, d in enumerate(per_launch):
        d["gap_next"] = per_launch[i + 1]["t0"] - d["t1"] if i + 1 < len(per_launch) else None
        d["prev_kind"] = per_launch[i - 1]["kind"] if i else None
    for kind in sorted(set(d["kind"] for d in per_launch)):
        ls = [d for d in per_launch if d["kind"] == kind]
        span, stagger, prol, loop, epi, drain, life, alive_frac, ramp_frac, tail_frac, mhz = [], [], [], [], [], [], [], [], [], [], []
        xcd_finish, xcd_mhz = [[] for _ in range(8)], [[] for _ in range(8)]
        gaps = [d["gap_next"] for d in ls if d["gap_next"] is not None and d["gap_next"] < 200]
        for d in ls:
            m = d["mask"]
            rt = rec["rt"][m] / 100.0 - d["t0"]
            clk = rec["clk"][m]
            xcc = rec["xcc"][m]
            S = d["t1"] - d["t0"]
            span.append(S)
            stagger.extend(rt[:, 0]); prol.extend(rt[:, 1] - rt[:, 0]); loop.extend(rt[:, 2] - rt[:, 1])
            epi.extend(rt[:, 3] - rt[:, 2]); drain.extend(rt[:, 4] - rt[:, 3]); life.extend(rt[:, 4] - rt[:, 0])
            n = rt.shape[0]
            alive_frac.append(float((rt[:, 4] - rt[:, 0]).sum() / (S * n)))
            ramp_frac.append(float(rt[:, 0].sum() / (S * n)))
            tail_frac.append(float((S - rt[:, 4]).sum() / (S * n)))
            wl = np.maximum(rt[:, 4] - rt[:, 0], 1e-3)
            f = (clk[:, 1] - clk[:, 0]) / wl          # cycles per microsecond = MHz
            mhz.extend(f)
            for x in range(8):
                mx = xcc == x
                if mx.any():
                    xcd_finish[x].append(float(rt[mx, 4].max() / S))
                    xcd_mhz[x].append(float(f[mx].mean()))
        phases = None
        if kind == 4:                                  # per phase: when its tiles start / end inside the launch, tile time
            phases = {}
            for ph in (4, 5, 6):
                st, en, lf = [], [], []
                for d in ls:
                    m = d["mask"] & (rec["kind"] == ph)
                    rt = rec["rt"][m] / 100.0 - d["t0"]
                    st.extend(rt[:, 0]); en.extend(rt[:, 4]); lf.extend(rt[:, 4] - rt[:, 0])
                phases[f"conv{ph - 4}"] = {"tile_start_us": pct(st), "tile_end_us": pct(en), "tile_time_us": pct(lf)}
        out[KIND.get(kind, str(kind))] = {
            "phases": phases,
            "launches": len(ls), "workgroups_per_launch": ls[0]["n"],
            "span_us": pct(span), "gap_to_next_launch_us": pct(gaps) if gaps else None,
            "entry_after_first_entry_us": pct(stagger), "prologue_us": pct(prol), "k_loop_us": pct(loop),
            "epilogue_issue_us": pct(epi), "store_drain_us": pct(drain), "workgroup_life_us": pct(life),
            "share_of_span_x_workgroups": {"alive": round(float(np.mean(alive_frac)), 4), "before_entry": round(float(np.mean(ramp_frac)), 4),
                                           "after_exit": round(float(np.mean(tail_frac)), 4)},
            "shader_mhz_over_workgroup_life": pct(mhz),
            "per_xcd_last_exit_over_span": [round(float(np.mean(v)), 4) if v else None for v in xcd_finish],
            "per_xcd_mhz": [round(float(np.mean(v)), 1) if v else None for v in xcd_mhz],
        }
    # whole recorded window: instrumented spans + gaps vs wall
    t_first, t_last = per_launch[0]["t0"], per_launch[-1]["t1"]
    inside = sum(d["t1"] - d["t0"] for d in per_launch)
    out["_window"] = {"instrumented_launches": len(per_launch), "first_entry_to_last_exit_us": round(t_last - t_first, 1),
                      "sum_of_launch_spans_us": round(inside, 1),
                      "sum_of_gaps_below_20us": round(sum(d["gap_next"] for d in per_launch if d["gap_next"] is not None and d["gap_next"] < 20), 1),
                      "n_gaps_below_20us": sum(1 for d in per_launch if d["gap_next"] is not None and d["gap_next"] < 20)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/wg_timeline")
    ap.add_argument("--plan-flags", type=int, default=0, help="OR-ed into every RDN plan's `reserved` (4 = BINHIP_PLAN_RDB3)")
    ap.add_argument("--warm", type=int, default=12, help="untimed windows before the recorded one (reach the package limit)")
    ap.add_argument("--zero", action="store_true", help="all-zero operands (cycle-bound control)")
    args = ap.parse_args()
    from bin_amd import _lib as L
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.utils import util
    from bin_amd.weights import reference_state_dict, synthetic_frames
    lib = L.lib()
    lib.binhip_set_timeline.restype = C.c_int
    lib.binhip_set_timeline.argtypes = [C.c_void_p, C.c_uint]
    dev = torch.device("cuda")
    net = bin_stage4_lstm()
    net.load_state_dict(reference_state_dict(0), strict=True)
    net = net.to(dev).eval().set_precision("f16x3")
    if args.plan_flags:
        for mod in net.rdn_modules():
            mod.plan_flags = getattr(mod, "plan_flags", 0) | args.plan_flags
    pads = util.pad_sizes(720, 1280)
    frames = [util.replicate_pad(f, pads).to(dev) for f in synthetic_frames(1234, 1, 720, 1280, 6)]
    if args.zero:
        frames = [torch.zeros_like(f) for f in frames]
        with torch.no_grad():
            for p in net.parameters():
                p.zero_()
    cap = 17 * (40 * 504 * 3 + 12 * 1008) + 4096
    buf = torch.zeros(4 + cap * REC_WORDS, dtype=torch.int32, device=dev)
    with torch.no_grad():
        for _ in range(args.warm):
            net(*frames)
        torch.cuda.synchronize()
        lib.binhip_set_timeline(C.c_void_p(buf.data_ptr()), cap)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        net(*frames)
        ev1.record()
        torch.cuda.synchronize()
        used = lib.binhip_set_timeline(C.c_void_p(0), 0)
        ms_instr = ev0.elapsed_time(ev1)
        ev0.record()
        net(*frames)
        ev1.record()
        torch.cuda.synchronize()
        ms_plain = ev0.elapsed_time(ev1)
    host = buf.cpu().numpy().view(np.uint32)
    n = int(min(used, cap))
    raw = host[4:4 + n * REC_WORDS].reshape(n, REC_WORDS)
    raw = raw[raw[:, 0] != 0]                 # (records of launches that did not fit, or of tiles outside the grid, stay zero)
    rec = parse(raw)
    summ = analyse(rec)
    summ["_window"].update({"records": n, "window_ms_recording": round(ms_instr, 3), "window_ms_stamps_off": round(ms_plain, 3),
                            "operands": "zero" if args.zero else "real", "plan_flags": args.plan_flags})
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    np.savez_compressed(args.out + ".npz", raw=raw)
    with open(args.out + ".json", "w") as f:
        json.dump(summ, f, indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
