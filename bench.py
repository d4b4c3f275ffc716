#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: interpolated frames/s at 1280x720 on N MI355X.

A "step" = one 6-frame forward of bin_stage4 (`netG(B1..B11)`, what the reference's test.py runs per
input frame, test.py:249-379) on synthetic 6 x [1,3,720,1280] U[0,1) frames replicate-padded to
[1,3,768,1344] by the test.py rule (:348-366), seeded random-init weights (bin_amd/weights.py), inputs
resident in HBM before the timed region.  One step = one interpolated frame (Ft_p[13]) + two restored
frames.  N>1: every rank runs its own windows (window-sharded inference, no data-path collective;
weak scaling), value = all ranks' windows / max-over-ranks time.

The headline `value` is measured in the fp32-class precision mode "f16x3" (every activation / weight is an fp16
hi+lo pair = 22 significand bits, three MFMA products, fp32 accumulate; 1e-6 max-abs vs the fp32 reference —
the reference computes fp32, RDN.py:141, so a narrower mode is never the headline).  The single-product fp16 mode
("f16", 3.5e-4 max-abs, inside north_star's 1e-3 bar) is reported as `tolerance_mode`, beside `value`.

Extra objects on the JSON line:
  roofline     — dominant kernel (RDB 3x3 conv Cin->32, 70 % of FLOPs): algorithmic bytes per launch AT THE STORAGE
                 WIDTH OF THE TIMED MODE (f16x3: 4 B/element = the fp32 yardstick; f16: 2 B) / its mean duration
                 measured with HIP events on the launch stream; `traffic` from the committed per-precision PMC passes
  cpu_baseline — the oracle (CPU restatement, kind "port") on the host cores: ONE full-size 768x1344 forward
  train        — BASELINE config 3/4 (8 x 256x256 crops per GPU, fwd + Charbonnier + bwd + Adam), own sub-object
  power        — shader clock / package power of the SAME work repeated right after the timed region (nothing samples inside it)
  power_bound  — the same forwards (and training steps) on ALL-ZERO operands right after: same instruction streams, idle
                 datapaths; ms / ms_zero next to the two clocks is the evidence for "this mode sits on the package power cap"
  harness      — N1 end to end: PNG files -> decode -> device u8 kernel -> net -> u8 kernel -> PNG files (bin_amd.test's folder
                 pipeline, f16x3, IO threads beside the GPU work) next to the same schedule with frames resident in HBM
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes); exported on the GPU boxes anyway

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402

H, W = 720, 1280
HBM_PEAK_GBS = 8000.0
# SURVEY.md §8(d) / BASELINE.md §4 constants at 768x1344 padded
FLOP_20, FLOP_17 = 29.386e12, 25.03e12
BYTES_20, BYTES_17 = 304.5e9, 259.0e9


MFMA_PEAK_TF = 2500.0          # dense f16 MFMA peak (MI355X_MICROARCH.md)
BYTES_PER_ELEM = {"f16": 2, "f16x3": 4}          # storage width of an activation / weight element in each mode
PRODUCTS = {"f16": 1, "f16x3": 3}                # MFMA products per algorithmic multiply-add
DTYPE = {"f16x3": "f16x3 (fp16 hi+lo pairs = 22-bit significands, 3 MFMA products, f32 accumulate; fp32-class: 1e-6 vs the fp32 reference)",
         "f16": "f16 MFMA inputs and storage, f32 accumulate (3.5e-4 vs the fp32 reference)"}


def rdb_conv_algorithmic_bytes(n, h2, w2, precision):
    """Layer-wise minimum of ONE launch of the dominant kernel at the STORAGE width of `precision`, averaged over the
    RDB convs it runs (Cin = 96,128,160 -> 32; conv #3, Cin = 192, lives in the fused conv+LFF kernel): read Cin
    planes once, write 32 once, weights once (SURVEY.md §8d per-layer figure; at 4 B/element this IS the fp32 figure)."""
    px = n * h2 * w2
    b = BYTES_PER_ELEM[precision]
    per = [(cin + 32) * b * px + (cin * 32 * 9) * b + 32 * 4 for cin in (96, 128, 160)]
    return sum(per) / len(per)


def rdb_conv_flops(n, h2, w2):
    px = n * h2 * w2
    per = [2.0 * 9 * cin * 32 * px for cin in (96, 128, 160)]
    return sum(per) / len(per)


def rdn_call_model(n_inputs, px):
    """(algorithmic FLOPs, algorithmic bytes at 4 B/element) of ONE RDN call over `px` half-resolution pixels, layer by
    layer as SURVEY.md §8(d) counts them: 2*Cin*Cout*k*k per output pixel; every conv reads its input once and writes
    its output once (concat free, elementwise fused), weights once.  At px = 258 048 this reproduces SURVEY's
    1450.7 / 1465.5 / 1495.3 GFLOP and 15.14 / 15.16 / 15.18 GB."""
    convs = [(5, 12 * n_inputs, 96, 1), (3, 96, 96, 1)]                       # SFENet1, SFENet2   (k, cin, cout, px multiplier)
    for _ in range(12):
        convs += [(3, 96 + 32 * c, 32, 1) for c in range(4)] + [(1, 224, 96, 1)]
    convs += [(1, 1152, 96, 1), (3, 96, 96, 1), (3, 96, 256, 1), (3, 64, 3, 4)]      # GFF.0, GFF.1, UPNet.0, UPNet.2 (full res)
    flops = sum(2.0 * k * k * ci * co * px * m for k, ci, co, m in convs)
    byts = sum(4.0 * (ci + co) * px * m + 4.0 * k * k * ci * co for k, ci, co, m in convs)
    return flops, byts


CALLS_17 = ((2, 5), (3, 6), (5, 6))          # (input frames, calls per 6-frame window) of the exact 17-call schedule


def window_model(px):
    """(FLOPs, bytes) of one 6-frame forward in the 17-call schedule (the ConvLSTM cells' 6 x 1.34 GFLOP at 720p excluded)."""
    f = b = 0.0
    for nin, calls in CALLS_17:
        cf, cb = rdn_call_model(nin, px)
        f += calls * cf
        b += calls * cb
    return f, b


def wgrad3x3_launch_model(batch, h2, w2):
    """Mean algorithmic (FLOPs, bytes at 4 B/element) of ONE launch of the 3x3 weight-gradient kernel in the training mix:
    the four-call schedule runs model1..4 with N = 5B, 6B, 4B, 2B images; each call has 12 x 4 dense-block convs
    (Cin = 96, 128, 160, 192 -> 32): the kernel reads X (Cin planes) and gY (32 planes) once and writes dW."""
    fl = by = 0.0
    cnt = 0
    for n in (5 * batch, 6 * batch, 4 * batch, 2 * batch):
        px = n * h2 * w2
        for cin in (96, 128, 160, 192):
            fl += 2.0 * 9 * cin * 32 * px
            by += 4.0 * (cin + 32) * px + 4.0 * 9 * cin * 32
            cnt += 1
    return fl / cnt, by / cnt


def pmc_traffic(precision):
    """HBM bytes per launch of the dominant kernel IN THIS PRECISION MODE from the committed rocprofv3 PMC passes of
    this same command (profiles/r03_pmc_traffic.json, else r02_: separate FETCH_SIZE / WRITE_SIZE runs per precision, FETCH
    doubled per the gfx950 calibration).  PMC counters cannot be read from inside the process, so bench.py reports the
    profiled figure, or null when no pass for this precision is committed."""
    for name in PMC_FILES:
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                return int(json.load(f)[precision]["traffic_bytes_per_launch"]), "profiles/" + name
        except Exception:
            pass
    return None, None


PMC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")
TRAFFIC_NOTE = ("HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same command (separate FETCH_SIZE / "
                "WRITE_SIZE runs, gfx950 corrections per MI355X_MICROARCH.md); PMC counters cannot be read in-process, so this "
                "is NOT measured by this run — `traffic_source` names the file")


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _host_cores():
    """Cores this process may really use: affinity mask, clipped by the cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline_worker(full=True):
    """(runs in a subprocess) Oracle forward on the host cores.  SURVEY §8(d): C2 at 768x1344.  One full-size 6-frame
    forward of the reference's literal 20-call schedule (no extrapolation; about two minutes on 16 cores), after a
    warm-up on a 48x96 crop; also a 192x336 crop (1/16 of the area) scaled by the pixel ratio, reported beside it to
    show how much a small-crop extrapolation flatters the CPU (cache-resident activations)."""
    from oracle import rdn_oracle as O
    from bin_amd.utils import util
    from bin_amd.weights import canonical_weights, synthetic_frames
    cores = min(_host_cores(), 64)
    torch.set_num_threads(cores)
    canon = {k: torch.from_numpy(v) for k, v in canonical_weights(0).items()}
    with torch.no_grad():
        O.bin_stage4_forward(synthetic_frames(1, 1, 48, 96, 6), canon)          # warm-up (thread pool, oneDNN)
        hs, ws = 192, 336
        frames = synthetic_frames(1234, 1, hs, ws, 6)
        t0 = time.time()
        O.bin_stage4_forward(frames, canon)
        t_crop = time.time() - t0
        crop_fps = (hs * ws) / (768.0 * 1344.0) / t_crop
        res = {"unit": "interpolated frames/s at 1280x720", "cores": cores, "kind": "port", "cpu": cpu_model(),
               "crop_extrapolation": {"value": round(crop_fps, 6),
                                      "note": f"{hs}x{ws} crop = {t_crop:.2f} s scaled by the pixel ratio; flatters the CPU"}}
        print("CPU_BASELINE_PARTIAL " + json.dumps(dict(res, value=round(crop_fps, 6), sample="crop extrapolation only "
                                                        "(the full-size forward did not finish in time)")), flush=True)
        if full:
            padded = [util.replicate_pad(f, util.pad_sizes(H, W)) for f in synthetic_frames(1234, 1, H, W, 6)]
            t0 = time.time()
            ref = O.bin_stage4_forward(padded, canon)
            t_full = time.time() - t0
            res["value"] = round(1.0 / t_full, 6)
            hip_path = os.environ.get("BIN_AMD_BENCH_HIP_OUT")
            if hip_path and os.path.exists(hip_path):
                # the oracle as the CHECKER of the very window the headline timed (same seeded frames, same seeded weights): BASELINE's
                # metric reads "... ; PSNR vs ref" — this is that figure, live, at full size, beside the tests' fixtures
                res["hip_vs_oracle"] = compare_with_oracle(torch.load(hip_path), ref, O, util)
            alt_path = os.environ.get("BIN_AMD_BENCH_ALT_OUT")
            if alt_path and os.path.exists(alt_path):          # the line's `tolerance_mode` (or `fp32_class`) run of the same window
                res["other_mode_vs_oracle"] = compare_with_oracle(torch.load(alt_path), ref, O, util)
                res["other_mode_vs_oracle"]["note"] = "the same check for the precision mode reported beside the headline (tolerance_mode / fp32_class)"
            res["sample"] = (f"oracle (PyTorch-CPU restatement of the reference, literal 20-call schedule): ONE full-size "
                             f"6-frame forward at 768x1344 = {t_full:.1f} s on {cores} threads of {res['cpu']} "
                             f"(no extrapolation)")
    return res


def compare_with_oracle(hip, ref, O, util):
    """(cpu_baseline leg only) the 14 outputs of the timed window against the oracle's, on the 720 x 1280 crop test.py writes."""
    pads = util.pad_sizes(H, W)
    worst, mse, diff_px, npx = 0.0, 0.0, 0, 0
    psnr_u8 = []
    for a, b in zip(hip, ref):
        a, b = a.float(), b.float()
        worst = max(worst, float((a - b).abs().max()))
        ac = a[..., pads[2]:pads[2] + H, pads[0]:pads[0] + W].clamp(0, 1)
        bc = b[..., pads[2]:pads[2] + H, pads[0]:pads[0] + W].clamp(0, 1)
        mse += float(((ac - bc).double() ** 2).mean())
        ia, ib = O.tensor2img(ac[0]), O.tensor2img(bc[0])
        diff_px += int((ia != ib).sum())
        npx += ia.size
        p = O.calculate_psnr(ia, ib)
        psnr_u8.append(None if p == float("inf") else round(float(p), 2))
    mse /= len(ref)
    return {"outputs": len(ref), "max_abs": worst,
            "psnr_db_float": (None if mse == 0 else round(10.0 * math.log10(1.0 / mse), 2)),
            "u8_values_differing": diff_px, "u8_values": npx,
            "psnr_db_u8_worst": (None if all(v is None for v in psnr_u8) else min(v for v in psnr_u8 if v is not None)),
            "note": "the 14 outputs of the timed 720p window (HIP, headline precision) against the oracle's outputs on the same seeded frames and "
                    "weights; psnr_db_float = 10 log10(1 / mean squared difference) over the cropped, clamped outputs; u8 = after the "
                    "reference's tensor2img (round to uint8): values that differ, worst per-image PSNR (null = every image identical)"}


def cpu_baseline(timeout_s=420, hip_outputs=None, alt_outputs=None):
    """Run the CPU baseline in a child process with a hard timeout so it can never stall the bench; if the full-size
    forward does not finish, the crop extrapolation it printed first is reported (and labelled as such)."""
    import subprocess
    code = ("import json,sys; sys.path.insert(0, %r); import bench; "
            "print('CPU_BASELINE ' + json.dumps(bench.cpu_baseline_worker()), flush=True)" % REPO)
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""
    if hip_outputs:
        env["BIN_AMD_BENCH_HIP_OUT"] = hip_outputs
    if alt_outputs:
        env["BIN_AMD_BENCH_ALT_OUT"] = alt_outputs
    out_txt, err_txt = "", ""
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s, env=env)
        out_txt, err_txt = out.stdout, out.stderr
    except subprocess.TimeoutExpired as e:
        out_txt = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        err_txt = f"cpu baseline exceeded {timeout_s}s and was cut"
    partial = None
    for ln in out_txt.splitlines():
        if ln.startswith("CPU_BASELINE "):
            return json.loads(ln[len("CPU_BASELINE "):])
        if ln.startswith("CPU_BASELINE_PARTIAL "):
            partial = json.loads(ln[len("CPU_BASELINE_PARTIAL "):])
    if partial is not None:
        return partial
    return {"value": None, "unit": "interpolated frames/s at 1280x720", "cores": _host_cores(), "kind": "port",
            "sample": "cpu baseline failed: " + (err_txt.strip().splitlines() or ["?"])[-1][:200]}


CLOCK_EXPLAINED_TOL = 0.03        # |cycles_data / cycles_zero - 1| up to which the time ratio counts as explained by the clock
AT_CAP_PPT_FRAC = 0.5             # share of the firmware's samples with the PPT (package power) limiter active = "at the cap"
AT_CAP_POWER_FRAC = 0.9           # fallback without limiter data: mean watts against the nominal cap


def power_bound_reading(ms, ms_zero, power, zero):
    """COMPUTE what the zero-operand control says (this used to be a constant string).  Inputs: the timed
    ms of the real-data run, the ms of the all-zero run, and the two sampler summaries.  Returns the fields that go on the line:
      cycles_data / cycles_zero (M cycles = ms x the mean shader clock of each pass), cycle_ratio,
      limiter (what the device says throttled the real-data pass), at_cap + how that was decided,
      reading = one of
        "not at the cap: ..."        the device did not report the power limiter (or, without limiter data, mean watts < 0.9 cap)
        "clock-explained: ..."       at the cap, and the real-data run needs the same cycles as the zero run (within 3 %):
                                     time ratio = clock ratio, the instruction schedule is not what limits
        "not clock-explained: ..."   at the cap, but real data costs more (or fewer) CYCLES than zeros at the sampled clocks;
                                     the other sampled domains (memory / fabric clock, slowest XCD) are listed as candidates
        "undetermined: ..."          a clock or a time is missing
    """
    def mean(d, k):
        v = (d or {}).get(k)
        return v.get("mean") if isinstance(v, dict) else None
    out = {}
    ck_d, ck_z = mean(power, "clock_mhz"), mean(zero, "clock_mhz")
    pw_d = mean(power, "power_w")
    lim = (power or {}).get("limiter") or {}
    fr = lim.get("active_frac") or {}
    out["limiter"] = {"source": lim.get("source", "unavailable"), "active_frac": fr or None, "dominant": lim.get("dominant"),
                      "xcd_below_host_limit_ppt_frac": (lim.get("xcd_below_host_limit_ppt_frac") or {}).get("mean"),
                      "zero_pass_active_frac": ((zero or {}).get("limiter") or {}).get("active_frac")}
    cap = (power or {}).get("power_cap_w")
    pmax = ((power or {}).get("power_w") or {}).get("max") if isinstance((power or {}).get("power_w"), dict) else None
    out["power_cap_w"] = cap
    out["power_cap_observed_w"] = pmax               # the highest socket power any sample of the real-data pass saw
    out["power_from_energy_w"] = {"data": (power or {}).get("power_from_energy_w"), "zero": (zero or {}).get("power_from_energy_w")}
    if "ppt_power" in fr:
        at_cap = fr["ppt_power"] >= AT_CAP_PPT_FRAC
        how = f"device: PPT limiter active in {fr['ppt_power']:.0%} of the firmware's samples (threshold {AT_CAP_PPT_FRAC:.0%})"
    elif pw_d is not None and cap:
        at_cap = pw_d >= AT_CAP_POWER_FRAC * cap
        how = f"no limiter data: mean {pw_d:.0f} W against {AT_CAP_POWER_FRAC} x the nominal cap {cap:.0f} W"
    else:
        at_cap, how = None, "no limiter data and no power samples"
    out["at_cap"], out["at_cap_rule"] = at_cap, how
    # Which clock?  amdsmi's GFX `clk` is ONE XCD's clock — XCD 0 (measured round 5: the nearest XCD in 140 of 140 samples) — and under
    # the package limit the eight XCDs settle 5-8 % apart (even-numbered ones ~5 % above odd-numbered ones on the boxes seen), and every XCD runs its own band of tiles, so the device-wide mean
    # (and, for the tail of a launch, the slowest XCD) is what the work sees.  Cycles are counted at the per-XCD MEAN when
    # the gpu_metrics table has it, at `clk` otherwise; both figures are on the line.
    xd, xz = (power or {}).get("xcd_clock_mhz"), (zero or {}).get("xcd_clock_mhz")
    use_xcd = bool(xd and xz and xd.get("mean") and xz.get("mean"))
    fd, fz = (xd["mean"], xz["mean"]) if use_xcd else (ck_d, ck_z)
    out["cycles_clock"] = ("per-XCD mean shader clock (gpu_metrics current_gfxclks)" if use_xcd
                           else "amdsmi GFX clk (= XCD 0 only; no per-XCD table on this box)")
    if not (fd and fz and ms and ms_zero):
        out.update(cycles_data_M=None, cycles_zero_M=None, cycle_ratio=None,
                   reading="undetermined: a clock or a time is missing (no smi source on this box?)")
        return out
    cyc_d, cyc_z = ms * fd * 1e-3, ms_zero * fz * 1e-3              # ms x MHz = 1e3 cycles
    cr = cyc_d / cyc_z
    out.update(cycles_data_M=round(cyc_d, 2), cycles_zero_M=round(cyc_z, 2), cycle_ratio=round(cr, 4))
    other = {}
    for k in ("mem_clock_mhz", "fabric_clock_mhz"):
        a, b = mean(power, k), mean(zero, k)
        if a and b:
            other[k] = {"data": a, "zero": b, "zero_over_data": round(b / a, 4)}
    if use_xcd:
        other["xcd_clock_mhz"] = {"data": xd, "zero": xz,
                                  "cycle_ratio_at_gfx_clk": None if not (ck_d and ck_z) else round(ms * ck_d / (ms_zero * ck_z), 4),
                                  "cycle_ratio_at_slowest_xcd": round(ms * xd["slowest_xcd_mean"] / (ms_zero * xz["slowest_xcd_mean"]), 4)}
    out["other_domains"] = other or None
    tr, clr = ms / ms_zero, fz / fd
    out["clock_ratio_used"] = round(clr, 4)
    # if the real-data run needs FEWER cycles than the zero run, part of it does not scale with the shader clock (HBM-bound
    # kernels, host gaps): the share s of the zero run's time that must be clock-independent for tr = s + (1 - s) * clr
    share = None
    if clr > 1.0 and tr < clr:
        share = round((clr - tr) / (clr - 1.0), 4)
    out["clock_independent_share_implied"] = share
    if at_cap is False:
        out["reading"] = (f"not at the cap: {how}; time ratio {tr:.3f}, clock ratio {clr:.3f}, cycle ratio {cr:.3f} - the zero-operand "
                          "speed-up cannot be attributed to the package power limit on this box")
    elif abs(cr - 1.0) <= CLOCK_EXPLAINED_TOL:
        out["reading"] = (f"clock-explained: {how}; real data and zeros need the same cycles ({cyc_d:.1f} M vs {cyc_z:.1f} M, ratio "
                          f"{cr:.3f}), so the time ratio {tr:.3f} is the clock ratio {clr:.3f}: the run is limited by the clock the "
                          "limiter leaves, not by its instruction schedule")
    else:
        cand = []
        for k, v in other.items():
            if k != "xcd_clock_mhz" and abs(v["zero_over_data"] - 1.0) > 0.01:
                cand.append(f"{k} {v['data']:.0f} vs {v['zero']:.0f}")
        if "xcd_clock_mhz" in other:
            x = other["xcd_clock_mhz"]
            cand.append(f"cycle ratio at the GFX clk (XCD 0) {x['cycle_ratio_at_gfx_clk']}, at the slowest XCD "
                        f"{x['cycle_ratio_at_slowest_xcd']:.3f}")
        if cr < 1.0 and share is not None:
            cand.append(f"fewer cycles on real data = a clock-independent share of the run (HBM-bound kernels, gaps): implied "
                        f"{share:.0%} of the zero run's time")
        out["reading"] = (f"not clock-explained: {how}; real data takes {(cr - 1) * 100:+.1f} % cycles against zeros ({cyc_d:.1f} M vs "
                          f"{cyc_z:.1f} M; time ratio {tr:.3f}, clock ratio {clr:.3f})"
                          + ("; " + "; ".join(cand) if cand else "; no other sampled domain differs"))
    return out


def power_bound_object(ms, power, zero, what):
    """`power`: sampler summary of the real-data power pass, `zero`: of the all-zero-operand pass (both carry ms_per_repetition)."""
    def mean(d, k):
        v = (d or {}).get(k)
        return v.get("mean") if isinstance(v, dict) else None
    ms_zero = zero.get("ms_per_repetition")
    ms_data = power.get("ms_per_repetition")
    out = {"what": f"{what}: the same launches on all-zero operands (same instruction streams, idle datapaths), right after the "
                   "timed region",
           "ms": round(ms, 3), "ms_data_pass": ms_data, "ms_zero": ms_zero,
           "ratio": None if not ms_zero else round(ms / ms_zero, 4),
           "clock_mhz": {"data": mean(power, "clock_mhz"), "zero": mean(zero, "clock_mhz"),
                         "note": "amdsmi GFX clk = XCD 0 only, one of the faster XCDs; the per-XCD mean / slowest are in other_domains.xcd_clock_mhz"},
           "clock_ratio": (None if not (mean(power, "clock_mhz") and mean(zero, "clock_mhz"))
                           else round(mean(zero, "clock_mhz") / mean(power, "clock_mhz"), 4)),
           "power_w": {"data": mean(power, "power_w"), "zero": mean(zero, "power_w")},
           "energy_j_per_repetition": {"data": power.get("energy_j_per_repetition"), "zero": zero.get("energy_j_per_repetition")}}
    out.update(power_bound_reading(ms, ms_zero, power, zero))
    out["zero_pass_repetitions"] = zero.get("repetitions")
    return out


def step_spread(ms_list):
    """min / median / max of the per-step device times of a timed region (one HIP event per step boundary on the launch stream):
    a single stalled step is invisible in `ms_per_step` = total / steps."""
    v = sorted(float(x) for x in ms_list)
    if not v:
        return None
    med = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
    return {"ms_min": round(v[0], 3), "ms_median": round(med, 3), "ms_max": round(v[-1], 3),
            "max_over_median": round(v[-1] / med, 4) if med else None, "steps": len(v)}


def clock_fields(ms, power):
    """Box-independent yardstick: cycles per step = ms x the per-XCD mean shader clock of the power pass that
    follows the timed region (boxes hold 1.6-1.9 GHz at the package limit; the cycle count moves far less than the milliseconds)."""
    x = (power or {}).get("xcd_clock_mhz") or {}
    f = x.get("mean") or ((power or {}).get("clock_mhz") or {}).get("mean")
    return {"xcd_clock_mhz_mean": f, "cycles_per_step_M": None if not (f and ms) else round(ms * f * 1e-3, 2),
            "cycles_clock": "per-XCD mean (gpu_metrics)" if x.get("mean") else "amdsmi GFX clk (XCD 0 only)"}


def timed_steps(fn, steps, sync_all):
    """The contract's timed region (EXACTLY `steps` calls of fn, closed by sync_all) plus one HIP event per step boundary."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        fn(i)
        ev[i + 1].record()
    sync_all()
    dt = time.perf_counter() - t0
    return dt, [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]


def make_train_model(prec, bwd_prec, world, rank, B, zero_data=False, S=256):
    """The training model of --mode train (BASELINE config 3/4) with one synthetic batch fed; also used by tools/stall_hunt.py."""
    import tempfile
    from bin_amd.models import create_model
    from bin_amd.weights import reference_state_dict
    tmp = tempfile.mkdtemp()
    opt = {"model": "bin", "gpu_ids": [0], "is_train": True, "dist": world > 1,
           "network_G": {"which_model_G": "bin_stage4", "nframes": 6, "version": 2, "precision": prec,
                         "backward_precision": bwd_prec},
           "path": {"pretrain_model_G": None, "strict_load": True, "models": tmp, "training_state": tmp},
           "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "weight_decay_G": 0, "ft_tsa_only": None,
                     "lr_G": 1e-4, "beta1": 0.9, "beta2": 0.99, "lr_scheme": "MultiStepLR", "lr_steps": [100000],
                     "restarts": None, "restart_weights": None, "lr_gamma": 0.5, "clear_state": False}}
    m = create_model(opt)
    m.netG.module.load_state_dict(reference_state_dict(0), strict=True)
    if prec == "f16":                        # timing diagnostic only: single-product training is gated (bin_amd/autograd.py)
        for mod in m.netG.module.rdn_modules():
            mod.allow_f16_training = True
    g = torch.Generator().manual_seed(7 + rank)
    batch = {"LQs": torch.rand(B, 6, 3, S, S, generator=g), "GTenh": torch.rand(B, 6, 3, S, S, generator=g),
             "GTinp": torch.rand(B, 5, 3, S, S, generator=g)}
    if zero_data:                            # diagnostic (invalid as a result): the same launches on all-zero operands
        batch = {k: torch.zeros_like(v) for k, v in batch.items()}
        with torch.no_grad():
            for prm in m.netG.module.parameters():
                prm.zero_()
    m.feed_data(batch)
    return m, batch


def make_train_step(batch=8, precision="f16x3"):
    m, _ = make_train_model(precision, None, 1, 0, batch)
    n = [0]

    def step():
        n[0] += 1
        m.optimize_parameters(n[0])
    return step


def make_infer_step(precision="f16x3"):
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.utils import util
    from bin_amd.weights import reference_state_dict, synthetic_frames
    dev = torch.device("cuda", torch.cuda.current_device())
    net = bin_stage4_lstm()
    net.load_state_dict(reference_state_dict(0), strict=True)
    net = net.to(dev).eval().set_precision(precision)
    frames = [util.replicate_pad(f, util.pad_sizes(H, W)).to(dev) for f in synthetic_frames(1234, 1, H, W, 6)]

    def step():
        with torch.no_grad():
            net(*frames)
    return step


def max_over_ranks(dt, dev, world):
    """Slowest rank's time (the contract's max-over-ranks).  nccl (= RCCL) reduces on the device; the gloo hook used to run
    the N > 1 flow on ONE GPU (tests/test_gpu_dist.py) reduces a host tensor."""
    import torch.distributed as dist
    if world == 1:
        return dt
    host = dist.get_backend() != "nccl"
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if host else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


HARNESS_YML = """\
name: bench_harness
model: bin
distortion: blur
scale: 4
gpu_ids: [0]
use_tb_logger: false
datasets:
  val:
    name: test
    mode: BIN
    dataroot_GT: {tmp}
    dataroot_LQ: {tmp}
network_G:
  which_model_G: bin_stage4
  nframes: 6
  version: 2
path:
  pretrain_model_G: ~
  save_path: {tmp}
  strict_load: true
  resume_state: ~
"""


def resident_clip_fps(net, frames, n_frames):
    """The harness leg's OWN workload with the IO taken away: the same n_frames - 1 windows of one clip (same frame ids per
    window incl. the clamped clip edges, same cross-window reuse, so the first window is a full 17-call forward like the folder
    run's), frames resident in HBM as padded fp32 tensors, the three u8 output kernels per window included — no decode, no
    upload, no D2H, no encode.  `harness.io_overlap_frac` = folder rate / this rate."""
    from bin_amd import harness, ops
    from bin_amd.utils import util
    clip = [frames[i % len(frames)].clone() for i in range(n_frames)]        # distinct tensors: the reuse memo keys on identity
    l, r, t, b = util.pad_sizes(H, W)
    cache = {}
    with torch.no_grad():
        net(*[clip[k] for k in harness.window_frame_ids(0, n_frames)])          # warm (no cache: nothing is reused from it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_frames - 1):
            out = net(*[clip[k] for k in harness.window_frame_ids(i, n_frames)], stage1_cache=cache)
            for k in (13, 8, 12):
                ops.frame_to_u8(out[k], t, l, H, W)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    del clip, cache
    return (n_frames - 1) / dt


def harness_bench(precision, n_frames, gpu_only_fps, io_threads=12, steady_gpu_only_fps=None):
    """SURVEY 8f N1 end to end, what a user of the reference's test.py runs (test.py:334-402): a folder of 720p PNG frames in, the
    interpolated + deblurred PNG frames out, through `bin_amd.test` (decode on a thread pool -> device u8 kernel + padding ->
    net with the exact stage-1 reuse -> device clamp/round/crop kernel -> D2H on a copy stream -> PNG encode on the pool).
    The frames are written to TMPDIR BEFORE anything is timed (smooth seeded images + mild noise, so the PNG codec sees
    realistic entropy); the folder is run twice into fresh output folders, the second pass (warm kernels, page cache) is the
    one reported; its wall time includes every decode, copy, encode and file write.  `gpu_only` = the same schedule with the
    frames resident in HBM (this line's `streaming` leg, same precision)."""
    import shutil
    import tempfile
    import numpy as np
    from PIL import Image
    from bin_amd import test as run_test
    tmp = tempfile.mkdtemp(prefix="bin_amd_harness_")
    try:
        g = np.random.Generator(np.random.PCG64(1))
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        clip = os.path.join(tmp, "test_blur", "clip0")
        os.makedirs(clip)
        noise = g.normal(0, 2.0, (H, W, 3)).astype(np.float32)

        def make(k):
            img = np.stack([127 + 100 * np.sin((xx + 9 * k) / 37.0 + c) * np.cos((yy - 5 * k) / 53.0 - c) for c in range(3)], -1)
            img = (img + np.roll(noise, 7 * k, axis=1)).clip(0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(clip, f"{8 * k:05d}.png"), compress_level=1)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=io_threads) as pool:      # (untimed set-up; an Adobe240 clip has 63-418 frames)
            list(pool.map(make, range(n_frames)))
        yml = os.path.join(tmp, "o.yml")
        with open(yml, "w") as f:
            f.write(HARNESS_YML.format(tmp=tmp))
        stats = None
        for rep in range(2):
            stats = {}
            run_test.main(["--input_path", os.path.join(tmp, "test_blur"), "--output_path", os.path.join(tmp, f"out{rep}"),
                           "--opt", yml, "--precision", precision, "--io_threads", str(io_threads)], stats=stats)
        n_png = sum(1 for _, _, fs in os.walk(os.path.join(tmp, "out1")) for x in fs if x.endswith(".png"))
        fps = stats["windows"] / stats["wall"]
        st = stats.get("stamps") or []
        steady = None
        if len(st) >= 24:               # the middle half of the clip: the host is throttled by the bounded writer queue, so the
            a, b = len(st) // 4, (3 * len(st)) // 4          # rate at which it QUEUES windows there is the pipeline's own rate
            steady = (b - a) / max(st[b] - st[a], 1e-9)
        return {"frames_per_s": round(fps, 3), "unit": "interpolated frames/s, PNG files in -> PNG files out",
                "steady_state_frames_per_s": None if steady is None else round(steady, 3),
                "gpu_only_frames_per_s": None if gpu_only_fps is None else round(gpu_only_fps, 3),
                "gpu_only_steady_state_frames_per_s": None if steady_gpu_only_fps is None else round(steady_gpu_only_fps, 3),
                "io_overlap_frac": None if not gpu_only_fps else round(fps / gpu_only_fps, 4),
                "windows": stats["windows"], "input_frames": n_frames, "wall_s": round(stats["wall"], 3),
                "timeline_s": stats.get("timeline"),
                "net_and_glue_ms_per_window": round(stats["net_s_per_window"] * 1e3, 2), "png_files_written": n_png,
                "precision": precision, "io_threads": io_threads,
                "note": "second of two passes over the same folder into a fresh output folder; wall time covers decode, H2D, the "
                        "u8 kernels, the net (10 RDN calls per window: exact reuse of every LSTM-free call of the previous window), D2H, PNG encode and file writes; "
                        "gpu_only = the SAME clip's windows with the frames resident in HBM (bench.resident_clip_fps: same frame ids, same reuse, first window "
                        "a full 17-call forward, u8 output kernels included; no decode / upload / D2H / encode); gpu_only_steady_state = the `streaming` leg of this "
                        "line (10 calls per window throughout)"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def train_bench(args, rank, world, dev, steps=None, warmup=None, standalone=True):
    """Secondary metric (BASELINE.json config 4): training samples/s, one process per GPU, DP gradient all-reduce.
    standalone=False: called from the default inference run (rank 0, N = 1) to put a driver-timed training figure on
    the same JSON line; returns the dict instead of printing it."""
    import torch.distributed as dist
    prec = args.train_precision
    bwd_prec = None
    if prec == "mixed":                      # fp32-class forward (exact loss / ReLU masks), single-product backward
        prec, bwd_prec = "f16x3", "f16"
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    m, batch = make_train_model(prec, bwd_prec, world, rank, args.batch, args.zero_data)
    B, S = args.batch, 256

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from bin_amd import ops, _lib as L
    from bin_amd.utils.smi import power_pass
    torch.cuda.reset_peak_memory_stats()
    for i in range(warmup):
        m.optimize_parameters(i + 1)
    sync_all()
    dt, step_ms = timed_steps(lambda i: m.optimize_parameters(warmup + i + 1), steps, sync_all)
    ops.check_status()                       # no activation / gradient left the fp16 storage range
    loss_value = float(m.loss.detach())
    # clock / package power of the same steps, sampled in a pass of its own (every rank runs it: collectives stay matched)
    counter = [warmup + steps]

    def one_step():
        counter[0] += 1
        m.optimize_parameters(counter[0])
    # (a FIXED number of steps: every rank must issue the same collectives)
    power = power_pass(one_step, dev, repetitions=(1 if args.no_power else 6), sync=torch.cuda.synchronize)
    # ---- roofline of the step and of its dominant kernel (3x3 weight gradient): two more steps AFTER the timed region
    # with the weight-gradient launches bracketed by HIP events on the stream they run on (BINHIP_PROF_WGRAD)
    # Two passes of two steps: (a) as in the timed steps (weight gradients on the side stream, sharing the chip with the
    # backward-data chain), (b) with the side stream off, so that a kernel's duration is its own — the figure a
    # `rocprofv3 --kernel-trace` of `BIN_AMD_WGRAD_STREAM=0 bench.py --mode train` reports (profiles/).
    kern, kern_overlapped = None, None
    lib = L.lib()
    net = m.netG.module
    prof_steps = 2
    step_no = counter[0]
    for exclusive in (False, True):
        handle = ctypes.c_void_p(0)
        if rank == 0:                        # every rank runs the extra steps (collectives stay matched); rank 0 times them
            L.check(lib.binhip_profiler_create(3, 32, L.PROF_WGRAD, prof_steps * 4 * 48, ctypes.byref(handle)), "profiler_create")
            net.set_profiler(handle, backward=True)
        saved = [mod.wgrad_side_stream for mod in net.rdn_modules()]
        if exclusive:
            for mod in net.rdn_modules():
                mod.wgrad_side_stream = False
        for i in range(prof_steps):
            step_no += 1
            m.optimize_parameters(step_no)
        torch.cuda.synchronize()
        for mod, v in zip(net.rdn_modules(), saved):
            mod.wgrad_side_stream = v
        if rank == 0:
            net.set_profiler(None, backward=True)
            kms, kn = ctypes.c_double(0), ctypes.c_int(0)
            L.check(lib.binhip_profiler_read(handle, ctypes.byref(kms), ctypes.byref(kn)), "profiler_read")
            lib.binhip_profiler_destroy(handle)
            if kn.value:
                if exclusive:
                    kern = (kms.value / kn.value * 1e-3, kn.value)
                else:
                    kern_overlapped = kms.value / kn.value * 1e-3
    dt = max_over_ranks(dt, dev, world)
    # ---- zero-operand control: the SAME steps with all-zero batch and weights — identical launches
    # and instruction streams, idle datapaths.  Last thing this model does (its weights are gone afterwards).
    power_bound = None
    if not args.zero_data and not args.no_power:
        with torch.no_grad():
            for prm in m.netG.module.parameters():
                prm.zero_()
        m.feed_data({k: torch.zeros_like(v) for k, v in batch.items()})
        for _ in range(2):
            one_step()
        torch.cuda.synchronize()
        zero = power_pass(one_step, dev, repetitions=5, sync=torch.cuda.synchronize)
        power_bound = power_bound_object(dt / steps * 1e3, power, zero, "training step")
        ops.check_status()
    line = {
        "metric": "training samples/sec (256x256 crops, 6-frame windows)", "value": round(world * B * steps / dt, 4),
        "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE[prec] + (" forward; single-product backward on the hi planes" if bwd_prec else ""),
        "data": ("all-zero operands (diagnostic, INVALID as a result)" if args.zero_data else "synthetic"),
        "loss": loss_value,
        "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
        "nccl_ranks": (dist.get_world_size() if dist.is_initialized() else 1),
        "backend": (dist.get_backend() if dist.is_initialized() else None),
        "power": power,
        "power_bound": power_bound,
        "step_ms": step_spread(step_ms),
        "roofline": {**train_roofline(B, S, dt / steps, prec, bwd_prec, kern, kern_overlapped), **clock_fields(dt / steps * 1e3, power),
                     "step_ms": step_spread(step_ms)},
        "config": {"workload": f"Adobe240 training, 256x256 crops, batch {B} per GPU, Charbonnier x17, Adam, "
                               f"DP flat gradient all-reduce (45.77 MB)", "precision": prec,
                   "backward_precision": bwd_prec or prec}}
    del m
    torch.cuda.empty_cache()
    if prec == "f16x3" and not bwd_prec and not args.zero_data and world == 1 and not args.no_extras:
        # the supported reduced-precision training mode beside the headline (never instead of it): f16x3 forward, single-product
        # backward — the counterpart of BASELINE config 4's "bf16" wording with a stated parity (profiles/r06_training_modes.md)
        m2, _ = make_train_model("f16x3", "f16", world, rank, args.batch)
        for i in range(2):
            m2.optimize_parameters(i + 1)
        torch.cuda.synchronize()
        dt2, ms2 = timed_steps(lambda i: m2.optimize_parameters(3 + i), 4, torch.cuda.synchronize)
        ops.check_status()
        line["mixed_mode"] = {"value": round(B * 4 / dt2, 4), "unit": "samples/s", "ms_per_step": round(dt2 / 4 * 1e3, 2),
                              "step_ms": step_spread(ms2), "config": {"precision": "f16x3", "backward_precision": "f16"},
                              "parity": "loss and ReLU masks exact (f16x3 forward); parameter gradients ~2e-3 relative vs torch autograd of the "
                                        "oracle (tests/test_gpu_train.py); converges like f16x3 (profiles/r06_training_modes.md)"}
        del m2
        torch.cuda.empty_cache()
    if not standalone:
        return line
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def fused_upnet_active():
    """Is the inference path running UPNet as one 5x5 convolution (BINHIP_PLAN_FUSED_UPNET, the default since round 6)?"""
    from bin_amd import _lib as L_
    from bin_amd.rdn_plan import default_plan_flags
    return bool(default_plan_flags() & L_.PLAN_FUSED_UPNET)


def kernel_roofline(prec, kern_ms, kern_n, hp, wp, ms, reuse_schedule):
    """`roofline` object of the inference window in precision `prec`: the dominant kernel (dense-block 3x3 conv,
    Cin -> 32) from its event-timed mean duration, plus the whole forward's algorithmic rates.  Every `frac` can be
    recomputed from the fields beside it."""
    if kern_n <= 0:
        return None
    flops = FLOP_17 if reuse_schedule else FLOP_20
    abytes = (BYTES_17 if reuse_schedule else BYTES_20) * BYTES_PER_ELEM[prec] / 4.0
    whole_gbs = abytes / (ms * 1e-3) / 1e9
    avg_s = kern_ms / kern_n * 1e-3
    ab = rdb_conv_algorithmic_bytes(1, hp // 2, wp // 2, prec)
    ach = ab / avg_s / 1e9
    mf = rdb_conv_flops(1, hp // 2, wp // 2) * PRODUCTS[prec] / avg_s / 1e12
    assert ach <= HBM_PEAK_GBS, f"kernel algorithmic rate {ach:.0f} GB/s exceeds the HBM peak"
    assert whole_gbs <= HBM_PEAK_GBS, f"whole-forward algorithmic rate {whole_gbs:.0f} GB/s exceeds the HBM peak"
    kname = ("conv_x3_kernel<3,2,8,0,0,false>" if prec == "f16x3" else "conv_mfma_kernel<3,1,1,2,8,1,1,2,0,false>")
    # Which roof?  Arithmetic intensity at the mode's storage width against the ridge of its product count: f16x3 executes 3
    # MFMA products per algorithmic multiply-add, so its ridge is 2500 / 3 TFLOP/s / 8 TB/s = 104 FLOP/B and the kernel's
    # 115 FLOP/B sits ON / above it -> the matrix pipe is the roof, and that pipe runs at the clock the package power cap
    # leaves (`power`, `power_bound`).  f16: 230 FLOP/B against a ridge of 312 -> HBM.
    ai = rdb_conv_flops(1, hp // 2, wp // 2) / ab
    ridge = MFMA_PEAK_TF * 1e12 / PRODUCTS[prec] / (HBM_PEAK_GBS * 1e9)
    traffic, traffic_src = pmc_traffic(prec)
    hbm = {"achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
    mfma = {"achieved": round(mf, 1), "peak": MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(mf / MFMA_PEAK_TF, 4),
            "algorithmic": {"achieved": round(mf / PRODUCTS[prec], 1), "frac": round(mf / PRODUCTS[prec] / MFMA_PEAK_TF, 4),
                            "note": "the conv's own FLOPs (one product per multiply-add) against the same peak: what the emulation of "
                                    "22-bit significands with fp16 products costs is the factor between the two fractions"},
            "note": f"executed MFMA FLOPs = {PRODUCTS[prec]} x the conv's algorithmic FLOPs; peak = nominal dense fp16 "
                    "at 2.4 GHz — scale by power.xcd_clock_mhz.mean / 2400 for the peak at the clock this run held",
            "reference_sustained_at_power_cap": {
                "random_fp16": 1663.0, "zeros": 2473.0, "unit": "TFLOP/s",
                "source": "profiles/r02_power_cap.md (register-resident v_mfma_f32_32x32x16_f16 alone, "
                          "tools/probe_mfma_power.hip); a round-2 measurement, NOT taken in this run — this run's "
                          "own clock, package power and zero-operand control are in `power` / `power_bound`"}}
    mfma_bound = ai >= ridge
    top = mfma if mfma_bound else hbm
    return {"bound": "mfma" if mfma_bound else "hbm",
            "regime": ("mfma@power-cap: arithmetic intensity at or above the ridge of the 3-product scheme; the matrix pipe runs at "
                       "the clock the package power cap leaves (see power / power_bound)") if mfma_bound
                      else "hbm: arithmetic intensity below the ridge",
            "arithmetic_intensity_flop_per_byte": round(ai, 1), "ridge_flop_per_byte": round(ridge, 1),
            "kernel": kname + " (RDB conv3x3 Cin->32 +ReLU, convs 0-2 of each dense block)",
            "precision": prec, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
            "traffic": traffic, "traffic_source": traffic_src, "traffic_note": TRAFFIC_NOTE,
            "avg_kernel_us": round(avg_s * 1e6, 2), "launches": kern_n,
            "algorithmic_bytes_per_launch": int(ab), "bytes_per_element": BYTES_PER_ELEM[prec],
            "algorithmic_flop_per_launch": int(rdb_conv_flops(1, hp // 2, wp // 2)),
            "hbm": hbm, "mfma": mfma,
            "whole_forward": whole_forward_rates(prec, flops, abytes, ms, hp, wp, reuse_schedule)}


def whole_forward_rates(prec, flops, abytes, ms, hp, wp, reuse_schedule):
    """Whole-window rates.  `algorithmic_*` = SURVEY 8(d)'s layer-wise count of the REFERENCE's layer list; with the fused UPNet the
    window executes less than that (one 5x5 convolution 96 -> 12 at half resolution instead of conv3x3 96 -> 256 + conv3x3 64 -> 3 at full
    resolution, and no 64-channel full-resolution intermediate), so the fractions are computed from what is EXECUTED."""
    calls = 17 if reuse_schedule else 20
    px = (hp // 2) * (wp // 2)
    ex_flops, ex_bytes, fused = flops, abytes, fused_upnet_active()
    if fused:
        ex_flops -= calls * px * 2.0 * (9 * 96 * 256 + 4 * 9 * 64 * 3 - 25 * 96 * 12)
        ex_bytes -= calls * px * 4 * 64 * BYTES_PER_ELEM[prec] * 2.0              # the intermediate's write and read
    gbs = ex_bytes / (ms * 1e-3) / 1e9
    assert gbs <= HBM_PEAK_GBS
    return {"algorithmic_GB": round(abytes / 1e9, 1), "algorithmic_TFLOP": round(flops / 1e12, 3),
            "executed_GB": round(ex_bytes / 1e9, 1), "executed_TFLOP": round(ex_flops / 1e12, 3), "fused_upnet": fused,
            "achieved_GBs": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4),
            "achieved_TFLOPs": round(ex_flops / (ms * 1e-3) / 1e12, 1),
            "mfma_frac": round(ex_flops * PRODUCTS[prec] / (ms * 1e-3) / 1e12 / MFMA_PEAK_TF, 4),
            "note": "rates and fractions from the EXECUTED work (UPNet = conv3x3 -> PixelShuffle -> conv3x3 without an activation runs as one "
                    "5x5 convolution on 12 sub-pixel channels: same function, 3.4 x fewer multiply-adds); algorithmic_* = the reference's layer list"}


def train_roofline(batch, size, step_s, prec, bwd_prec, kern, kern_overlapped=None):
    """Roofline object of the training step: algorithmic work = 3 x the forward's (forward + backward-data + backward-
    weight, each the same MACs and the same activation bytes; SURVEY §8d "training step ~ 3 x forward"), executed MFMA
    FLOPs = products x algorithmic (the single-product backward of the mixed mode executes 1 x on two thirds of it).
    `kern` = (mean seconds, launches) of the dominant kernel, the 3x3 weight gradient, from HIP events."""
    h2 = w2 = size // 2
    f_fwd, b_fwd = window_model(h2 * w2)
    flops = 3.0 * f_fwd * batch
    byts = 3.0 * b_fwd * batch * BYTES_PER_ELEM[prec] / 4.0
    prod_f, prod_b = PRODUCTS[prec], (1 if bwd_prec else PRODUCTS[prec])
    executed = (prod_f + 2.0 * prod_b) / 3.0 * flops
    out = {"algorithmic_TFLOP_per_step": round(flops / 1e12, 2), "algorithmic_GB_per_step": round(byts / 1e9, 1),
           "note": "3 x the 17-call forward of one 6-frame sample (SURVEY 8d layer model: the reference's layer list) x batch; bytes at the storage "
                   "width.  The step executes somewhat less: UPNet runs fused, forward and backward (one 5x5 convolution on 12 sub-pixel channels "
                   "instead of conv3x3 96->256 + conv3x3 64->3: 8 % of a call's multiply-adds become 1 %), so the whole-step rates are "
                   "reference-equivalent work per second, an upper reading of the executed rate",
           "mfma": {"achieved": round(executed / step_s / 1e12, 1), "peak": MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": round(executed / step_s / 1e12 / MFMA_PEAK_TF, 4),
                    "algorithmic_TFLOPs": round(flops / step_s / 1e12, 1)},
           "hbm": {"achieved": round(byts / step_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(byts / step_s / 1e9 / HBM_PEAK_GBS, 4)}}
    if kern:
        avg_s, n = kern
        kf, kb = wgrad3x3_launch_model(batch, h2, w2)
        kb *= BYTES_PER_ELEM[prec] / 4.0
        wg_traffic, wg_src = pmc_wgrad_traffic()
        wg_ai, wg_ridge = kf / kb, MFMA_PEAK_TF * 1e12 / prod_b / (HBM_PEAK_GBS * 1e9)
        out["dominant_kernel"] = {
            "kernel": "wgrad3x3_xrow_kernel (3x3 weight gradient of the dense-block convs, Cin = 96..192 -> 32)",
            "avg_kernel_us": round(avg_s * 1e6, 2), "launches": n, "launches_per_step": 4 * 48,
            "share_of_step": round(avg_s * 4 * 48 / step_s, 4),
            "avg_kernel_us_beside_backward_data": None if kern_overlapped is None else round(kern_overlapped * 1e6, 2),
            "algorithmic_bytes_per_launch": int(kb),
            "bound": "mfma" if wg_ai >= wg_ridge else "hbm",
            "regime": "mfma@power-cap" if wg_ai >= wg_ridge else "hbm",
            "arithmetic_intensity_flop_per_byte": round(wg_ai, 1), "ridge_flop_per_byte": round(wg_ridge, 1),
            "achieved": round((kf * prod_b / avg_s / 1e12) if wg_ai >= wg_ridge else (kb / avg_s / 1e9), 1),
            "peak": MFMA_PEAK_TF if wg_ai >= wg_ridge else HBM_PEAK_GBS, "unit": "TFLOP/s" if wg_ai >= wg_ridge else "GB/s",
            "frac": round((kf * prod_b / avg_s / 1e12 / MFMA_PEAK_TF) if wg_ai >= wg_ridge else (kb / avg_s / 1e9 / HBM_PEAK_GBS), 4),
            "hbm": {"achieved": round(kb / avg_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(kb / avg_s / 1e9 / HBM_PEAK_GBS, 4)},
            "traffic": wg_traffic, "traffic_source": wg_src,
            "mfma": {"achieved": round(kf * prod_b / avg_s / 1e12, 1), "peak": MFMA_PEAK_TF, "unit": "TFLOP/s",
                     "frac": round(kf * prod_b / avg_s / 1e12 / MFMA_PEAK_TF, 4)},
            "timing": "HIP event pairs around each launch, 2 steps right after the timed region with the weight gradients on the "
                      "main stream (a kernel's own duration, as rocprofv3 --kernel-trace of BIN_AMD_WGRAD_STREAM=0 reports "
                      "it); `avg_kernel_us_beside_backward_data`: the same launches on the side stream, sharing the chip with "
                      "the backward-data chain as in the timed steps"}
    return out


def pmc_wgrad_traffic():
    """HBM-side bytes per launch of the 3x3 weight-gradient kernel from the committed PMC passes (profiles/), or null."""
    for name in PMC_FILES:
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                v = json.load(f).get("wgrad3x3")
            if v:
                return int(v["traffic_bytes_per_launch"]), "profiles/" + name
        except Exception:
            pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default=os.environ.get("BIN_AMD_BENCH_PRECISION", "f16x3"),
                    choices=["f16", "f16x3"],
                    help="f16x3 (default): fp32-class, the headline; f16: the tolerance mode (3.5e-4 vs the fp32 reference)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs reported beside `value` (tolerance mode, streaming, training step)")
    ap.add_argument("--streams", type=int, default=None, help="concurrent RDN calls (HIP streams) in the forward")
    ap.add_argument("--four-calls", action="store_true",
                    help="A/B: force the four-call schedule (stage s of both windows batched along N; the default only for "
                         "small frames and training) for the 720p inference window")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power", action="store_true",
                    help="skip the power passes after the timed regions (clock / watts sampling, the zero-operand control)")
    ap.add_argument("--no-harness", action="store_true", help="skip the PNG-in -> PNG-out folder leg")
    ap.add_argument("--harness-frames", type=int, default=81, help="synthetic 720p PNG frames of the folder leg")
    ap.add_argument("--calib", action="store_true",
                    help="also run one 256 MiB device copy (known HBM bytes) to calibrate rocprofv3 FETCH/WRITE_SIZE")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="train: BASELINE config 4 — one optimize_parameters() (fwd + Charbonnier + bwd + grad "
                         "all-reduce + Adam) on 256x256 crops, --batch samples per GPU (secondary metric, own JSON line)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--train-precision", default="f16x3", choices=["f16", "f16x3", "mixed"],
                    help="precision of --mode train (default f16x3: fp32-class gradients)")
    ap.add_argument("--pipeline", action="store_true",
                    help="let consecutive steps overlap (side streams wait on the resident frames, not on the previous join)")
    ap.add_argument("--zero-data", action="store_true",
                    help="diagnostic (invalid as a result): all-zero frames and weights — same instruction stream, minimal "
                         "datapath toggling; a large speed-up means the real run is power/clock-limited, not issue-limited")
    ap.add_argument("--variant", action="append", default=[],
                    help="tools/ A/B only (needs BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so): class=variant for "
                         "binhip_set_variant, or tail=depth for binhip_set_tail_depth")
    ap.add_argument("--reference-schedule", action="store_true",
                    help="run the reference's literal 20 RDN calls + 12 cells instead of the exact 17 + 6")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand with --gpus N: re-launch as one process per GPU (what the driver does itself for N > 1)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29531"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    # One process per GPU over RCCL ("nccl") is the product path.  BIN_AMD_BENCH_BACKEND=gloo + BIN_AMD_BENCH_SHARE_GPU=1 is a test
    # hook: RCCL refuses two ranks on one device, so the N > 1 flow (barriers, max-over-ranks time, whole-job value, the DP
    # gradient all-reduce of --mode train) is exercised on a 1-GPU box with every rank on cuda:(local_rank % device_count).
    backend = os.environ.get("BIN_AMD_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if os.environ.get("BIN_AMD_BENCH_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    for v in args.variant:                   # side-build switches (never present in the product library)
        from bin_amd import _lib as _L
        k, val = v.rsplit("=", 1)
        if k == "tail":
            _L.lib().binhip_set_tail_depth(int(val))
        else:
            _L.lib().binhip_set_variant(int(k), int(val))
    if args.mode == "train":
        return train_bench(args, rank, world, dev)

    from bin_amd import _lib as L
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.utils import util
    from bin_amd.weights import reference_state_dict, synthetic_frames

    net = bin_stage4_lstm()
    net.load_state_dict(reference_state_dict(0), strict=True)
    net = net.to(dev).eval().set_precision(args.precision)
    net.reuse_schedule = not args.reference_schedule
    if args.streams is not None:
        net.n_streams = args.streams
    if args.four_calls:
        net.four_calls_infer = "1"

    pads = util.pad_sizes(H, W)
    frames = [util.replicate_pad(f, pads).to(dev) for f in synthetic_frames(1234 + rank, 1, H, W, 6)]
    if args.zero_data:
        frames = [torch.zeros_like(f) for f in frames]
        with torch.no_grad():
            for prm in net.parameters():
                prm.zero_()
    hp, wp = frames[0].shape[2], frames[0].shape[3]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = net(*frames)
        if args.calib:
            src = torch.empty(64 << 20, dtype=torch.float32, device=dev).normal_()
            dst = torch.empty_like(src)
            for _ in range(3):
                dst.copy_(src)
            del src, dst
        sync_all()
        lib = L.lib()
        # --pipeline: the frames are resident and complete (sync_all above), so input_events=[] lets consecutive steps
        # overlap — the next window's stage-1 calls start on idle streams while the previous window's lone stage-4 call
        # still runs.  Measured +0.4 % (31.43 vs 31.30 frames/s): every kernel already fills both workgroup slots of
        # every CU, so another stream's kernels only slip into the ramp/drain.  Off by default.
        kw_in = {"input_events": []} if (net.resolved_streams() > 1 and not args.four_calls and args.pipeline) else {}
        from bin_amd.utils.smi import power_pass
        outs = [None]

        def one_window(i):
            outs[0] = net(*frames, **kw_in)
        dt, step_ms = timed_steps(one_window, args.steps, sync_all)
        out = outs[0]
        hip_out_path = alt_out_path = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.zero_data:
            # the timed window's outputs, kept for the cpu_baseline leg: the oracle it times there also checks them (hip_vs_oracle)
            import tempfile
            fd, hip_out_path = tempfile.mkstemp(suffix=".pt", prefix="bin_amd_bench_out_")
            os.close(fd)
            torch.save([o.detach().float().cpu() for o in out], hip_out_path)
        # shader clock / package power of the same forwards, in a pass of its own right after the timed region
        do_power = rank == 0 and not args.no_power
        power = power_pass(lambda: net(*frames, **kw_in), dev, min_seconds=1.0, sync=torch.cuda.synchronize) if do_power else None
        # ---- roofline leg: the dominant kernel's mean duration, HIP events on the launch stream.  With several
        # streams kernels of different RDN calls overlap and a per-kernel duration is not meaningful, so this pass
        # re-runs the same forward serially (n_streams = 1) right after the timed region; `value` is unaffected.
        from bin_amd import ops, rdn_plan
        ops.check_status()                    # nothing in the timed region left the fp16 storage range
        launches_per_step = (17 if net.reuse_schedule else 20) * 36
        prof_steps = min(args.steps, 5)
        prof = rank == 0 and prof_steps * launches_per_step <= 16384

        def dominant_kernel_pass():
            """(summed ms, launches) of the dense-block conv class (3x3, 32 outputs, plane epilogue) over `prof_steps`
            serial forwards of the CURRENT precision, HIP event pairs on the launch stream."""
            ms_, n_ = ctypes.c_double(0), ctypes.c_int(0)
            saved_streams = net.n_streams
            net.n_streams = 1
            net(*frames)
            torch.cuda.synchronize()
            handle = ctypes.c_void_p(0)
            L.check(lib.binhip_profiler_create(3, 32, L.EPI_PLANES, prof_steps * launches_per_step,
                                               ctypes.byref(handle)), "profiler_create")
            net.set_profiler(handle)
            for _ in range(prof_steps):
                net(*frames)
            torch.cuda.synchronize()
            net.set_profiler(None)
            L.check(lib.binhip_profiler_read(handle, ctypes.byref(ms_), ctypes.byref(n_)), "profiler_read")
            lib.binhip_profiler_destroy(handle)
            net.n_streams = saved_streams
            return ms_.value, n_.value

        kern_ms, kern_n = dominant_kernel_pass() if prof else (0.0, 0)
        extras = rank == 0 and not args.no_extras
        # ---- streaming leg (reported beside `value`, never instead of it): consecutive windows of
        # one clip, sliding by one frame, with the exact cross-window reuse -> 10 instead of 17 RDN calls per window (rounds 1-3: 13)
        stream_fps = None
        if extras and net.reuse_schedule:
            clip_frames = frames + [f.clone() for f in frames[:4]]      # 10 resident padded frames -> 5 windows
            cache = {}
            net(*clip_frames[0:6], stage1_cache=cache)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            nwin = 0
            for rep in range(2):
                for i in range(1, 5):
                    net(*clip_frames[i:i + 6], stage1_cache=cache)
                    nwin += 1
                for i in range(3, -1, -1):
                    net(*clip_frames[i:i + 6], stage1_cache=cache)      # sliding back repeats as many calls
                    nwin += 1
            torch.cuda.synchronize()
            stream_fps = nwin / (time.perf_counter() - ts)
        # ---- the same workload in the OTHER precision mode, reported beside `value` (never instead of it)
        alt = None
        other = "f16" if args.precision == "f16x3" else "f16x3"
        if extras:
            net.set_precision(other)
            for _ in range(2):
                net(*frames)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            n_alt = max(3, args.steps // 2)
            for _ in range(n_alt):
                net(*frames)
            torch.cuda.synchronize()
            t_alt = (time.perf_counter() - ta) / n_alt
            if hip_out_path:                 # the other mode's outputs of the same window, for the same live check
                alt_out_path = hip_out_path[:-3] + "_alt.pt"
                torch.save([o.detach().float().cpu() for o in net(*frames)], alt_out_path)
            alt_power = power_pass(lambda: net(*frames), dev, min_seconds=0.6, sync=torch.cuda.synchronize) if do_power else None
            alt_ms, alt_n = dominant_kernel_pass() if prof else (0.0, 0)
            alt = {"precision": other, "dtype": DTYPE[other], "value": round(1.0 / t_alt, 4),
                   "unit": "interpolated frames/s", "n_gpus": 1, "ms_per_step": round(t_alt * 1e3, 3),
                   "power": alt_power,
                   "roofline": kernel_roofline(other, alt_ms, alt_n, hp, wp, t_alt * 1e3, net.reuse_schedule),
                   "parity": "f16: max-abs <= 1e-3 / |dPSNR| <= 0.01 dB, f16x3: max-abs <= 2e-5 vs the fp32 reference "
                             "(tests/test_gpu_net.py incl. the full-size 720p fixture tests/golden/g8_720p.npz)"}
            net.set_precision(args.precision)
            ops.check_status()
        # ---- zero-operand control of the headline mode: same forwards, all-zero frames and weights
        power_bound = None
        if do_power and not args.zero_data:
            saved = [prm.detach().clone() for prm in net.parameters()]
            zframes = [torch.zeros_like(f) for f in frames]
            for prm in net.parameters():
                prm.zero_()
            for _ in range(2):
                net(*zframes, **kw_in)
            torch.cuda.synchronize()
            zero = power_pass(lambda: net(*zframes, **kw_in), dev, min_seconds=0.6, sync=torch.cuda.synchronize)
            for prm, sv in zip(net.parameters(), saved):
                prm.copy_(sv)
            del saved, zframes
            power_bound = power_bound_object(dt / args.steps * 1e3, power, zero, "720p window, " + args.precision)
            ops.check_status()
        if world > 1:
            dist.barrier()
    assert all(torch.isfinite(o).all() for o in out)
    del out

    dt = max_over_ranks(dt, dev, world)

    # ---- BASELINE config 3/4 on this GPU: one driver-timed training figure on the same line (N = 1 default run)
    train = None
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            train = train_bench(args, rank, world, dev, steps=4, warmup=2, standalone=False)
        except Exception as e:          # the headline must still print
            train = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- N1 end to end: PNG folder in -> PNG folder out through bin_amd.test, headline precision
    harness = None
    if rank == 0 and world == 1 and not args.no_extras and not args.no_harness and not args.zero_data:
        try:
            matched = resident_clip_fps(net, frames, args.harness_frames) if net.reuse_schedule else None
            harness = harness_bench(args.precision, args.harness_frames, matched, steady_gpu_only_fps=stream_fps)
        except Exception as e:          # the headline must still print
            harness = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        value = world * args.steps / dt
        prec = args.precision
        ms = dt / args.steps * 1e3
        # (kernel_roofline refuses a byte yardstick that would exceed the HBM peak: round 1's f16 line did)
        roof = kernel_roofline(prec, kern_ms, kern_n, hp, wp, ms, net.reuse_schedule) if prof else None
        if roof is not None:
            roof.update(clock_fields(ms, power))
            roof["step_ms"] = step_spread(step_ms)
        line = {
            "metric": "interpolated frames/sec at 1280x720", "value": round(value, 4),
            "unit": "interpolated frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[prec],
            "data": ("all-zero operands (diagnostic, INVALID as a result)" if args.zero_data else "synthetic"),
            "config": {"workload": "Adobe240 test_blur 1280x720 inference, batch=1 per GPU "
                                   "(6-frame window padded to 768x1344 by the test.py rule), window-sharded",
                       "schedule": ("17 RDN calls + 6 ConvLSTM cells (exact reuse)" + ("; UPNet of every call as one fused 5x5 convolution" if fused_upnet_active() else "")) if net.reuse_schedule
                                   else "20 RDN calls + 12 ConvLSTM cells (reference literal)",
                       "precision": prec, "streams": net.resolved_streams(), "pipelined_steps": bool(kw_in),
                       "four_call_schedule": bool(args.four_calls),
                       "parity": "max-abs <= 2e-5 (f16x3) vs the fp32 reference (tests/)" if prec == "f16x3"
                                 else "max-abs <= 1e-3 (f16) vs the fp32 reference (tests/)"},
            "step_ms": step_spread(step_ms),
            "roofline": roof,
            "power": power,
            "power_bound": power_bound,
            "harness": harness,
            "nccl_ranks": (dist.get_world_size() if world > 1 else 1),
            "backend": (dist.get_backend() if world > 1 else None),
            "tolerance_mode" if other == "f16" else "fp32_class": alt,
            "streaming": None if stream_fps is None else {
                "value": round(stream_fps, 3), "unit": "interpolated frames/s",
                "note": "consecutive windows of one clip (sliding by one frame) with exact cross-window reuse (the next window repeats 4 stage-1, 2 stage-2 and 1 stage-3 call of this one, none of which sees ConvLSTM state): 10 RDN calls per "
                        "window instead of 17; same outputs bit for bit (tests/test_gpu_net.py); not the headline value"},
            "train": train,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(hip_outputs=hip_out_path, alt_outputs=alt_out_path)
            for pth in (hip_out_path, alt_out_path):
                if pth and os.path.exists(pth):
                    os.remove(pth)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
