"""Build the committed round-3 profile summaries from what `tools/gpu_profile_r3.sh` left under gpurun_out/.

    python tools/assemble_profiles_r3.py      # after: gpurun -- bash tools/gpu_profile_r3.sh

Writes profiles/r03_kernel_stats_720p.md, r03_train_kernel_stats.md, r03_pmc_traffic.md/.json, r03_bench_f16x3.json and
r03_bench_train.json.  Every number in the prose is computed here from the run's own files; nothing is typed in by hand.
"""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
DOM_X3 = "conv_x3_kernel<3, 2, 8, 0, 0, false>"
DOM_F16 = "conv_mfma_kernel<3, 1, 1, 2, 8, 1, 1, 2, 0, false>"
WG3 = "wgrad3x3_xrow_kernel<3>"


def read(name):
    with open(os.path.join(G, name)) as f:
        return f.read()


def jline(name):
    """The last bench.py JSON line in a log (rocprofv3 appends its own lines after it)."""
    return json.loads([ln for ln in read(name).splitlines() if ln.startswith('{"metric"')][-1])


def row(md, key):
    """(calls, total ms, avg us) of the first table row whose kernel name contains `key`."""
    for ln in md.splitlines():
        if key in ln and ln.startswith("|"):
            c = [x.strip() for x in ln.strip("|").split("|")]
            return int(c[1]), float(c[2]), float(c[3])
    raise KeyError(key)


def total(md):
    m = re.search(r"Total kernel time ([\d.]+) ms", md)
    return float(m.group(1)) if m else None


def main():
    sx, sf, st = read("r3v_stats_x3.md"), read("r3v_stats_f16.md"), read("r3v_stats_train.md")
    bp = jline("r3v_kt_x3.log")                      # the bench line printed INSIDE the profiled run
    b, bt = jline("r3v_bench.json"), jline("r3v_bench_train.json")
    pm = json.loads(read("r3v_pmc_traffic.json"))
    n, _, avg = row(sx, DOM_X3)
    ab = b["roofline"]["algorithmic_bytes_per_launch"]
    tr = pm["f16x3"]["traffic_bytes_per_launch"]
    gflop = b["roofline"]["mfma"]["achieved"] * b["roofline"]["avg_kernel_us"] / 1e3   # executed MFMA GFLOP of one launch (3 products)
    pw = bp.get("power", {})
    ck, wt = (pw.get("clock_mhz") or {}).get("mean"), (pw.get("power_w") or {}).get("mean")
    with open(os.path.join(P, "r03_kernel_stats_720p.md"), "w") as f:
        f.write(f"""# Round 3 — rocprofv3 --kernel-trace --stats of the default bench command (MI355X, 720p window, f16x3 headline)

Command on the GPU box (`tools/gpu_profile_r3.sh`, final round-3 build; this file is written by `tools/assemble_profiles_r3.py`):
`export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3/kt_x3 -o kt -- python bench.py --steps 3
--warmup 1 --no-cpu-baseline --no-extras` (default precision f16x3, default schedule = 1 stream, 17 RDN calls per window).  The run holds 8
forwards (1 warm-up + 3 timed + 1 + 3 of the serial roofline leg); per forward 612 dense-block conv launches (`{DOM_X3}`), 204
fused tails, 34 wide 3x3 layers, 17 each of UPNet.0, UPNet.2, GFF.0, SFENet1.  bench.py's line in the same (profiled) run: {bp['value']} frames/s,
{bp['ms_per_step']} ms / window, live HIP-event average of the dominant kernel {bp['roofline']['avg_kernel_us']} us, shader clock {ck} MHz at {wt} W
(sampled during the timed region).  Un-profiled on the same box (`r03_bench_f16x3.json`): **{b['value']} frames/s, {b['ms_per_step']} ms / window**,
dominant kernel {b['roofline']['avg_kernel_us']} us by events, {b['power']['clock_mhz']['mean']} MHz at {b['power']['power_w']['mean']} W.

Dominant kernel: **{avg:.2f} us** average over {n} launches -> {ab / 1e6:.1f} MB algorithmic (4 B per element) / {avg:.2f} us = {ab / avg / 1e6:.2f} TB/s =
**{ab / avg / 1e6 / 8:.3f} of the 8 TB/s HBM peak**, {gflop / 3:.1f} GFLOP x 3 products / {avg:.2f} us = {gflop / avg:.2f} PFLOP/s = {gflop / avg / 2.5:.2f} of the 2.5 PFLOP/s
dense fp16 peak; PMC traffic {tr / 1e6:.1f} MB per launch (`r03_pmc_traffic.md`) = {tr / ab:.2f} x algorithmic.  Kernel names carry the round-3 template
arguments: the trailing `true/false` is the XTRA epilogue (residual / second skip / ReLU mask operands, grouped loads before the stores), the
`float const*` parameter the bias as a scalar-loaded pointer (`r03_experiments.md` §5).

{sx.strip()}

## f16 (tolerance mode: `--precision f16`, 3 streams in the timed region, serial in the roofline leg)

UPNet.2 runs as `final_dot2_kernel<1>` here.

{sf.strip()}
""")
    nw, _, avgw = row(st, WG3)
    dk = bt["roofline"]["dominant_kernel"]
    abw, trw = dk["algorithmic_bytes_per_launch"], pm["wgrad3x3"]["traffic_bytes_per_launch"]
    with open(os.path.join(P, "r03_train_kernel_stats.md"), "w") as f:
        f.write(f"""# Round 3 — training step (BASELINE config 3/4: 8 x 256x256 crops per GPU, f16x3) kernel stats, MI355X

`BIN_AMD_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --mode train --batch 8 --steps 2 --warmup 1`
(7 steps in the trace: 1 warm-up + 2 timed + 2 + 2 of the live-timing passes; the side stream is off so that kernel durations do not include
each other; written by `tools/assemble_profiles_r3.py`).  Un-profiled on the same box (`r03_bench_train.json`): **{bt['ms_per_step']} ms / step =
{bt['value']} samples/s**, {bt['power']['power_w']['mean']} W at {bt['power']['clock_mhz']['mean']} MHz.  Per step: ~1 100 launches; the weight relayouts are 24
`relayout_batch_kernel` launches (round 2: 528 launches of three kernels, 2.7 ms).

The 3x3 weight gradient (`{WG3}`): {avgw:.1f} us average over the {nw} launches of this trace (all batch sizes of the four-call
schedule; round 2: 160.6); bench.py's own event-timed average of the 192 dense-block launches per step, side stream off: {dk['avg_kernel_us']} us
(`train.roofline.dominant_kernel`; {dk['avg_kernel_us_beside_backward_data']} us when sharing the chip with the backward-data chain as in the timed steps)
— {abw / 1e6:.1f} MB algorithmic / {dk['avg_kernel_us']} us = {dk['achieved'] / 1e3:.2f} TB/s = {dk['frac']:.2f} of the HBM peak, {dk['mfma']['achieved'] / 1e3:.2f} PFLOP/s executed = {dk['mfma']['frac']:.2f} of the MFMA
peak; PMC traffic {trw / 1e6:.1f} MB per launch = {trw / abw:.2f} x algorithmic (`r03_pmc_traffic.md`).  The rolling-row rewrite with two rows per stage and
a column-major tile walk (-15 % traffic) were built and measured this round and are not faster (`r03_experiments.md` §1, §6).

{st.strip()}
""")
    with open(os.path.join(P, "r03_pmc_traffic.md"), "w") as f:
        f.write("""# Round 3 — HBM-side traffic per kernel launch (rocprofv3 PMC passes, MI355X)

`tools/gpu_profile_r3.sh`: separate `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` passes (no other trace
domains) over `python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --calib [--precision f16]` and over `BIN_AMD_WGRAD_STREAM=0
python bench.py --mode train --batch 8 --steps 1 --warmup 1`; traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB units; FETCH_SIZE counts half of
the bytes of wide streaming reads on gfx950 — checked in the same run on the three 256 MiB device copies `--calib` adds:
131 083 KiB fetched / 262 144 KiB written per copy).  `bench.py` reads `r03_pmc_traffic.json` for `roofline.traffic` (the entry of the
precision it timed) and for `train.roofline.dominant_kernel.traffic` (`wgrad3x3`).  Written by `tools/assemble_profiles_r3.py`.

## f16x3 (headline mode)
""" + read("r3v_pmc_traffic_f16x3.md").strip() + "\n\n## f16 (tolerance mode)\n" + read("r3v_pmc_traffic_f16.md").strip()
                + "\n\n## training step, f16x3 (no calibration copies in this run: the first line's small copies are torch's own)\n"
                + read("r3v_pmc_traffic_train.md").strip() + "\n")
    shutil.copy(os.path.join(G, "r3v_pmc_traffic.json"), os.path.join(P, "r03_pmc_traffic.json"))
    for src, dst in (("r3v_bench.json", "r03_bench_f16x3.json"), ("r3v_bench_train.json", "r03_bench_train.json")):
        with open(os.path.join(P, dst), "w") as f:
            f.write(json.dumps(jline(src)) + "\n")
    print("dominant", avg, "us  frac", round(ab / avg / 1e6 / 8, 3), " window", b["ms_per_step"], "ms  train", bt["ms_per_step"], "ms  wgrad", dk["avg_kernel_us"])


if __name__ == "__main__":
    main()
