"""Build the committed profile summaries of a round from what a `tools/gpu.sh` profile run left under gpurun_out/.

    gpurun --timeout 3000 -- bash tools/gpu.sh "tag r5p" \
        "kt x3 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-power" \
        "kt f16 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-power --precision f16" \
        "env BIN_AMD_WGRAD_STREAM=0" "kt train --mode train --batch 8 --steps 2 --warmup 1 --no-power" \
        "traffic wgrad3x3 --mode train --batch 8 --steps 1 --warmup 1 --no-power" "env BIN_AMD_WGRAD_STREAM=" \
        "traffic f16x3 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-power --calib" \
        "traffic f16 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-power --calib --precision f16" \
        "sq rdb 3 160" "sq tail 3 192" "sq wgrad 3 160" "bench" "bench --mode train"
    python tools/assemble_profiles.py --tag r5p --round 5

Writes profiles/rNN_kernel_stats_720p.md, rNN_train_kernel_stats.md, rNN_pmc_traffic.md/.json, rNN_pmc_sq.md, rNN_bench_f16x3.json,
rNN_bench_train.json.  Every number in the prose is computed here from the run's own files; nothing is typed in by hand.
"""
import argparse
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
DOM_X3 = "conv_x3_kernel<3, 2, 8, 0, 0, false>"
TAIL_X3 = "rdb_tail_x3_kernel"
WG3 = "wgrad3x3_xrow_kernel<3>"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r5p")
    ap.add_argument("--round", type=int, default=5)
    a = ap.parse_args()
    T, R = a.tag, f"r{a.round:02d}"

    def read(name):
        with open(os.path.join(G, f"{T}_{name}")) as f:
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

    def mean(d, k):
        v = (d or {}).get(k)
        return v.get("mean") if isinstance(v, dict) else None

    def xcd(d):
        return ((d or {}).get("xcd_clock_mhz") or {}).get("mean")

    def xcd_z(pb_):
        return ((((pb_ or {}).get("other_domains") or {}).get("xcd_clock_mhz") or {}).get("zero") or {}).get("mean")

    sx, sf, st = read("stats_x3.md"), read("stats_f16.md"), read("stats_train.md")
    bp = jline("kt_x3.log")                      # the bench line printed INSIDE the profiled run
    b, bt = jline("bench_1.json"), jline("bench_2.json")
    pm = json.loads(read("pmc_traffic.json"))
    n, _, avg = row(sx, DOM_X3)
    nt_, _, avgt = row(sx, TAIL_X3)
    roof = b["roofline"]
    ab = roof["algorithmic_bytes_per_launch"]
    tr = pm["f16x3"]["traffic_bytes_per_launch"]
    gflop = roof["algorithmic_flop_per_launch"] * 3 / 1e9       # executed MFMA GFLOP of one launch (3 products)
    pb = b.get("power_bound") or {}
    with open(os.path.join(P, f"{R}_kernel_stats_720p.md"), "w") as f:
        f.write(f"""# Round {a.round} — rocprofv3 --kernel-trace --stats of the default bench command (MI355X, 720p window, f16x3 headline)

Command on the GPU box (`tools/gpu.sh "kt x3 ..."`, see the header of `tools/assemble_profiles.py`, which wrote this file):
`export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/x3 -o kt -- python bench.py --steps 3
--warmup 1 --no-cpu-baseline --no-extras --no-power` (default precision f16x3, default schedule = 1 stream, 17 RDN calls per window).  The run
holds 8 forwards (1 warm-up + 3 timed + 1 + 3 of the serial roofline leg); per forward 612 dense-block conv launches (`{DOM_X3}`), 204
fused tails, 34 wide 3x3 layers, 17 each of GFF.0, SFENet1 and the fused UPNet (`conv_x3_kernel<5, 2, 8, 4, 0, false>` + `upnet_ring_kernel`: round 6, one 5x5 convolution on 12 sub-pixel channels instead of UPNet.0 + UPNet.2).  bench.py's line in the same (profiled) run: {bp['value']} frames/s,
{bp['ms_per_step']} ms / window, live HIP-event average of the dominant kernel {bp['roofline']['avg_kernel_us']} us.  Un-profiled on the same box
(`{R}_bench_f16x3.json`): **{b['value']} frames/s, {b['ms_per_step']} ms / window**, dominant kernel {roof['avg_kernel_us']} us by events,
{xcd(b.get('power'))} MHz (per-XCD mean; amdsmi's GFX clk = XCD 0 alone: {mean(b.get('power'), 'clock_mhz')}) at {(b.get('power') or {}).get('power_from_energy_w')} W by the energy
counter ({mean(b.get('power'), 'power_w')} W point-sampled; a pass of its own right after the timed region); the same forwards on ALL-ZERO operands: {pb.get('ms_zero')} ms
at {xcd_z(pb)} MHz -> time ratio {pb.get('ratio')} against a clock ratio of {pb.get('clock_ratio_used')}, cycle ratio {pb.get('cycle_ratio')}.
`power_bound.reading`: "{pb.get('reading')}"

Dominant kernel: **{avg:.2f} us** average over {n} launches -> {gflop / 3:.1f} GFLOP x 3 products / {avg:.2f} us = {gflop / avg:.3f} PFLOP/s =
**{gflop / avg / 2.5:.3f} of the 2.5 PFLOP/s dense fp16 peak** (`roofline.bound` = "{roof['bound']}": {roof['arithmetic_intensity_flop_per_byte']} FLOP/B on a ridge of
{roof['ridge_flop_per_byte']} for three products); by bytes, {ab / 1e6:.1f} MB algorithmic (4 B per element) / {avg:.2f} us = {ab / avg / 1e6:.2f} TB/s =
{ab / avg / 1e6 / 8:.3f} of the 8 TB/s HBM peak; PMC traffic {tr / 1e6:.1f} MB per launch (`{R}_pmc_traffic.md`) = {tr / ab:.2f} x algorithmic.  The fused
dense-block tail (`{TAIL_X3}`): {avgt:.2f} us over {nt_} launches.  SFENet1 (`conv_x3_kernel<5, 2, 8, 0, 1, false>`; round 5: the half-empty last
chunk of the 24- / 36-channel calls on tap pairs): {row(sx, 'conv_x3_kernel<5, 2, 8, 0, 1, false>')[2]:.1f} us (round 4: 155.2).

{sx.strip()}

## f16 (tolerance mode: `--precision f16`, 3 streams in the timed region, serial in the roofline leg)

UPNet runs fused here too (`conv_mfma_kernel<5, 1, 1, 2, 8, 1, 1, 2, 4, false>` + `upnet_ring_kernel`).

{sf.strip()}
""")
    nw, _, avgw = row(st, WG3)
    dk = bt["roofline"]["dominant_kernel"]
    abw, trw = dk["algorithmic_bytes_per_launch"], pm["wgrad3x3"]["traffic_bytes_per_launch"]
    tpb = bt.get("power_bound") or {}
    with open(os.path.join(P, f"{R}_train_kernel_stats.md"), "w") as f:
        f.write(f"""# Round {a.round} — training step (BASELINE config 3/4: 8 x 256x256 crops per GPU, f16x3) kernel stats, MI355X

`BIN_AMD_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --mode train --batch 8 --steps 2 --warmup 1 --no-power`
(the side stream is off so that kernel durations do not include each other; written by `tools/assemble_profiles.py`).  Un-profiled on the
same box (`{R}_bench_train.json`): **{bt['ms_per_step']} ms / step = {bt['value']} samples/s**, {(bt.get('power') or {}).get('power_from_energy_w')} W (energy counter) at
{xcd(bt.get('power'))} MHz (per-XCD mean); the same steps on all-zero operands: {tpb.get('ms_zero')} ms at {xcd_z(tpb)} MHz (time ratio {tpb.get('ratio')},
clock ratio {tpb.get('clock_ratio_used')}, cycle ratio {tpb.get('cycle_ratio')}).  `power_bound.reading`: "{tpb.get('reading')}"

The 3x3 weight gradient (`{WG3}`): {avgw:.1f} us average over the {nw} launches of this trace (all batch sizes of the four-call
schedule); bench.py's own event-timed average of the 192 dense-block launches per step, side stream off: {dk['avg_kernel_us']} us
(`train.roofline.dominant_kernel`; {dk['avg_kernel_us_beside_backward_data']} us when sharing the chip with the backward-data chain as in the timed steps)
— {abw / 1e6:.1f} MB algorithmic / {dk['avg_kernel_us']} us = {dk['hbm']['achieved'] / 1e3:.2f} TB/s = {dk['hbm']['frac']:.2f} of the HBM peak, {dk['mfma']['achieved'] / 1e3:.2f} PFLOP/s executed =
{dk['mfma']['frac']:.2f} of the MFMA peak; PMC traffic {trw / 1e6:.1f} MB per launch = {trw / abw:.2f} x algorithmic (`{R}_pmc_traffic.md`).

{st.strip()}
""")
    with open(os.path.join(P, f"{R}_pmc_traffic.md"), "w") as f:
        f.write(f"""# Round {a.round} — HBM-side traffic per kernel launch (rocprofv3 PMC passes, MI355X)

`tools/gpu.sh "traffic KEY ..."`: separate `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` passes (no other trace
domains) over `python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-power --calib [--precision f16]` and over
`BIN_AMD_WGRAD_STREAM=0 python bench.py --mode train --batch 8 --steps 1 --warmup 1 --no-power`; traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB units;
FETCH_SIZE counts half of the bytes of wide streaming reads on gfx950 — checked in the same run on the three 256 MiB device copies `--calib`
adds).  `bench.py` reads `{R}_pmc_traffic.json` for `roofline.traffic` (the entry of the precision it timed; `roofline.traffic_source` names the
file) and for `train.roofline.dominant_kernel.traffic` (`wgrad3x3`).  Written by `tools/assemble_profiles.py`.

## f16x3 (headline mode)
""" + read("pmc_traffic_f16x3.md").strip() + "\n\n## f16 (tolerance mode)\n" + read("pmc_traffic_f16.md").strip()
                + "\n\n## training step, f16x3 (no calibration copies in this run: the first line's small copies are torch's own)\n"
                + read("pmc_traffic_wgrad3x3.md").strip() + "\n")
    shutil.copy(os.path.join(G, f"{T}_pmc_traffic.json"), os.path.join(P, f"{R}_pmc_traffic.json"))
    for src, dst in (("bench_1.json", f"{R}_bench_f16x3.json"), ("bench_2.json", f"{R}_bench_train.json")):
        with open(os.path.join(P, dst), "w") as f:
            f.write(json.dumps(jline(src)) + "\n")

    # ---- SQ counters: the raw table + the derived launch-level figures
    sq = read("pmc_sq.log")
    # (tools/gpu.sh appends: keep the LAST section of each kernel when a tag was used for more than one run)
    secs = {}
    for sec in re.split(r"(?m)^(?==== pmc_one.py )", sq):
        m = re.match(r"=== pmc_one.py (\w+)", sec)
        if m and len(sec.splitlines()) > 2:
            secs[m.group(1)] = sec.strip()
    sq = "\n".join(secs.values())
    vals = {}
    cur = None
    for ln in sq.splitlines():
        m = re.match(r"=== pmc_one.py (\w+)", ln)
        if m:
            cur = m.group(1)
            vals[cur] = {}
            continue
        m = re.match(r".*\s(SQ_\w+|GRBM_\w+)\s+([\d.]+)\s+\(n=\d+\)", ln)
        if m and cur:
            vals[cur][m.group(1)] = float(m.group(2))

    def derived(v):
        launch = v["GRBM_GUI_ACTIVE"] / 8.0                      # cycles (8 XCDs count in parallel)
        busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0            # per SIMD (256 CUs x 4)
        life = v["SQ_WAVE_CYCLES"] * 4.0 / v["SQ_WAVES"]         # cycles a wave exists
        return {"launch_kcyc": launch / 1e3, "mfma_busy_kcyc": busy / 1e3, "mfma_share": busy / launch,
                "wave_life_share": life / launch, "parked_Mquad": v["SQ_WAIT_ANY"] / 1e6,
                "lds_busy": v["SQ_LDS_IDX_ACTIVE"] / 256.0 / launch,
                "bank_conflict": v["SQ_LDS_BANK_CONFLICT"], "lds_insts": v["SQ_INSTS_LDS"], "valu_insts": v["SQ_INSTS_VALU"]}
    names = {"rdb": "dense-block conv (`conv_x3_kernel<3,2,8,0,0,false>`, 160 -> 32)", "tail": "fused tail (`rdb_tail_x3_kernel`)",
             "wgrad": "3x3 weight gradient (`wgrad3x3_xrow_kernel<3>`, 160 -> 32)"}
    lines = ["| | " + " | ".join(names[k] for k in vals) + " |", "|---|" + "---:|" * len(vals)]
    d = {k: derived(v) for k, v in vals.items()}
    for label, key, fmt in (("launch length (GRBM_GUI_ACTIVE / 8 XCDs), k cycles", "launch_kcyc", "{:.1f}"),
                            ("MFMA pipe busy per SIMD (SQ_VALU_MFMA_BUSY_CYCLES / 1024), k cycles", "mfma_busy_kcyc", "{:.1f}"),
                            ("= share of the launch", "mfma_share", "{:.1%}"),
                            ("a wave's life (SQ_WAVE_CYCLES x 4 / SQ_WAVES), share of the launch", "wave_life_share", "{:.1%}"),
                            ("parked quad-cycles per launch (SQ_WAIT_ANY), M", "parked_Mquad", "{:.1f}"),
                            ("LDS array busy (SQ_LDS_IDX_ACTIVE / 256 CUs / launch)", "lds_busy", "{:.1%}"),
                            ("LDS instructions per launch", "lds_insts", "{:.0f}"),
                            ("VALU instructions per launch (MFMA included)", "valu_insts", "{:.0f}"),
                            ("LDS bank conflicts per launch (SQ_LDS_BANK_CONFLICT)", "bank_conflict", "{:.0f}")):
        lines.append(f"| {label} | " + " | ".join(fmt.format(d[k][key]) for k in vals) + " |")
    with open(os.path.join(P, f"{R}_pmc_sq.md"), "w") as f:
        f.write(f"""# Round {a.round} — SQ / GRBM counters of the dominant fp32-class kernels (MI355X)

Two `rocprofv3 --pmc ... --kernel-trace` passes per kernel (8 SQ slots each, no other trace domains; `tools/gpu.sh "sq rdb 3 160"` etc.) over
`python tools/pmc_one.py rdb 3 160` (dense-block conv 3x3, Cin = 160 -> 32 + ReLU), `pmc_one.py tail 3 192` (fused conv #3 + LFF + residual)
— both at the 720p working size 384x672 — and `pmc_one.py wgrad 3 160` (3x3 weight gradient 160 -> 32 on 40 x 128 x 128), 10 launches each.
Values are per launch, summed over the chip (`tools/pmc_sum.py`).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles,
SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE cycles (MI355X_MICROARCH.md).  Earlier rounds: `r03_pmc_sq.md`, `r02_pmc_sq.md`.  The table
below is computed from the raw block by `tools/assemble_profiles.py`.

```
{sq.strip()}
```

## Derived

{chr(10).join(lines)}
""")
    print("dominant", avg, "us  mfma frac", round(gflop / avg / 2.5, 3), " window", b["ms_per_step"], "ms  train", bt["ms_per_step"],
          "ms  wgrad", dk["avg_kernel_us"], " tail", avgt)


if __name__ == "__main__":
    main()
