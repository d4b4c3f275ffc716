"""Per-kernel HBM-side traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).
usage: pmc_traffic.py <fetch_dir> <write_dir> [--json out.json --key f16x3]
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts exactly half of the bytes of wide streaming
reads (MI355X_MICROARCH.md §HBM) — verified in the same run on the 256 MiB device copies bench.py --calib adds — so
traffic = 2 x FETCH + WRITE."""
import csv
import glob
import json
import sys
from collections import defaultdict


BIG = {}          # counter -> the largest per-dispatch values of the device-copy kernel (the --calib 256 MiB copies)


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    big = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
            if "copyBuffer" in k:
                big.append(float(r["Counter_Value"]))
    BIG[counter] = sorted(big)[-3:]
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    key = sys.argv[sys.argv.index("--key") + 1] if "--key" in sys.argv else "f16x3"
    F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    print(f"calibration, the three 256 MiB (262144 KiB) device copies of --calib: FETCH_SIZE {BIG.get('FETCH_SIZE')} KiB, "
          f"WRITE_SIZE {BIG.get('WRITE_SIZE')} KiB")
    print("| kernel | launches | 2xFETCH MB | WRITE MB | traffic MB |\n|---|---:|---:|---:|---:|")
    rows = {}
    for k in sorted(F, key=lambda k: -F[k][0] * F[k][1]):
        if k not in W or "at::native" in k or "copyBuffer" in k:
            continue
        f2, w = 2 * F[k][0] * 1024 / 1e6, W[k][0] * 1024 / 1e6
        rows[k] = {"launches": F[k][1], "fetch2_MB": round(f2, 2), "write_MB": round(w, 2), "traffic_MB": round(f2 + w, 2)}
        print(f"| `{k[:80]}` | {F[k][1]} | {f2:.1f} | {w:.1f} | {f2 + w:.1f} |")
    if out_json:
        try:
            doc = json.load(open(out_json))
        except Exception:
            doc = {}
        want = {"f16x3": "conv_x3_kernel<3, 2, 8, 0, 0, false>", "f16": "conv_mfma_kernel<3, 1, 1, 2, 8, 1, 1, 2, 0, false>",
                "wgrad3x3": "wgrad3x3_xrow_kernel<3>"}[key]          # the dominant kernel of the profiled run
        dom = [k for k in rows if want in k]
        doc[key] = {"kernel": dom[0] if dom else None,
                    "traffic_bytes_per_launch": int(rows[dom[0]]["traffic_MB"] * 1e6) if dom else None,
                    "calibration_KiB": {"copy_bytes_KiB": 262144, "FETCH_SIZE": BIG.get("FETCH_SIZE"), "WRITE_SIZE": BIG.get("WRITE_SIZE")},
                    "kernels": rows}
        json.dump(doc, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
