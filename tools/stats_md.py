"""Turn a rocprofv3 --stats kernel_stats.csv into the markdown table committed under profiles/."""
import csv, glob, sys
src = sys.argv[1]
files = glob.glob(src + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("| kernel | calls | total ms | avg µs | min µs | max µs | % |\n|---|---:|---:|---:|---:|---:|---:|")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.2f} | "
          f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
print(f"\nTotal kernel time {tot/1e6:.1f} ms")

# Outliers: a kernel whose slowest launch is > 8x its average gets one line saying WHICH launch that was (ordinal among the kernel's
# own launches and among all dispatches, start offset from the first dispatch) — a first-launch code load reads differently from a
# stall in the middle of a step (one rdb_tail_x3_kernel launch of 16 956 us in the training trace).
traces = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)
if traces:
    per = {}
    t_first = None
    with open(traces[0]) as f:
        for i, r in enumerate(csv.DictReader(f)):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            t_first = s if t_first is None else min(t_first, s)
            per.setdefault(r["Kernel_Name"], []).append((e - s, s, i))
    lines = []
    for name, ls in per.items():
        avg = sum(d for d, _, _ in ls) / len(ls)
        order = sorted(ls, key=lambda v: v[1])
        worst = max(order, key=lambda v: v[0])
        if len(ls) >= 8 and worst[0] > 8 * avg:
            k = order.index(worst)
            rest = (sum(d for d, _, _ in ls) - worst[0]) / (len(ls) - 1)
            lines.append(f"* `{name[:80]}`: slowest launch {worst[0] / 1e3:.1f} µs is launch #{k + 1} of {len(ls)} of this kernel "
                         f"(dispatch #{worst[2] + 1} of the trace, {(worst[1] - t_first) / 1e6:.1f} ms after the first dispatch); "
                         f"average of the other {len(ls) - 1}: {rest / 1e3:.2f} µs")
    if lines:
        print("\nOutlier launches (max > 8 x average):")
        print("\n".join(lines))
