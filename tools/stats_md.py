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
