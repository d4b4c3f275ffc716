"""Summarise rocprofv3 --pmc CSV output: mean counter value per launch, per kernel (name prefix filter optional).
usage: pmc_sum.py <dir> [name-substring]"""
import csv
import glob
import sys
from collections import defaultdict

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if flt and flt not in name:
            continue
        k = (name[:70], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
for (name, cnt), (tot, n) in sorted(acc.items()):
    print(f"{name:70s} {cnt:32s} {tot / n:16.1f}  (n={n})")
