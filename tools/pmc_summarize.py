"""Aggregates a rocprofv3 --pmc counter_collection CSV per kernel: calls, mean and total of each counter.
usage: python tools/pmc_summarize.py <dir with *counter_collection.csv> <out.csv>"""
import csv, glob, os, sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: [0, 0.0])
for f in files:
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            k = (row.get("Kernel_Name", "?")[:120], row.get("Counter_Name", "?"))
            a = acc[k]
            a[0] += 1
            a[1] += float(row.get("Counter_Value", 0) or 0)
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Mean", "Total"])
    for (k, c), (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, tot / max(n, 1), tot])
print("wrote", out, "from", len(files), "files,", len(acc), "rows")
