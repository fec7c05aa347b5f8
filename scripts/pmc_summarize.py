"""Sum rocprofv3 --pmc counter CSVs per kernel: python scripts/pmc_summarize.py <dir> [substring]
Prints, per kernel name (truncated), the number of dispatches and the per-dispatch mean of every counter."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if want and want not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add((f, r["Dispatch_Id"]))
for k in acc:
    n = len(disp[k])
    print(f"{k}  dispatches={n}")
    for c, v in sorted(acc[k].items()):
        print(f"    {c:28s} {v / n:.6g}")
