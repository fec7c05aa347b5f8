"""Per-kernel sums of whatever counters a rocprofv3 --pmc output directory holds (largest-grid launches averaged).
    python scripts/pmc_counters_dump.py <dir> [kernel-name regex]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "."
per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
grid = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        d = (f, r["Dispatch_Id"])
        per[k][d][r["Counter_Name"]] += float(r["Counter_Value"])
        grid[(k, d)] = int(r.get("Grid_Size", 0) or 0)
out = {}
for k, disp in per.items():
    if not re.search(pat, k):
        continue
    gmax = max(grid[(k, d)] for d in disp)
    sel = [v for d, v in disp.items() if grid[(k, d)] == gmax]
    out[k] = {c: sum(v[c] for v in sel) / len(sel) for c in sel[0]}
    out[k]["launches"] = len(sel)
    out[k]["grid_size"] = gmax
print(json.dumps(out, indent=1))
