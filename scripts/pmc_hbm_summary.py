"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately,
as /opt/skills/guides/MI355X_MICROARCH.md prescribes):
    python scripts/pmc_hbm_summary.py <dir_fetch> <dir_write> > profiles/rNN_pmc_summary.json
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, hence
hbm_bytes_per_launch_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (calibrated on check_distance_kernel, whose
traffic is known: poses in + flags out)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def collect(root, counter):
    per = defaultdict(lambda: defaultdict(float))          # kernel -> dispatch -> value (summed over XCDs / instances)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            per[k][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    return per


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    fv, wv = list(fetch.get(k, {}).values()), list(write.get(k, {}).values())
    e = {}
    if fv:
        e.update(FETCH_SIZE_KB_per_launch_max=max(fv), FETCH_SIZE_KB_per_launch_mean=sum(fv) / len(fv), launches_FETCH_SIZE=len(fv))
    if wv:
        e.update(WRITE_SIZE_KB_per_launch_max=max(wv), WRITE_SIZE_KB_per_launch_mean=sum(wv) / len(wv), launches_WRITE_SIZE=len(wv))
    if fv and wv:
        e["hbm_bytes_per_launch_corrected"] = (2 * max(fv) + max(wv)) * 1024
    out[k] = e
pk = [k for k in out if k.startswith('plan_kernel<')]
if pk:
    out['plan_kernel'] = out[max(pk, key=lambda k: out[k].get('hbm_bytes_per_launch_corrected', 0))]
json.dump(out, sys.stdout, indent=1)
