"""The 20 BenchmarkCases' own problems run to termination (cap 30 000), one problem per launch with the expansion lookahead -- the
`cases20` block of bench.py as a stand-alone A/B probe: python scripts/cases20_bench.py [--lib path/to/libavp_hip_<variant>.so]"""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
a = ap.parse_args()
if a.lib:
    os.environ["AVP_HIP_LIB"] = os.path.abspath(a.lib)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from automatedvaletparking_amd import costmap, config, _native, path_planner, workloads  # noqa: E402

cfg = config.default_config()
veh = costmap.Vehicle()
maps = workloads.case_maps(range(1, 21), cfg, device="cuda")
out = {}
for k, m in enumerate(maps, 1):
    c = m.case
    dm = _native.DeviceMap(m, veh, cfg, max_pops=30000)
    bp = path_planner.BatchPlanner(dm, max_nodes=1 << 19, max_path=256, mode=1, lookahead=True)
    st, go = dm.dev_tensor(np.array([[c.x0, c.y0, c.theta0]])), dm.dev_tensor(np.array([[c.xf, c.yf, c.thetaf]]))
    bp.plan_dev(st, go)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, _, _ = bp.plan_dev(st, go)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    r = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[0]
    out[f"Case{k}"] = (int(r["status"]), int(r["n_pops"]), round(ms, 2))
    del bp, dm
print(json.dumps({"lib": os.path.basename(_native.LIB_PATH), "total_ms": round(sum(v[2] for v in out.values()), 1), "cases": out}))
