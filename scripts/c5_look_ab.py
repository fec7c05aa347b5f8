import json, os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import bench
from automatedvaletparking_amd import workloads, config, costmap, _native, path_planner
cfg = config.default_config()
m, c5, starts, goals, _ = workloads.c5_problems(cfg, 1024, device="cuda")
veh = costmap.Vehicle()
for cap in (300, 1000, 3000):
    dm = _native.DeviceMap(m, veh, c5, device=0, max_pops=cap)
    stt, got = dm.dev_tensor(starts), dm.dev_tensor(goals)
    row = {"lib": os.path.basename(_native.LIB_PATH), "cap": cap}
    for name, look in (("off", False), ("auto", None)):
        bp = path_planner.BatchPlanner(dm, max_nodes=max(16384, 12 * cap), max_path=256, lookahead=look)
        bp.plan_dev(stt, got, want_paths=True); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); r = bp.plan_dev(stt, got, want_paths=True); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        row["ms_" + name] = round(min(ts), 2); row["look_" + name] = bool(bp.last_lookahead); row["mode_" + name] = int(bp.last_mode) if hasattr(bp, "last_mode") else None
        del bp
    print(json.dumps(row), flush=True)
