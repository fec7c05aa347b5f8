"""The expansion lookahead against the batch size: Case1 map, n random start/goal pairs (config[1]'s sampler), pop cap 1000, workgroup
form, with and without the lookahead; identical results asserted. One JSON line per n.
usage: python scripts/look_scale.py [n ...]      (the library decides whether the lookahead runs: `lookahead_used`)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import bench
    from automatedvaletparking_amd import _native, path_planner, config, costmap, workloads
    ns = [int(a) for a in sys.argv[1:]] or [256, 512, 768, 1024, 1536, 2048, 3072]
    cap = 1000
    cfg, veh = config.default_config(), costmap.Vehicle()
    m = costmap.Map(file=os.path.join(bench.CASES, "Case1.csv"), discrete_size=cfg["map_discrete_size"])
    dm = _native.DeviceMap(m, veh, cfg, device=0, max_pops=cap)
    for n in ns:
        st, go = workloads.sample_pairs(m, dm.check_batch, n, np.random.default_rng(20260927))
        stt, got = dm.dev_tensor(st), dm.dev_tensor(go)
        row, keep = {"n": n, "lib": os.path.basename(os.environ.get("AVP_HIP_LIB", "libavp_hip.so"))}, {}
        for name, look in (("off", False), ("on", True)):
            bp = path_planner.BatchPlanner(dm, max_nodes=bench.MAX_NODES, max_path=bench.MAX_PATH, mode=1, lookahead=look)
            bp.plan_dev(stt, got, want_paths=True)
            torch.cuda.synchronize()
            ts = []
            for _ in range(4):
                t0 = time.perf_counter()
                res, paths, _ = bp.plan_dev(stt, got, want_paths=True)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)
            keep[name] = (rec, paths.cpu().numpy())
            row["ms_" + name] = round(min(ts), 2)
            if look:
                row["lookahead_used"] = bool(bp.last_lookahead)
                if bp._look is not None:
                    c = bp._look[:1024].cpu().numpy().view(np.uint64)
                    row["record_pop_frac"] = round(float(c[8]) / max(int(rec["n_pops"].sum()), 1), 3)
                    row["jobs"] = int(c[0])
            del bp
        if os.environ.get("LOOK_SCALE_QUAD"):
            bp = path_planner.BatchPlanner(dm, max_nodes=bench.MAX_NODES, max_path=bench.MAX_PATH, mode=4)
            bp.plan_dev(stt, got, want_paths=True)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                bp.plan_dev(stt, got, want_paths=True)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            row["ms_quad"] = round(min(ts), 2)
            del bp
        a, b = keep["off"], keep["on"]
        same = all(np.array_equal(a[0][f], b[0][f]) for f in a[0].dtype.names if f not in ("slot", "phase_cycles"))
        same = same and all(np.array_equal(a[1][i, :a[0]["n_final"][i]], b[1][i, :a[0]["n_final"][i]]) for i in range(len(a[0])))
        row["identical"] = bool(same)
        row["pops"] = int(a[0]["n_pops"].sum())
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
