"""The second stage of the multi-GPU two-stage deal as one GPU sees it: k searches of the 4 096 set that all run to the pop cap, workgroup form, with and
without the lookahead (k = 64 ... 512: 4 096 / N deferred searches per rank are ~100 at N = 8, ~200 at N = 4, ~400 at N = 2). One JSON line per k."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from automatedvaletparking_amd import _native, path_planner, config, costmap, workloads
    ks = [int(a) for a in sys.argv[1:]] or [64, 100, 128, 200, 256, 400, 512]
    cap = 1000
    cfg, veh = config.default_config(), costmap.Vehicle()
    m = costmap.Map(file=os.path.join(bench.CASES, "Case1.csv"), discrete_size=cfg["map_discrete_size"])
    dm = _native.DeviceMap(m, veh, cfg, device=0, max_pops=cap)
    st, go = workloads.sample_pairs(m, dm.check_batch, 4096, np.random.default_rng(20260927))
    bp = path_planner.BatchPlanner(dm, max_nodes=bench.MAX_NODES, max_path=bench.MAX_PATH, mode=4)
    res, _, _ = bp.plan_dev(dm.dev_tensor(st), dm.dev_tensor(go), want_paths=True)
    rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)
    long_ = np.nonzero(rec["n_pops"] >= cap)[0]
    del bp
    for k in ks:
        idx = long_[:k]
        stt, got = dm.dev_tensor(st[idx]), dm.dev_tensor(go[idx])
        row = {"capped_searches": int(len(idx)), "lib": os.path.basename(os.environ.get("AVP_HIP_LIB", "libavp_hip.so"))}
        for name, look in (("off", False), ("on", True)):
            bp = path_planner.BatchPlanner(dm, max_nodes=bench.MAX_NODES, max_path=bench.MAX_PATH, mode=1, lookahead=look)
            bp.plan_dev(stt, got, want_paths=True)
            torch.cuda.synchronize()
            ts = []
            for _ in range(4):
                t0 = time.perf_counter()
                r, _, _ = bp.plan_dev(stt, got, want_paths=True)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            row["ms_" + name] = round(min(ts), 2)
            if look and bp._look is not None:
                c = bp._look[:1024].cpu().numpy().view(np.uint64)
                pops = int(r.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)["n_pops"].sum())
                row["record_pop_frac"] = round(float(c[8]) / max(pops, 1), 3)
                row["jobs"] = int(c[0])
            del bp
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
