"""A/B of the problem order (avp_plan_batch_ex `order`): index order vs decreasing start-goal distance, on the 4 096-problem
batch (workgroup form) and the 16 384-problem batch (both forms). Prints one JSON object."""
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
    from automatedvaletparking_amd import _native, path_planner, config, costmap
    cfg, veh = config.default_config(), costmap.Vehicle()
    m = costmap.Map(file=os.path.join(bench.CASES, "Case1.csv"), discrete_size=cfg["map_discrete_size"])
    dm = _native.DeviceMap(m, veh, cfg, device=0, max_pops=1000)
    st4, go4 = bench.sample_pairs(m, dm, 4096, np.random.default_rng(20260927))
    st16 = np.concatenate([st4] * 4)
    go16 = np.concatenate([np.roll(go4, 17 * k, axis=0) for k in range(4)])
    out = {"source_hash": bench.source_hash()}
    for label, st, go, modes in (("batch4096", st4, go4, (1,)), ("batch16384", st16, go16, (1, 2))):
        stt, got = dm.dev_tensor(st), dm.dev_tensor(go)
        for mode in modes:
            ref = None
            for lf in (False, True):
                bp = path_planner.BatchPlanner(dm, max_nodes=bench.MAX_NODES, max_path=bench.MAX_PATH, mode=mode, lookahead=False, longest_first=lf)
                bp.plan_dev(stt, got, want_paths=True)
                torch.cuda.synchronize()
                ts = []
                for _ in range(2):
                    t0 = time.perf_counter()
                    res, paths, _ = bp.plan_dev(stt, got, want_paths=True)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)
                key = "%s_mode%d_%s" % (label, mode, "longest_first" if lf else "index_order")
                out[key] = {"ms": min(ts), "pops": int(rec["n_pops"].sum())}
                if ref is None:
                    ref = rec
                else:
                    out[key]["identical_results"] = bool(all(np.array_equal(ref[f], rec[f]) for f in rec.dtype.names if f not in ("slot", "phase_cycles")))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
