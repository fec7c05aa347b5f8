"""How the cost of a pop grows with the length of a search (diagnostics): bench.py's config[1] problem set at several pop
caps -- time per batch with and without the expansion lookahead, the lookahead's counters, and the per-pop phase cycles of
the capped searches from the instrumented instantiation (which runs without the lookahead).
usage: python scripts/cap_growth.py [cap ...]      (default 300 1000 2000 3000; the node arena grows with the cap -- 12 nodes per pop, at least the
bench's 16 384 -- and every entry says whether the expansion lookahead was on: its record store has a fixed size since round 6)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ["init", "pop", "res_classify", "res_write", "resolve||shot", "children||substeps", "rs_words..replay", "slow_resolve", "(sweep)", "finish",
         "res_push", "rs_words_only", "children_w0", "shot_round0", "shot_all", "pop_ahead"]


def main():
    import torch
    import bench
    from automatedvaletparking_amd import _native, path_planner, config, costmap, workloads
    caps = [int(v) for v in sys.argv[1:]] or [300, 1000, 2000, 3000]
    cfg, veh = config.default_config(), costmap.Vehicle()
    m = costmap.Map(file=os.path.join(bench.CASES, "Case1.csv"), discrete_size=cfg["map_discrete_size"])
    st = go = None
    for cap in caps:
        dm = _native.DeviceMap(m, veh, cfg, device=0, max_pops=cap)
        if st is None:
            st, go = workloads.sample_pairs(m, dm.check_batch, 256, np.random.default_rng(20260927))
        stt, got = dm.dev_tensor(st), dm.dev_tensor(go)
        max_nodes = max(bench.MAX_NODES, 12 * cap)           # (as bench.py's cap_sweep: a search makes up to 10 nodes per pop)
        out = {"cap": cap, "max_nodes": max_nodes}
        for name, look in (("without_lookahead", False), ("with_lookahead", True)):
            bp = path_planner.BatchPlanner(dm, max_nodes=max_nodes, max_path=bench.MAX_PATH, mode=1, lookahead=look)
            bp.plan_dev(stt, got)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                res, _, _ = bp.plan_dev(stt, got)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:256]
            capped = rec["status"] == 4
            e = {"ms_best": round(min(ts), 3), "lookahead": bool(bp.last_lookahead), "us_per_pop_of_the_longest_search": round(min(ts) * 1e3 / max(int(rec["n_pops"].max()), 1), 3), "pops": int(rec["n_pops"].sum()), "capped": int(capped.sum()), "out_of_node_slots": int((rec["status"] == 5).sum()),
                 "capped_mean": {k: round(float(rec[k][capped].mean()), 1) for k in ("n_nodes", "n_open", "n_closed", "h_cells", "h_misses", "n_checks", "n_rs")} if capped.any() else None}
            if bp._look is not None:
                c = bp._look[:1024].cpu().numpy().view(np.uint64)
                e.update(jobs_posted=int(c[0]), records_used=int(c[8]), record_pop_frac=round(float(c[8]) / max(int(rec["n_pops"].sum()), 1), 4),
                         next_node_lookups={"not_posted": int(c[72]), "pending": int(c[73]), "ready": int(c[74]), "waited": int(c[75])}, adopted_late=int(c[76]),
                         children_posted_by_dive_prediction=int(c[77]), children_posted_by_helpers=int(c[71]), lookahead_workspace_bytes=int(bp._look.numel()))
            out[name] = e
        bp = path_planner.BatchPlanner(dm, max_nodes=max_nodes, max_path=bench.MAX_PATH, mode=1, lookahead=False)
        resp, _, _ = bp.plan_dev(stt, got, profile=True)
        torch.cuda.synchronize()
        rp = resp.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:256]
        capped = rp["status"] == 4
        if not capped.any():                                # (every long search ran out of node slots first: status CAPACITY)
            print(json.dumps(out), flush=True)
            continue
        ph = rp["phase_cycles"].astype(np.float64)
        out["phase_cyc_per_pop"] = {n: round(float((ph[capped, k] / rp["n_pops"][capped]).mean())) for k, n in enumerate(NAMES)}
        out["cyc_per_pop"] = round(float((ph[capped][:, [1, 4, 5, 6, 7]].sum(axis=1) / rp["n_pops"][capped]).mean()))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
