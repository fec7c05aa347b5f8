"""A/B of the expansion lookahead (avp_plan_batch_ex) on the bench's config[1] workload: identical results, time with and
without, the helpers' job counters, and the per-wave timeline of the pops that were served from a record (instrumented
instantiation). Prints one JSON object (committed as profiles/r06_lookahead.json).
usage: python scripts/look_bench.py [n_problems] [cap]"""
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
    from automatedvaletparking_amd import _native, path_planner, config, costmap
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    cfg, veh = config.default_config(), costmap.Vehicle()
    m = costmap.Map(file=os.path.join(bench.CASES, "Case1.csv"), discrete_size=cfg["map_discrete_size"])
    dm = _native.DeviceMap(m, veh, cfg, device=0, max_pops=cap)
    from automatedvaletparking_amd import workloads
    st, go = workloads.sample_pairs(m, dm.check_batch, n, np.random.default_rng(20260927))
    stt, got = dm.dev_tensor(st), dm.dev_tensor(go)
    out = {"workload": "Case1 map, %d random start/goal pairs, pop cap %d (bench.py's config[1] problem set)" % (n, cap),
           "source_hash": bench.source_hash()}
    keep = {}
    for name, look in (("without_lookahead", False), ("with_lookahead", True)):
        bp = path_planner.BatchPlanner(dm, max_nodes=bench.MAX_NODES, max_path=bench.MAX_PATH, mode=1, lookahead=look)
        bp.plan_dev(stt, got, want_paths=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            res, paths, _ = bp.plan_dev(stt, got, want_paths=True)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)
        keep[name] = (rec, paths.cpu().numpy())
        e = {"ms_per_batch": [round(t, 3) for t in ts], "ms_best": min(ts), "pops": int(rec["n_pops"].sum()), "lookahead_used": bool(bp.last_lookahead)}
        if bp._look is not None:
            c = bp._look[:1024].cpu().numpy().view(np.uint64)
            e.update(jobs_posted=int(c[0]), children_halves_made=int(c[24]), shot_halves_made=int(c[88]), records_used=int(c[8]),
                     helper_workgroups=int(c[48]), record_pop_frac=float(c[8]) / max(int(rec["n_pops"].sum()), 1),
                     next_node_lookups={"not_posted": int(c[72]), "pending": int(c[73]), "ready": int(c[74]), "waited": int(c[75])}, records_adopted_late=int(c[76]),
                     misses_by_kind={"not_posted": {"dive_below_record_pop": int(c[96]), "dive_below_long_pop": int(c[97]), "other": int(c[98])}, "pending": {"dive_below_record_pop": int(c[99]), "dive_below_long_pop": int(c[100]), "other": int(c[101])}},
                     children_posted_by_dive_prediction=int(c[77]), children_posted_by_helpers=int(c[71]), claims_refused_entry_busy=int(c[78]), copies_refused_by_seqlock=int(c[79]),
                     records_never_used_frac=1.0 - float(c[8]) / max(int(c[0]), 1),
                     lookahead_workspace_bytes=int(bp._look.numel()))
        out[name] = e
    a, b = keep["without_lookahead"], keep["with_lookahead"]
    same = all(np.array_equal(a[0][f], b[0][f]) for f in a[0].dtype.names if f not in ("slot", "phase_cycles"))
    same = same and all(np.array_equal(a[1][i, :a[0]["n_final"][i]], b[1][i, :a[0]["n_final"][i]]) for i in range(len(a[0])))
    out["identical_results"] = bool(same)
    out["speedup"] = out["without_lookahead"]["ms_best"] / out["with_lookahead"]["ms_best"]

    # the pops served from a record, per wave (instrumented instantiation, lookahead on)
    bp = path_planner.BatchPlanner(dm, max_nodes=bench.MAX_NODES, max_path=bench.MAX_PATH, mode=1, lookahead=True)
    bp.plan_dev(stt, got, True, profile=True)
    res, _, _ = bp.plan_dev(stt, got, True, profile=True)
    torch.cuda.synchronize()
    rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)
    ph = rec["phase_cycles"].astype(np.float64)
    long_ = rec["n_pops"] >= int(0.9 * cap)
    hits, pops = ph[long_, 63].sum(), float(rec["n_pops"][long_].sum())
    out["record_pops_of_capped_problems"] = {
        "problems": int(long_.sum()), "pops": int(pops), "record_pops": int(hits), "record_pop_frac": hits / max(pops, 1),
        "cycles_since_pop_start_per_wave": {"children_ready": [ph[long_, 16 + 5 * w + 0].sum() / max(hits, 1) for w in range(8)],
                                            "words_done": [ph[long_, 16 + 5 * w + 1].sum() / max(hits, 1) for w in range(8)],
                                            "shot_chain_done": [ph[long_, 16 + 5 * w + 2].sum() / max(hits, 1) for w in range(8)],
                                            "resolution_done": [ph[long_, 16 + 5 * w + 3].sum() / max(hits, 1) for w in range(8)],
                                            "end_of_pop": [ph[long_, 16 + 5 * w + 4].sum() / max(hits, 1) for w in range(8)]},
        "resolution_classify_write_push_cycles_per_pop": [ph[long_, k].sum() / pops for k in (2, 3, 10)],
        "note": "instrumented kernel (s_memtime reads cost ~10 %); the timeline covers record pops only, wave 0 resolves the children"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
