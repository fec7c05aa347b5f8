"""A/B of the expansion lookahead on the bench's config[1] workload: identical results, time with and without.
usage: python scripts/look_bench.py [n_problems] [cap]"""
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
    from automatedvaletparking_amd import _native, path_planner
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    from automatedvaletparking_amd import config, costmap
    cfg, veh = config.default_config(), costmap.Vehicle()
    m = costmap.Map(file=os.path.join(bench.CASES, "Case1.csv"), discrete_size=cfg["map_discrete_size"])
    dm0 = _native.DeviceMap(m, veh, cfg, device=0, max_pops=cap)
    st, go = bench.sample_pairs(m, dm0, n, np.random.default_rng(20260927))
    g = bench.Group(m, veh, cfg, st, go, 0, cap, mode=1)
    dm = g.bp.dm
    out = {}
    for name, look in (("off", False), ("on", True), ("off2", False), ("on2", True)):
        bp = path_planner.BatchPlanner(dm, max_nodes=g.bp.max_nodes, mode=1, lookahead=look)
        bp.plan_dev(g.st_t, g.go_t, want_paths=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            res, paths, _ = bp.plan_dev(g.st_t, g.go_t, want_paths=True)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)
        out[name] = (rec, paths.cpu().numpy(), min(ts))
        if bp._look is not None:
            c = bp._look[:512].cpu().numpy().view(np.uint64)
            print("   jobs posted %d, records made %d, records used %d, helpers %d" % (c[0], c[24], c[8], c[48]))
        print(name, "lookahead used:", bp.last_lookahead, "ms:", [round(t, 2) for t in ts], "pops:", int(rec["n_pops"].sum()), flush=True)
    a, b = out["off"], out["on"]
    same = True
    for f in a[0].dtype.names:
        if f in ("slot", "phase_cycles"):
            continue
        if not np.array_equal(a[0][f], b[0][f]):
            same = False
            print("FIELD DIFFERS:", f, int((a[0][f] != b[0][f]).sum() if a[0][f].ndim == 1 else -1))
    for i in range(len(a[0])):
        nf = int(a[0]["n_final"][i])
        if not np.array_equal(a[1][i, :nf], b[1][i, :nf]):
            same = False
            print("PATH DIFFERS:", i)
            break
    print("identical:", same, " speedup: %.3f" % (a[2] / b[2]))




def profile_record_pops():
    """per-wave timeline of the pops served from a record (instrumented instantiation with the lookahead on)"""
    import torch
    import bench
    from automatedvaletparking_amd import _native, path_planner, config, costmap
    cfg, veh = config.default_config(), costmap.Vehicle()
    m = costmap.Map(file=os.path.join(bench.CASES, "Case1.csv"), discrete_size=cfg["map_discrete_size"])
    dm = _native.DeviceMap(m, veh, cfg, device=0, max_pops=1000)
    st, go = bench.sample_pairs(m, dm, 256, np.random.default_rng(20260927))
    bp = path_planner.BatchPlanner(dm, max_nodes=bench.MAX_NODES, mode=1, lookahead=True)
    stt, got = dm.dev_tensor(st), dm.dev_tensor(go)
    bp.plan_dev(stt, got, True, profile=True)
    res, _, _ = bp.plan_dev(stt, got, True, profile=True)
    torch.cuda.synchronize()
    rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)
    ph = rec["phase_cycles"].astype(np.float64)
    long_ = rec["n_pops"] >= 900
    hits = ph[long_, 63].sum()
    c = bp._look[:512].cpu().numpy().view(np.uint64)
    print("   jobs posted %d, records made %d, records used %d, helpers %d" % (c[0], c[24], c[8], c[48]))
    pops = rec["n_pops"][long_].sum()
    print("long problems: %d, pops %d, record pops %d (%.1f %%)" % (long_.sum(), pops, hits, 100 * hits / pops))
    print("record pops, cycles per pop at barrier k, per wave (0 = children ready, 3 = resolution done, 4 = end):")
    for wv in range(8):
        print("  wave %d:" % wv, " ".join("%8.0f" % (ph[long_, 16 + 5 * wv + k].sum() / max(hits, 1)) for k in (0, 3, 4)))
    print("PH_POP per pop %.0f; resolve classify/write/push per pop %.0f %.0f %.0f" % (
        ph[long_, 1].sum() / pops, ph[long_, 2].sum() / pops, ph[long_, 3].sum() / pops, ph[long_, 10].sum() / pops))


if len(sys.argv) > 1 and sys.argv[1] == "profile":
    profile_record_pops()
    sys.exit(0)

if __name__ == "__main__":
    main()
