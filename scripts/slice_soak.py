"""Soak of the time-sliced group forms' inter-workgroup hand-over (park area, resume ring, agent-scope fences): --launches
consecutive launches per kernel form of a batch with more problems than the form has groups, SHORT slices (every long search
changes groups dozens of times), every launch's result digested (status, pops, counters, way-points of every problem) and
compared with the digest of the unsliced launch.

    python scripts/slice_soak.py [--launches 100] [--slice-pops 4]
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=100)
ap.add_argument("--slice-pops", type=int, default=4)
ap.add_argument("--cap", type=int, default=300)
ap.add_argument("--extra", type=int, default=512, help="problems beyond the form's group count")
a = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from automatedvaletparking_amd import config, costmap, workloads, _native, path_planner  # noqa: E402

cfg, veh = config.default_config(), costmap.Vehicle()
m = workloads.case_map(1, cfg)
dm = _native.DeviceMap(m, veh, cfg, device=0, max_pops=a.cap)
L = _native.lib()
FIELDS = ("status", "n_pops", "n_astar", "n_final", "n_checks", "n_rs", "n_closed", "n_open", "h_cells", "h_misses", "global_index", "n_nodes", "rs_L")


def digest(res, paths, n):
    rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:n]
    pa = paths.cpu().numpy()
    h = hashlib.sha256()
    for k in FIELDS:
        h.update(np.ascontiguousarray(rec[k]).tobytes())
    for i in range(n):
        h.update(np.ascontiguousarray(pa[i, :int(rec["n_final"][i])]).tobytes())
    return h.hexdigest()[:16], int((rec["n_pops"] > a.slice_pops).sum()), int(rec["n_pops"].sum())


out = {"slice_pops": a.slice_pops, "pop_cap": a.cap, "launches_per_form": a.launches, "forms": {}}
for mode, name in ((4, "four waves per problem"), (3, "a pair of waves per problem"), (2, "one wave per problem")):
    n = int(L.avp_plan_slots(dm.h, C.c_int32(mode))) + a.extra
    st, go = workloads.sample_pairs(m, dm.check_batch, n, np.random.default_rng(workloads.SEED + mode))
    stt, got = dm.dev_tensor(st), dm.dev_tensor(go)
    off = path_planner.BatchPlanner(dm, max_nodes=8192, max_path=256, mode=mode, time_slice=False)
    r, p, _ = off.plan_dev(stt, got)
    torch.cuda.synchronize()
    want, parked, pops = digest(r, p, n)
    del off
    on = path_planner.BatchPlanner(dm, max_nodes=8192, max_path=256, mode=mode, time_slice=True, slice_pops=a.slice_pops)
    bad, ms = 0, []
    for k in range(a.launches):
        t0 = time.perf_counter()
        r, p, _ = on.plan_dev(stt, got)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
        bad += digest(r, p, n)[0] != want
    out["forms"][name] = {"problems": n, "searches_longer_than_a_slice": parked, "pops": pops, "slices_at_least": pops // a.slice_pops - n,
                          "digest_unsliced": want, "launches_with_a_different_digest": int(bad), "time_sliced": bool(on.last_time_sliced),
                          "ms_median": float(np.median(ms)), "ms_min": float(min(ms)), "ms_max": float(max(ms))}
    del on
out["lib"] = os.path.basename(_native.LIB_PATH)
print(json.dumps(out))
