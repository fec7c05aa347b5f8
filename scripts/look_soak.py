"""Soak of the expansion lookahead's inter-workgroup hand-offs (diagnostics): --launches consecutive launches
of bench.py's config[1] problem set with the lookahead on, every launch's result digested (status, pops, counters,
way-points of every problem) and compared with the digest of a launch WITHOUT the lookahead. Run once per build:

    python scripts/look_soak.py                                   # the product library
    python scripts/look_soak.py --lib .../libavp_hip_<v>.so       # PL_LOOK_SLEEP / PL_LOOK_WAIT / PL_LOOK_ATOMICS / PL_LOOK_FAULT builds
"""
import argparse
import hashlib
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--launches", type=int, default=300)
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--entries-log2", type=int, default=0, help="records in the lookahead's store = 2^this (0 = the library's default): a small store makes tags collide and entries change hands")
a = ap.parse_args()
if a.lib:
    os.environ["AVP_HIP_LIB"] = os.path.abspath(a.lib)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from automatedvaletparking_amd import config, costmap, workloads, _native, path_planner  # noqa: E402

cfg, veh = config.default_config(), costmap.Vehicle()
m = workloads.case_map(1, cfg)
dm = _native.DeviceMap(m, veh, cfg, device=0, max_pops=1000)
st, go = workloads.sample_pairs(m, dm.check_batch, a.n, np.random.default_rng(workloads.SEED))
stt, got = dm.dev_tensor(st), dm.dev_tensor(go)
FIELDS = ("status", "n_pops", "n_astar", "n_final", "n_checks", "n_rs", "n_closed", "n_open", "h_cells", "h_misses", "global_index", "n_nodes", "rs_L")


def digest(res, paths):
    rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:a.n]
    pa = paths.cpu().numpy()
    h = hashlib.sha256()
    for k in FIELDS:
        h.update(np.ascontiguousarray(rec[k]).tobytes())
    for i in range(a.n):
        h.update(np.ascontiguousarray(pa[i, :int(rec["n_final"][i])]).tobytes())
    return h.hexdigest()[:16]


off = path_planner.BatchPlanner(dm, max_nodes=16384, max_path=256, mode=1, lookahead=False)
r, p, _ = off.plan_dev(stt, got)
torch.cuda.synchronize()
want = digest(r, p)
on = path_planner.BatchPlanner(dm, max_nodes=16384, max_path=256, mode=1, lookahead=True, look_entries_log2=a.entries_log2)
bad, used, ms = 0, [], []
for k in range(a.launches):
    t0 = time.perf_counter()
    r, p, _ = on.plan_dev(stt, got)
    torch.cuda.synchronize()
    ms.append((time.perf_counter() - t0) * 1e3)
    bad += digest(r, p) != want
    used.append(int(on._look[:1024].cpu().numpy().view(np.uint64)[8]))
print(json.dumps({"lib": os.path.basename(_native.LIB_PATH), "launches": a.launches, "problems": a.n, "digest_without_lookahead": want,
                  "launches_with_a_different_digest": int(bad), "lookahead_used": bool(on.last_lookahead),
                  "ms_median": float(np.median(ms)), "ms_min": float(min(ms)), "ms_max": float(max(ms)),
                  "records_used_min_median_max": [int(min(used)), int(np.median(used)), int(max(used))],
                  "record_store_entries": 1 << (a.entries_log2 or 18), "lookahead_workspace_bytes": int(on._look.numel()),
                  "last_launch": {"claims_refused_entry_busy": int(on._look[:1024].cpu().numpy().view(np.uint64)[78]),
                                  "copies_refused_by_seqlock": int(on._look[:1024].cpu().numpy().view(np.uint64)[79])}}))
