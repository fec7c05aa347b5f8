"""The L2-backed instantiations (maps whose column bitmaps + node coordinates do not fit LDS next to the kernel's state:
`check_distance_kernel<false>`, `plan_kernel<false, ..>`, `plan_wave_kernel<false, ..>`) and the largest BenchmarkCase
(Case19: 620 x 290 nodes, 5 696 obstacle points, tables staged in LDS) -- timing of the footprint kernel and of a plan batch
on each (diagnostics; VERDICT r2 #7/#8: every round-2 profile was Case1).

    python scripts/large_map_bench.py            # prints one JSON object
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from automatedvaletparking_amd import config, costmap, sampling, workloads, _native, path_planner  # noqa: E402

cfg, veh = config.default_config(), costmap.Vehicle()


def run(m, label, n_pairs=256, cap=300):
    dm = _native.DeviceMap(m, veh, cfg, max_pops=cap)
    rng = np.random.default_rng(3)
    b = m.boundary
    n = 1 << 20
    poses = np.stack([rng.uniform(b[0] + 6, b[1] - 6, n), rng.uniform(b[2] + 6, b[3] - 6, n), rng.uniform(-np.pi, np.pi, n)], 0)
    t = dm.dev_tensor(poses)
    out = dm.empty(n, torch.uint8)
    dm.check_batch_dev(t[0], t[1], t[2], out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dm.check_batch_dev(t[0], t[1], t[2], out=out)
    e1.record()
    torch.cuda.synchronize()
    cms = e0.elapsed_time(e1) / 5
    st, go = workloads.sample_pairs(m, dm.check_batch, n_pairs, np.random.default_rng(11), chunk=8 * n_pairs)
    res = {"map": label, "nx": int(m.cost_map.shape[0]), "ny": int(m.cost_map.shape[1]), "obstacle_points": int(dm.P),
           "table_bytes": int(((m.cost_map.shape[1] + 63) // 64 * m.cost_map.shape[0] + sum(m.cost_map.shape)) * 8),
           "check_ms_per_2^20": cms, "checks_per_s": n / (cms * 1e-3), "colliding_frac": float(out.float().mean().item())}
    for mode, key in ((1, "workgroup_form"), (4, "quad_form")):
        bp = path_planner.BatchPlanner(dm, max_nodes=8192, max_path=256, mode=mode, lookahead=False)
        stt, got = dm.dev_tensor(st), dm.dev_tensor(go)
        bp.plan_dev(stt, got)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r, _, _ = bp.plan_dev(stt, got)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        rec = r.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:n_pairs]
        res[key] = {"ms": ms, "pops": int(rec["n_pops"].sum()), "expansions_per_s": float(rec["n_pops"].sum()) / (ms * 1e-3),
                    "completed": int(((rec["status"] == 0) | (rec["status"] == 1)).sum()), "capped": int((rec["status"] == 4).sum())}
    return res


out = [run(workloads.case_map(19, cfg), "BenchmarkCases/Case19 (tables in LDS)")]
with tempfile.TemporaryDirectory() as td:
    polys = sampling.synthetic_polygon_map(seed=9, size=100.0, n_obs=400)
    p = os.path.join(td, "big.csv")
    sampling.write_tpcap_csv(p, (12.0, 12.0, 0.0), (88.0, 88.0, 0.5), polys)      # (the map spans start / goal +- 12 m: 100 m x 100 m)
    big = costmap.Map(file=p, discrete_size=0.1, device="cuda")
out.append(run(big, "synthetic 100 m x 100 m at 0.1 m, 400 polygons (tables through L1/L2: the <false> instantiations)"))
print(json.dumps(out))
