"""Per-phase cycle breakdown of plan_kernel on the bench workload (diagnostics)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from automatedvaletparking_amd import costmap, config, sampling, _native, path_planner
cfg = config.default_config(); veh = costmap.Vehicle()
m = costmap.Map(file=os.path.join(ROOT, "data", "BenchmarkCases", "Case1.csv"), discrete_size=cfg["map_discrete_size"])
dm = _native.DeviceMap(m, veh, cfg, max_pops=1000)
bp = path_planner.BatchPlanner(dm, max_nodes=16384, max_path=256)
rng = np.random.default_rng(20260927)
free = []
while len(free) < 512:
    cand = sampling.sample_free_poses(m.boundary, m.case.obs, 2048, rng, margin=6.0, reject=False)
    hit = dm.check_batch(cand)
    free += [p for p, h in zip(cand, hit) if not h and sampling.pose_is_free(p[0], p[1], p[2], m.case.obs)]
poses = np.array(free[:512]); starts, goals = poses[0::2], poses[1::2]
st, go = dm.dev_tensor(starts), dm.dev_tensor(goals)
for _ in range(2):
    res, _, _ = bp.plan_dev(st, go)
torch.cuda.synchronize()
rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:256]
names = ["init", "pop", "(unused)", "(unused)", "resolve||shot", "children||substeps", "rs_words..replay", "slow_resolve", "(sweep)", "finish"]
ph = rec["phase_cycles"].astype(np.float64)
cap = rec["status"] == 4
one = rec["n_pops"] == 1
print("capped problems:", int(cap.sum()), " mean cycles per pop by phase:")
for k, n in enumerate(names):
    print(f"  {n:18s} {ph[cap, k].mean() / 1000:10.1f} cyc/pop   one-pop problems total: {ph[one, k].mean():12.0f} cyc")
print("total cycles capped mean", ph[cap][:, [0,1,2,3,4,5,6,7,9]].sum(axis=1).mean(), " one-pop mean", ph[one][:, [0,1,2,3,4,5,6,7,9]].sum(axis=1).mean())
print("h_cells one-pop mean", rec["h_cells"][one].mean(), "misses", rec["h_misses"][one].mean())
mid = (rec["status"] == 0) & (rec["n_pops"] > 1)
for name, sel in (("capped", cap), ("solved >1 pop", mid)):
    if sel.any():
        pops = rec["n_pops"][sel].astype(np.float64)
        print(f"{name}: n_rs/pop {np.mean(rec['n_rs'][sel] / pops):.2f}  n_checks/pop {np.mean(rec['n_checks'][sel] / pops):.2f}  "
              f"nodes/pop {np.mean(rec['n_nodes'][sel] / pops):.2f}  closed/pop {np.mean(rec['n_closed'][sel] / pops):.2f}  "
              f"h_misses/pop {np.mean(rec['h_misses'][sel] / pops):.3f}  mean pops {pops.mean():.0f}")
