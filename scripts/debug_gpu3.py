import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import gold, case_map_from_gold, CASES
from automatedvaletparking_amd import costmap, config, _native, path_planner
from oracle import oracle
import test_gpu_plan as T
cfg = config.default_config(); veh = costmap.Vehicle()
g = np.load(os.path.join(ROOT, "tests", "golden", sys.argv[1]))
m, st, go = T._gold_problem(g)
cap = 30000
dm = _native.DeviceMap(m, veh, cfg, max_pops=cap)
bp = path_planner.BatchPlanner(dm, n_slots=1, max_nodes=1 << 19)
res = bp.plan(st[None, :], go[None, :], max_trace=cap)[0]
o = oracle.Oracle(m, veh, cfg, max_pops=cap)
with oracle.portable_libm():
    w = o.plan(st, go, max_trace=cap)
print("gpu", res.status, res.n_pops, res.counters)
print("orc", w["status"], w["n_pops"], {k: w[k] for k in ("n_closed", "n_open", "n_checks", "n_rs", "n_dij_calls", "n_closed_hit", "n_open_hit", "n_improved", "n_collided", "n_pushed")})
t = res.trace; wt = w["trace"]
# per pop: in_radius?
gx, gy = go[0], go[1]
d = np.sqrt((t[:, 3] - gx) ** 2 + (t[:, 4] - gy) ** 2)
print("pops in radius", int((d < 18).sum()), "of", len(d), "min/max dist", d.min(), d.max())
print("near threshold", d[np.abs(d - 18) < 1e-3])
