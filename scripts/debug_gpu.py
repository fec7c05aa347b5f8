import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import gold, case_map_from_gold
from automatedvaletparking_amd import costmap, config, _native, path_planner
from oracle import oracle
cfg = config.default_config(); veh = costmap.Vehicle()
np.set_printoptions(precision=17, linewidth=250)
# ---- RS
g4 = gold("g4_rs.npz")
dm = _native.DeviceMap(case_map_from_gold(1), veh, cfg)
r = dm.rs_optimal_batch(g4["q0"], g4["q1"], maxc=float(g4["maxc"]), maxpts=g4["pts"].shape[1])
bad = np.where(r["status"] != 0)[0]
print("RS bad count", len(bad), "statuses", np.unique(r["status"], return_counts=True))
for i in bad[:5]:
    print(i, r["status"][i], g4["q0"][i], g4["q1"][i], "gold npts", g4["npts"][i], "L", g4["L"][i], "got npts", r["npts"][i])
ok = r["status"] == 0
print("types equal", np.array_equal(r["types"][ok], g4["types"][ok]), "L maxdiff", np.abs(r["L"][ok]-g4["L"][ok]).max(), "bit-eq frac", (r["L"][ok]==g4["L"][ok]).mean())
tm = np.where((r["types"][ok] != g4["types"][ok]).any(axis=1))[0]
print("type mismatches", len(tm))
for i in tm[:5]:
    j = np.where(ok)[0][i]
    print(j, g4["q0"][j], g4["q1"][j], r["types"][j], g4["types"][j], r["L"][j], g4["L"][j])
print("npts equal", np.array_equal(r["npts"][ok], g4["npts"][ok]))
# ---- plan
g = gold("g6_trace_case1.npz")
m = case_map_from_gold(1)
pl = path_planner.PathPlanner(config=cfg, map=m, vehicle=veh)
c = m.case
res = pl.plan_batch([[c.x0, c.y0, c.theta0]], [[c.xf, c.yf, c.thetaf]], max_trace=200)[0]
gp = g["pops"]; t = res.trace
print("status", res.status, "pops", res.n_pops, len(gp), res.counters)
n = min(len(t), len(gp))
d = np.where(~((t[:n, :7] == gp[:n, :7]).all(axis=1)) | (np.abs(t[:n, 7:9] - gp[:n, 7:9]).max(axis=1) > 1e-9))[0]
print("first differing pops", d[:10])
for i in d[:3]:
    print("GPU ", t[i]); print("GOLD", gp[i])
o = oracle.Oracle(m, veh, cfg)
w = o.plan([c.x0, c.y0, c.theta0], [c.xf, c.yf, c.thetaf], want_h=True)
print("oracle counters", {k: w[k] for k in ("n_closed", "n_open", "n_checks", "n_rs", "n_dij_calls", "n_dij_closed", "n_closed_hit", "n_open_hit", "n_improved", "n_collided", "n_pushed")})
