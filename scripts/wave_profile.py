"""Per-phase cycles of the wave form (diagnostics): runs the instrumented instantiation of plan_wave_kernel
(avp_plan_batch_ex, mode 2 | 0x100) on a batch of --n problems of the bench workload (Case1 map, pop cap 1000) and prints,
for the capped searches, the shader cycles per pop by phase (lane 0 of the problem's wave), the collision passes and
Reeds-Shepp queries per pop, and the un-instrumented time of the same batch.

    python scripts/wave_profile.py [--lib path/to/libavp_hip_<variant>.so] [--n 4096]
"""
import argparse
import json
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--mode", type=int, default=2, help="2 = one wave per problem, 3 = a pair of waves per problem")
a = ap.parse_args()
if a.lib:
    os.environ["AVP_HIP_LIB"] = os.path.abspath(a.lib)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from automatedvaletparking_amd import costmap, config, sampling, _native, path_planner  # noqa: E402

cfg = config.default_config()
veh = costmap.Vehicle()
m = costmap.Map(file=os.path.join(ROOT, "data", "BenchmarkCases", "Case1.csv"), discrete_size=cfg["map_discrete_size"])
dm = _native.DeviceMap(m, veh, cfg, max_pops=1000)
rng = np.random.default_rng(20260927)
free = []
while len(free) < 512:
    cand = sampling.sample_free_poses(m.boundary, m.case.obs, 2048, rng, margin=6.0, reject=False)
    hit = dm.check_batch(cand)
    free += [p for p, h in zip(cand, hit) if not h and sampling.pose_is_free(p[0], p[1], p[2], m.case.obs)]
poses = np.array(free[:512])
st, go = dm.dev_tensor(poses[0::2]), dm.dev_tensor(poses[1::2])
rep = max(1, a.n // 256)
bs = torch.cat([st] * rep).contiguous()
bg = torch.cat([go.roll(k, 0) for k in range(rep)]).contiguous()
n = bs.shape[0]
bp = path_planner.BatchPlanner(dm, max_nodes=16384, max_path=256, mode=a.mode)


def timed(profile):
    bp.plan_dev(bs, bg, profile=profile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    res, _, _ = bp.plan_dev(bs, bg, profile=profile)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:n]


ms, rec = timed(False)
msp, rp = timed(True)
names = ["init", "pop", "children", "substeps", "rs", "shot", "resolve_fast", "resolve_slow", "finish", "n_passes", "n_rs_queries"]
cap = rp["status"] == 4
ph = rp["phase_cycles"].astype(np.float64)
pops = rp["n_pops"].astype(np.float64)
out = {"lib": os.path.basename(_native.LIB_PATH), "mode": a.mode, "problems_per_wg": int(_native.lib().avp_plan_group(a.mode)), "n": int(n), "ms": round(ms, 2),
       "ms_instrumented": round(msp, 2), "capped": int(cap.sum()), "expansions_per_s": round(float(rec["n_pops"].sum()) / ms * 1e3),
       "same_pops": bool(np.array_equal(rec["n_pops"], rp["n_pops"]))}
if cap.any():
    per = {nm: round(float((ph[cap, k] / pops[cap]).mean()), 1) for k, nm in enumerate(names)}
    out["capped_per_pop"] = per
    out["capped_cycles_per_pop"] = round(sum(per[nm] for nm in names[1:8]))
one = rp["n_pops"] == 1
if one.any():
    out["one_pop_problem_cycles"] = {nm: round(float(ph[one, k].mean())) for k, nm in enumerate(names[:9])}
print(json.dumps(out))
