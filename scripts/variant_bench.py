"""Kernel-variant A/B harness (diagnostics): times config[1] (Case1 map, 256 random pairs, pop cap 1000) and a
saturating 2048-problem batch on the library named by --lib (default: the product library), prints a digest of the
results (any two variants must print the same digest: status, pops, counters and way-points of every problem) and,
through the instrumented kernel, the per-phase cycles of the capped problems.

    python scripts/variant_bench.py [--lib path/to/libavp_hip_<variant>.so] [--steps 5] [--no-profile]
"""
import argparse
import hashlib
import json
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--big", type=int, default=2048)
ap.add_argument("--no-profile", action="store_true")
ap.add_argument("--big-mode", type=int, default=0)
ap.add_argument("--slice", choices=["auto", "on", "off"], default="auto", help="time slicing of the group forms for the big batch")
ap.add_argument("--slice-pops", type=int, default=None)
ap.add_argument("--cap", type=int, default=1000, help="pop cap of every search")
ap.add_argument("--no-big", action="store_true", help="skip the saturating batch")
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
dm = _native.DeviceMap(m, veh, cfg, max_pops=a.cap)
bp = path_planner.BatchPlanner(dm, max_nodes=16384, max_path=256)
rng = np.random.default_rng(20260927)
free = []
while len(free) < 512:
    cand = sampling.sample_free_poses(m.boundary, m.case.obs, 2048, rng, margin=6.0, reject=False)
    hit = dm.check_batch(cand)
    free += [p for p, h in zip(cand, hit) if not h and sampling.pose_is_free(p[0], p[1], p[2], m.case.obs)]
poses = np.array(free[:512])
starts, goals = poses[0::2], poses[1::2]
st, go = dm.dev_tensor(starts), dm.dev_tensor(goals)


def timed(s_t, g_t, steps):
    bp.plan_dev(s_t, g_t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        res, paths, _ = bp.plan_dev(s_t, g_t)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, res, paths


ms, res, paths = timed(st, go, a.steps)
rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:256]
pa = paths.cpu().numpy()
h = hashlib.sha256()
for k in ("status", "n_pops", "n_astar", "n_final", "n_checks", "n_rs", "n_closed", "n_open", "h_cells", "global_index", "n_nodes", "rs_L"):
    h.update(np.ascontiguousarray(rec[k]).tobytes())
for i in range(256):
    h.update(np.ascontiguousarray(pa[i, :int(rec["n_final"][i]), :]).tobytes())
out = {"lib": os.path.basename(_native.LIB_PATH), "c2_ms": round(ms, 3), "c2_plans_per_s": round(256 / ms * 1e3, 1),
       "c2_expansions_per_s": round(float(rec["n_pops"].sum()) / ms * 1e3), "solved": int((rec["status"] == 0).sum()),
       "capped": int((rec["status"] == 4).sum()), "digest": h.hexdigest()[:16], "cap": a.cap}
if not a.no_big:
    # saturating batch: the 256 starts against rolled goals
    rep = a.big // 256
    bs = torch.cat([st] * rep).contiguous()
    bg = torch.cat([go.roll(k, 0) for k in range(rep)]).contiguous()
    bp_big = path_planner.BatchPlanner(dm, max_nodes=16384, max_path=256, mode=a.big_mode, time_slice={"auto": None, "on": True, "off": False}[a.slice], slice_pops=a.slice_pops)
    bp_save, bp = bp, bp_big
    msb, resb, _ = timed(bs, bg, 2)
    bp = bp_save
    recb = resb.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:a.big]
    hb = hashlib.sha256()
    for k in ("status", "n_pops", "n_astar", "n_final", "n_checks", "n_rs", "n_closed", "n_open", "h_cells", "global_index", "n_nodes", "rs_L"):
        hb.update(np.ascontiguousarray(recb[k]).tobytes())
    out.update({"big_digest": hb.hexdigest()[:16], "big_mode": a.big_mode, "time_sliced": bp_big.last_time_sliced, "slice_pops": a.slice_pops, "big_n": a.big, "big_ms": round(msb, 3), "big_plans_per_s": round(a.big / msb * 1e3, 1),
                "big_expansions_per_s": round(float(recb["n_pops"].sum()) / msb * 1e3), "big_solved": int((recb["status"] == 0).sum())})
if not a.no_profile and hasattr(_native.lib(), "avp_plan_batch_profile"):
    try:
        resp, _, _ = bp.plan_dev(st, go, profile=True)
        torch.cuda.synchronize()
        rp = resp.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:256]
        cap = rp["status"] == 4
        ph = rp["phase_cycles"].astype(np.float64)
        names = ["init", "pop", "res_classify", "res_write", "resolve||shot", "children||substeps", "rs_words..replay", "slow_resolve", "(sweep)", "finish",
                 "res_push", "rs_words_only", "children_w0", "shot_round0", "shot_all", "pop_ahead"]
        out["phase_cyc_per_pop"] = {n: round(float(ph[cap, k].mean() / 1000), 0) for k, n in enumerate(names) if n != "-"}
        out["cyc_per_pop"] = round(float(ph[cap][:, [1, 4, 5, 6, 7]].sum(axis=1).mean() / 1000), 0)
        out["wave_arrivals_kcyc"] = [[round(float(ph[cap, 16 + 5 * w + k].mean() / 1e6), 1) for k in range(5)] for w in range(8)]
        out["probes_cyc"] = {n: round(float(ph[cap, 56 + k].mean() / 1000)) for k, n in enumerate(
            ["chk_setup", "chk_gather", "chk_all", "shot_accept", "shot_fold", "shot_book", "origins", "-"]) if n != "-"}
        one = rp["n_pops"] == 1
        out["one_pop_total_cyc"] = round(float(ph[one][:, [0, 1, 4, 5, 6, 7, 9]].sum(axis=1).mean()))
    except Exception as e:     # an old library without the entry
        out["profile_error"] = str(e)[:80]
print(json.dumps(out))
