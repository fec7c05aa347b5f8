"""The shipped `path_planner.split_path` / `split_path_batch` (reference `path_plan/path_planner.py:112-192`) against
EVERY golden that holds the reference's own split output: the reference's `final_path` goes in, its `split_concat` /
`split_len` / `change_gear` (or its IndexError when the path has no gear change, :181) must come out. The extension
poses are collision-checked by the HIP kernel (distance or two-circle checker as the fixture's config says)."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLD
from test_gpu_plan import _gold_problem

pytestmark = pytest.mark.gpu

SPLIT_GOLDENS = sorted(p for p in glob.glob(os.path.join(GOLD, "g*.npz"))
                       if os.path.basename(p).startswith(("g6_", "g7_", "g8_synth_c", "g10_")) and "split_error" in np.load(p).files)


def _setup(path, cfg, vehicle):
    from automatedvaletparking_amd import collision_check
    g = np.load(path)
    m, _, _ = _gold_problem(g)
    c2 = dict(cfg)
    if "synth_c5" in path:
        c2["flag_radius"] = 1e9
    if "cfg_json" in g.files:
        c2.update(json.loads(str(g["cfg_json"])))
    cls = collision_check.two_circle_checker if c2["collision_check"] == "circle" else collision_check.distance_checker
    return g, c2, cls(map=m, vehicle=vehicle, config=c2)


def _as_lists(P):
    return [[float(v) for v in row] for row in P]


@pytest.mark.parametrize("path", SPLIT_GOLDENS)
def test_split_path_vs_reference(path, vehicle, cfg):
    from automatedvaletparking_amd import path_planner
    g, c2, chk = _setup(path, cfg, vehicle)
    fp = _as_lists(g["final_path"])
    if str(g["split_error"]) == "IndexError":
        with pytest.raises(IndexError):
            path_planner.split_path(fp, c2, vehicle, chk)
        return
    seg, gear = path_planner.split_path(fp, c2, vehicle, chk)
    assert gear == int(g["change_gear"])
    assert [len(s) for s in seg] == list(g["split_len"])
    assert np.array_equal(np.array(sum(seg, []), dtype=np.float64).reshape(-1, 3), g["split_concat"])
    assert all(isinstance(q, list) and isinstance(q[0], float) for s in seg for q in s)


def test_split_path_batch_all_goldens(vehicle, cfg):
    """The batched form: per map/config group, all paths' extension poses in one check launch."""
    from automatedvaletparking_amd import path_planner
    assert len(SPLIT_GOLDENS) >= 80
    n_err = 0
    for path in SPLIT_GOLDENS:
        g, c2, chk = _setup(path, cfg, vehicle)
        # the fixture's path twice + a reversed copy: batch entries must not influence each other
        fp = _as_lists(g["final_path"])
        outs = path_planner.split_path_batch([fp, fp[::-1], fp], c2, vehicle, chk)
        for o in (outs[0], outs[2]):
            if str(g["split_error"]) == "IndexError":
                assert isinstance(o, IndexError)
                n_err += 1
            else:
                seg, gear = o
                assert gear == int(g["change_gear"]) and [len(s) for s in seg] == list(g["split_len"])
                assert np.array_equal(np.array(sum(seg, []), dtype=np.float64).reshape(-1, 3), g["split_concat"])
    assert n_err > 0          # the IndexError path is exercised


def test_plan_batch_split_matches_goldens(vehicle, cfg):
    """plan_batch(..., split=True) end to end on the Case1 goldens: segments of the planned (not the stored) paths equal
    the reference's to 1e-6 with the same lengths, and `split_error` marks the no-gear-change paths."""
    from automatedvaletparking_amd import path_planner
    from conftest import case_map_from_gold
    files = [p for p in SPLIT_GOLDENS if "g7_random_case1_" in p or p.endswith("g6_trace_case1.npz")]
    gs = [np.load(p) for p in files]
    m = case_map_from_gold(1)
    st = np.array([g["start"] if "start" in g.files else g["map_poses"][:3] for g in gs])
    go = np.array([g["goal"] if "goal" in g.files else g["map_poses"][3:] for g in gs])
    pl = path_planner.PathPlanner(config=cfg, map=m, vehicle=vehicle)
    res = pl.plan_batch(st, go, split=True)
    for r, g in zip(res, gs):
        assert r.status == 0
        if str(g["split_error"]) == "IndexError":
            assert r.split_error == "IndexError" and r.segments is None
        else:
            assert r.change_gear == int(g["change_gear"]) and [len(s) for s in r.segments] == list(g["split_len"])
            assert np.abs(np.array(sum(r.segments, [])) - g["split_concat"]).max() < 1e-6


def test_main_driver_batch_writes_segments(tmp_path):
    """`main.py --batch N`: one Planned_<case>_<i>.tsv per solved problem with a gear change, each equal to the scalar
    split_path of that problem's path."""
    from automatedvaletparking_amd import main as drv, path_planner, costmap, config as cfgmod, collision_check
    out = tmp_path / "pre"
    assert drv.main(["--case_name", "Case1", "--out_dir", str(out), "--batch", "24", "--seed", "3", "--max_pops", "400"]) == 0
    z = np.load(str(out / "Batch_Case1.npz"))
    files = sorted(glob.glob(str(out / "Planned_Case1_*.tsv")))
    assert files and len(files) == int((z["change_gear"] >= 0).sum())
    cfg = cfgmod.default_config()
    veh = costmap.Vehicle()
    m = costmap.Map(file=os.path.join(os.path.dirname(GOLD), "..", "data", "BenchmarkCases", "Case1.csv"), discrete_size=cfg["map_discrete_size"], device="cuda")
    chk = collision_check.distance_checker(map=m, vehicle=veh, config=cfg)
    for f in files[:6]:
        i = int(os.path.basename(f)[len("Planned_Case1_"):-4])
        rows = np.loadtxt(f, skiprows=1).reshape(-1, 4)
        seg, gear = path_planner.split_path([[float(v) for v in q] for q in z[f"path_{i}"]], cfg, veh, chk)
        assert gear == int(z["change_gear"][i])
        assert np.array_equal(rows[:, 1:4], np.array(sum(seg, [])))
        assert [int((rows[:, 0] == k).sum()) for k in range(len(seg))] == [len(s) for s in seg]
