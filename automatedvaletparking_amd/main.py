"""Planning-only driver with the reference's command line (`main.py:143-171`): builds the map, runs
`PathPlanner.path_planning()` on the GPU and writes the way-points of every gear segment.

    python -m automatedvaletparking_amd.main --case_name Case1 [--config_name config] [--out_dir solution_preopt]
    python -m automatedvaletparking_amd.main --case_name Case1 --batch 256 --seed 1     # random pairs on that map

The reference's later stages (QP smoothing, velocity profile, OCP) consume `split_path` exactly as
this driver leaves it; they are outside this package."""
from __future__ import annotations

import argparse
import os
import time

import numpy as np

from . import config as cfgmod
from . import costmap, path_planner, sampling
from .record_solution import DataRecorder, waypoints_to_trajectory

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "BenchmarkCases")


def write_segments(path: str, split_path) -> None:
    """Tab-separated x, y, theta rows, one block per gear segment (segment index in column 0)."""
    with open(path, "w") as f:
        f.write("segment\tx\ty\ttheta\n")
        for k, seg in enumerate(split_path):
            for x, y, th in seg:
                f.write(f"{k}\t{float(x)!r}\t{float(y)!r}\t{float(th)!r}\n")


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="hybrid A* planning stage on MI355X")
    ap.add_argument("--config_name", type=str, default=None)
    ap.add_argument("--config_dir", type=str, default=None)
    ap.add_argument("--case_name", type=str, default="Case1")
    ap.add_argument("--case_dir", type=str, default=_DATA)
    ap.add_argument("--out_dir", type=str, default="solution_preopt")
    ap.add_argument("--batch", type=int, default=0, help="plan this many random start/goal pairs instead of the case's own")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max_pops", type=int, default=0)
    ap.add_argument("--solution_dir", type=str, default=None,
                    help="also write the way-points in the reference's Solution_<case>.csv layout (v, a, sigma, omega, t = 0)")
    ap.add_argument("--host_raster", action="store_true", help="rasterise the obstacle edges with numpy instead of on the GPU")
    a = ap.parse_args(argv)

    cfg = cfgmod.read_config(a.config_name, a.config_dir)
    park_map = costmap.Map(file=os.path.join(a.case_dir, a.case_name + ".csv"), discrete_size=cfg["map_discrete_size"],
                           device=None if a.host_raster else "cuda")
    ego = costmap.Vehicle()
    planner = path_planner.PathPlanner(config=cfg, map=park_map, vehicle=ego)
    os.makedirs(a.out_dir, exist_ok=True)
    t0 = time.perf_counter()
    if a.batch <= 0:
        original_path, info, split = planner.path_planning()
        out = os.path.join(a.out_dir, f"Planned_{a.case_name}.tsv")
        write_segments(out, split)
        if a.solution_dir:
            DataRecorder.record(save_path=a.solution_dir, save_name=a.case_name + ".csv", trajectory=waypoints_to_trajectory(original_path))
        print(f"{a.case_name}: {len(original_path)} way-points, {info['change_gear']} gear changes, "
              f"RS tail {''.join(info['rs_path'].ctypes)} L={info['rs_path'].L:.4f} m, {time.perf_counter() - t0:.3f} s -> {out}")
        return 0
    from . import _native
    dm = _native.DeviceMap(park_map, ego, cfg, max_pops=a.max_pops or 1000)
    rng = np.random.default_rng(a.seed)
    free = []
    while len(free) < 2 * a.batch:
        cand = sampling.sample_free_poses(park_map.boundary, park_map.case.obs, 8 * a.batch, rng, margin=6.0, reject=False)
        hit = dm.check_batch(cand)
        free += [p for p, h in zip(cand, hit) if not h and sampling.pose_is_free(p[0], p[1], p[2], park_map.case.obs)]
    poses = np.array(free[:2 * a.batch])
    planner._batch = path_planner.BatchPlanner(dm)
    # a_star_plan + split_path per problem = path_planning() (path_planner.py:45-56); every extension pose of the batch
    # is collision-checked in one launch
    res = planner.plan_batch(poses[0::2], poses[1::2], split=True)
    dt = time.perf_counter() - t0
    ok = sum(r.ok for r in res)
    n_seg = 0
    for i, r in enumerate(res):
        if r.segments is not None:       # one file per solved problem, the layout of the single-problem run (main.py:66-70 consumes it)
            write_segments(os.path.join(a.out_dir, f"Planned_{a.case_name}_{i}.tsv"), r.segments)
            n_seg += 1
    np.savez_compressed(os.path.join(a.out_dir, f"Batch_{a.case_name}.npz"), starts=poses[0::2], goals=poses[1::2],
                        status=np.array([r.status for r in res]), n_pops=np.array([r.n_pops for r in res]),
                        change_gear=np.array([-1 if r.change_gear is None else r.change_gear for r in res]),
                        **{f"path_{i}": r.final_path for i, r in enumerate(res) if r.ok})
    print(f"{a.case_name}: {a.batch} problems, {ok} solved, {n_seg} segment files (the others have no gear change: the "
          f"reference raises IndexError there), {sum(r.n_pops for r in res)} expansions, {dt:.3f} s")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
