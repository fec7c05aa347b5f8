"""BASELINE.json configs 3-5 at full size: problem builders and the GPU-vs-oracle comparison shared by
tests/test_gpu_configs.py (collected by `pytest -m gpu`) and tests/config_sweep.py (the same runs as a script that
prints throughput lines). Test infrastructure: it uses the CPU oracle as the checker."""
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CAP = 300                      # pop cap of the full-size runs (the reference has none; see DESIGN.md "Workload")
THREADS = min(256, os.cpu_count() or 1)


def free_pairs(m, dm, n_pairs, rng):
    """SURVEY 8(d) sampler: footprint-free (GPU check) poses outside every obstacle polygon, paired up."""
    from automatedvaletparking_amd import workloads
    return workloads.sample_pairs(m, dm.check_batch, n_pairs, rng, chunk=8 * n_pairs)


def plan_and_compare(m, veh, cfg, starts, goals, cap=CAP, threads=THREADS, max_nodes=8192):
    """Plans the batch on the GPU (with traces) and on the oracle in its PINNED mode (platform glibc libm, the reference's
    Dijkstra pop order; one problem per host thread) and compares every observable field (tests/_parity.py).
    Returns (results, [(index, what differs)], gpu seconds, cpu seconds)."""
    import torch
    import _parity
    from automatedvaletparking_amd import _native, path_planner
    from oracle import oracle
    dm = _native.DeviceMap(m, veh, cfg, max_pops=cap)
    bp = path_planner.BatchPlanner(dm, max_nodes=max_nodes)
    bp.plan(starts[:8], goals[:8], max_trace=cap)                       # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = bp.plan(starts, goals, max_trace=cap)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    o = oracle.Oracle(m, veh, cfg, max_pops=cap)
    t1 = time.perf_counter()
    bad, h_diff = _parity.compare_pinned(o, res, starts, goals, cap, threads)
    t_cpu = time.perf_counter() - t1
    plan_and_compare.h_diff = getattr(plan_and_compare, "h_diff", 0) + len(h_diff)
    return res, bad, t_gpu, t_cpu


def c3_problems(k, cfg, veh, pairs=128, cap=CAP):
    """C3, map k of 20: BenchmarkCase k rasterised on the device, `pairs` random pairs, seed 20260927 + k."""
    from automatedvaletparking_amd import _native, costmap
    m = costmap.Map(file=os.path.join(ROOT, "data", "BenchmarkCases", f"Case{k}.csv"), discrete_size=cfg["map_discrete_size"], device="cuda")
    dm = _native.DeviceMap(m, veh, cfg, max_pops=cap)
    st, go = free_pairs(m, dm, pairs, np.random.default_rng(20260927 + k))
    return m, st, go


def c4_map():
    """C4: 24 m x 24 m, discrete_size 0.12 -> 200 x 200 nodes, 32 regular n-gons (seed 4)."""
    from automatedvaletparking_amd import costmap, sampling
    with tempfile.TemporaryDirectory() as td:
        polys = sampling.synthetic_polygon_map(seed=4)
        p = os.path.join(td, "c4.csv")
        sampling.write_tpcap_csv(p, (12.0, 12.0, 0.0), (12.0, 12.0, 0.5), polys)
        m = costmap.Map(file=p, discrete_size=0.12, device="cuda")
    return m, polys


def c4_poses(m, n=4096):
    rng = np.random.default_rng(4)
    poses = np.stack([rng.uniform(m.boundary[0] + 3, m.boundary[1] - 3, n), rng.uniform(m.boundary[2] + 3, m.boundary[3] - 3, n),
                      rng.uniform(-np.pi, np.pi, n)], 1)
    return poses, rng


def c4_stress_poses(m, o, n=4096, seed=44):
    """A second config[3] pose set that stresses the checkers instead of their early exit: the spec'd sampler (c4_poses,
    kept for the bench line) draws 99.5 % colliding poses on this map. Here, from 16 n uniform candidates classified by the
    ORACLE's distance checker (collision flag + number of obstacle points under the footprint's AABB,
    collision_check.py:55-69): 15 % near misses (collision free with the most near points: every one of them goes through
    the exact point test and none ends the pose early), 30 % other collision-free poses, 55 % colliding ones, shuffled.
    -> (poses, dict of the fractions)."""
    rng = np.random.default_rng(seed)
    k = 16 * n
    cand = np.stack([rng.uniform(m.boundary[0] + 1, m.boundary[1] - 1, k), rng.uniform(m.boundary[2] + 1, m.boundary[3] - 1, k),
                     rng.uniform(-np.pi, np.pi, k)], 1)
    hit, near = o.check_batch(cand, kind=0, want_near=True)
    hit = np.asarray(hit).astype(bool)
    free_idx = np.where(~hit)[0]
    n_nm, n_free = (15 * n) // 100, (30 * n) // 100
    assert len(free_idx) >= n_nm + n_free, "not enough collision-free candidates"
    by_near = free_idx[np.argsort(-near[free_idx], kind="stable")]
    nm = by_near[:n_nm]
    rest = by_near[n_nm:]
    other = rest[rng.permutation(len(rest))[:n_free]]
    coll = np.where(hit)[0][:n - n_nm - n_free]
    sel = np.concatenate([nm, other, coll])
    sel = sel[rng.permutation(len(sel))]
    info = dict(free_frac=float((~hit[sel]).mean()), near_miss_frac=n_nm / len(sel), near_miss_min_near_points=int(near[nm].min()),
                near_miss_mean_near_points=float(near[nm].mean()), colliding_frac=float(hit[sel].mean()))
    return cand[sel], info


def c5_problems(cfg, n=1024):
    """C5: parking lot (2 x 60 cars, one empty bay = goal), n starts in the aisle (seed 5), flag_radius 1e9 so that
    the Reeds-Shepp shot runs at every pop."""
    from automatedvaletparking_amd import costmap, sampling
    obs, goal, aisle = sampling.parking_lot_map()
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "c5.csv")
        sampling.write_tpcap_csv(p, (aisle[0] + 8.0, 0.5 * (aisle[2] + aisle[3]), 0.0), goal, obs)
        m = costmap.Map(file=p, discrete_size=cfg["map_discrete_size"], device="cuda")
    c5 = dict(cfg)
    c5["flag_radius"] = 1e9
    rng = np.random.default_rng(5)
    starts = np.stack([rng.uniform(m.boundary[0] + 4, m.boundary[1] - 4, n), rng.uniform(aisle[2] + 1.2, aisle[3] - 1.2, n),
                       rng.choice([0.0, np.pi], n) + rng.normal(0, 0.05, n)], 1)
    goals = np.tile(np.array(goal), (n, 1))
    return m, c5, starts, goals, obs
