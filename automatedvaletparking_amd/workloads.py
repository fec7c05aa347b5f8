"""The synthetic problem sets of BASELINE.json's configs (C2 .. C5), defined once: bench.py, the GPU
parity tests and the CPU parity-chain test all draw the same problems from here.

Pure host code. The footprint check used for rejection sampling is passed in (`checker(map) -> callable(poses) -> hits`):
the HIP check kernel in bench.py and the GPU tests, the CPU oracle in the CPU-only tests -- the two agree bit for bit
(tests/test_gpu_check.py), so the sets are the same either way (tests/test_gpu_workloads.py).
"""
import os
import tempfile

import numpy as np

from . import costmap, sampling

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = os.path.join(ROOT, "data", "BenchmarkCases")
SEED = 20260927


def sample_pairs(m, check_batch, n_pairs, rng, chunk=None):
    """SURVEY 8(d) sampler: footprint-free poses outside every obstacle polygon (obstacles are hollow in the costmap),
    paired up start / goal. chunk: candidates drawn per round (part of the set's definition: the generator is consumed
    per round)."""
    chunk = chunk or 8 * min(n_pairs, 512)
    free = []
    while len(free) < 2 * n_pairs:
        cand = sampling.sample_free_poses(m.boundary, m.case.obs, chunk, rng, margin=6.0, reject=False)
        hit = check_batch(cand)
        free += [p for p, h in zip(cand, hit) if not h and sampling.pose_is_free(p[0], p[1], p[2], m.case.obs)]
    poses = np.array(free[:2 * n_pairs])
    return poses[0::2], poses[1::2]


def case_map(k, cfg, device=None):
    return costmap.Map(file=os.path.join(CASES, f"Case{k}.csv"), discrete_size=cfg["map_discrete_size"], device=device)


def case_maps(ks, cfg, device=None):
    """Several BenchmarkCases at once: batched ingest on the device (Map.load_batch: one rasteriser launch for all of them) or, with
    device=None, the host rasteriser file by file. Same maps either way."""
    files = [os.path.join(CASES, f"Case{k}.csv") for k in ks]
    if device is None:
        return [costmap.Map(file=f, discrete_size=cfg["map_discrete_size"]) for f in files]
    return costmap.Map.load_batch(files, discrete_size=cfg["map_discrete_size"], device=device)


def case1_pairs(cfg, checker, n, device=None):
    """config[1] (n = 256) and north_star's target batch (n = 4096): Case1 map, n random pairs, seed 20260927."""
    m = case_map(1, cfg, device)
    st, go = sample_pairs(m, checker(m), n, np.random.default_rng(SEED))
    return m, st, go


def c3_map_pairs(k, cfg, checker, pairs=128, device=None, m=None):
    """config[2], map k of 20: BenchmarkCase k (or the already built map m), `pairs` random pairs, seed 20260927 + k."""
    m = m if m is not None else case_map(k, cfg, device)
    st, go = sample_pairs(m, checker(m), pairs, np.random.default_rng(SEED + k), chunk=8 * pairs)
    return m, st, go


def c4_map(device=None):
    """config[3]: 24 m x 24 m, discrete_size 0.12 -> 200 x 200 nodes, 32 regular n-gons (seed 4)."""
    with tempfile.TemporaryDirectory() as td:
        polys = sampling.synthetic_polygon_map(seed=4)
        p = os.path.join(td, "c4.csv")
        sampling.write_tpcap_csv(p, (12.0, 12.0, 0.0), (12.0, 12.0, 0.5), polys)
        m = costmap.Map(file=p, discrete_size=0.12, device=device)
    return m, polys


def c4_poses(m, n=4096):
    """the pure check_batch stress of config[3]: n poses without rejection (and the generator, for the plan pairs)."""
    rng = np.random.default_rng(4)
    poses = np.stack([rng.uniform(m.boundary[0] + 3, m.boundary[1] - 3, n), rng.uniform(m.boundary[2] + 3, m.boundary[3] - 3, n),
                      rng.uniform(-np.pi, np.pi, n)], 1)
    return poses, rng


def c4_plan_pairs(m, checker, pairs=256):
    _, rng = c4_poses(m, 4096)
    return sample_pairs(m, checker(m), pairs, rng, chunk=8 * pairs)


def c5_problems(cfg, n=1024, device=None):
    """config[4]: parking lot (2 x 60 cars, one empty bay = goal), n starts in the aisle (seed 5), flag_radius 1e9 so
    that the Reeds-Shepp shot runs at every pop. -> (map, config, starts, goals, obstacles)"""
    obs, goal, aisle = sampling.parking_lot_map()
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "c5.csv")
        sampling.write_tpcap_csv(p, (aisle[0] + 8.0, 0.5 * (aisle[2] + aisle[3]), 0.0), goal, obs)
        m = costmap.Map(file=p, discrete_size=cfg["map_discrete_size"], device=device)
    c5 = dict(cfg)
    c5["flag_radius"] = 1e9
    rng = np.random.default_rng(5)
    starts = np.stack([rng.uniform(m.boundary[0] + 4, m.boundary[1] - 4, n), rng.uniform(aisle[2] + 1.2, aisle[3] - 1.2, n),
                       rng.choice([0.0, np.pi], n) + rng.normal(0, 0.05, n)], 1)
    goals = np.tile(np.array(goal), (n, 1))
    return m, c5, starts, goals, obs
