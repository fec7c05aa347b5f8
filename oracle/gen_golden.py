#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE, build container only).

Imports the UNMODIFIED reference from /root/reference (one empty `shapely` stub, MPLBACKEND=Agg,
no bytecode written) and dumps numeric input/output vectors for the hybrid-A* hot path into
tests/golden/*.npz. Nothing from the reference's source text is written anywhere: fixtures hold
numbers only. The GPU box never runs this script (the reference does not exist there).

Recipe follows SURVEY.md Appendix B. The only acceleration applied to the reference is the
trace-identical O(1) `openlist_index.count` list subclass for `Dijkstra`.

Usage:  python oracle/gen_golden.py micro            # G1-G5 micro vectors (minutes)
        python oracle/gen_golden.py trace 1 2 3 ...   # G6 pop traces of BenchmarkCases (minutes-hours)
        python oracle/gen_golden.py random 1 8        # G7: 8 random start/goal problems on Case1
        python oracle/gen_golden.py synth             # G8 synthetic maps (C4, C5 small variants)
        python oracle/gen_golden.py variants 4        # G10: other config.yaml values on Case4
"""
import io
import os
import sys
import time
import types
import signal
import contextlib

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.modules.setdefault("shapely", types.ModuleType("shapely"))
sys.modules.setdefault("shapely.geometry", types.ModuleType("shapely.geometry"))
REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import numpy as np  # noqa: E402
import warnings  # noqa: E402

warnings.filterwarnings("ignore")

from config import read_config as ref_read_config  # noqa: E402
from map import costmap as ref_costmap  # noqa: E402
from collision_check import collision_check as ref_cc  # noqa: E402
from path_plan import rs_curve as ref_rs  # noqa: E402
from path_plan import compute_h as ref_h  # noqa: E402
from path_plan import hybrid_a_star as ref_has  # noqa: E402
from path_plan import path_planner as ref_pp  # noqa: E402

from automatedvaletparking_amd import sampling  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
CASES = os.path.join(REF, "BenchmarkCases")
TYPE_CODE = {"S": 0, "L": 1, "R": 2}


class _CountList(list):
    """list with O(1) count(): trace-identical replacement for Dijkstra.openlist_index."""

    def __init__(self):
        super().__init__()
        self._d = {}

    def append(self, v):
        super().append(v)
        self._d[v] = self._d.get(v, 0) + 1

    def count(self, v):
        return self._d.get(v, 0)


_orig_dij_init = ref_h.Dijkstra.__init__


def _dij_init(self, map):
    _orig_dij_init(self, map)
    self.openlist_index = _CountList()
    self._queries = []


_orig_compute_path = ref_h.Dijkstra.compute_path


def _compute_path(self, node_x, node_y):
    d, cl = _orig_compute_path(self, node_x, node_y)
    self._queries.append((float(node_x), float(node_y), int(d), len(cl)))
    return d, cl


ref_h.Dijkstra.__init__ = _dij_init
ref_h.Dijkstra.compute_path = _compute_path


def config():
    return ref_read_config.read_config("config")


def load_map(csv, cfg, start=None, goal=None, discrete=None):
    m = ref_costmap.Map(file=csv, discrete_size=cfg["map_discrete_size"] if discrete is None else discrete)
    if start is not None:
        m.case.x0, m.case.y0, m.case.theta0 = [float(v) for v in start]
    if goal is not None:
        m.case.xf, m.case.yf, m.case.thetaf = [float(v) for v in goal]
    return m


def map_arrays(m):
    ix, iy = np.where(m.cost_map == 255)
    S = int((m.boundary[1] - m.boundary[0]) / m._discrete_x)
    Sy = int((m.boundary[3] - m.boundary[2]) / m._discrete_y)
    return dict(boundary=np.asarray(m.boundary, dtype=np.float64), nx=m.cost_map.shape[0], ny=m.cost_map.shape[1],
                dx=float(m._discrete_x), dy=float(m._discrete_y), S=S, Sy=Sy,
                cells=np.stack([ix, iy], 1).astype(np.int32),
                xs=np.asarray(m.map_position[0]), ys=np.asarray(m.map_position[1]),
                poses=np.array([m.case.x0, m.case.y0, m.case.theta0, m.case.xf, m.case.yf, m.case.thetaf]))


class Timeout(Exception):
    pass


def _alarm(signum, frame):
    raise Timeout()


def run_plan(csv, cfg, start=None, goal=None, discrete=None, timeout=3600, record_checks=False):
    """One reference plan with a full pop trace. Returns a dict of numpy arrays."""
    t0 = time.time()
    m = load_map(csv, cfg, start, goal, discrete)
    veh = ref_costmap.Vehicle()
    out = {"status": "ok"}
    out.update({"map_" + k: v for k, v in map_arrays(m).items()})
    pops = []
    checks = []
    rs_calls = [0]
    n_checks = [0]

    signal.signal(signal.SIGALRM, _alarm)
    signal.alarm(int(timeout))
    pl = None
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            pl = ref_pp.PathPlanner(config=cfg, map=m, vehicle=veh)
            t_init = time.time() - t0
            astar = pl.planner
            og = astar.open_list.get

            def get_logged():
                n = og()
                gid = m.convert_position_to_index(n.x, n.y)
                pops.append((n.index, -1 if n.parent_index is None else n.parent_index, gid, float(n.x), float(n.y),
                             float(n.theta), float(n.g), float(n.h), float(n.f), 1.0 if n.forward else 0.0,
                             float("nan") if n.steering_angle is None else float(n.steering_angle)))
                return n

            astar.open_list.get = get_logged

            def wrap_check(chk):
                oc = chk.check

                def check(node_x, node_y, theta):
                    r = oc(node_x=node_x, node_y=node_y, theta=theta)
                    n_checks[0] += 1
                    if record_checks:
                        checks.append((float(node_x), float(node_y), float(theta), 1.0 if r else 0.0))
                    return r
                chk.check = check

            wrap_check(astar.collision_checker)
            wrap_check(pl.collision_checker)
            orig_opt = ref_rs.calc_optimal_path

            def opt_logged(*a, **k):
                rs_calls[0] += 1
                return orig_opt(*a, **k)

            ref_rs.calc_optimal_path = opt_logged
            try:
                final_path, astar_path, rs_path = pl.a_star_plan()
            finally:
                ref_rs.calc_optimal_path = orig_opt
            t_search = time.time() - t0 - t_init
            try:
                split, change_gear = pl.split_path(final_path)
                out["split_error"] = ""
            except IndexError:
                split, change_gear = [], -1
                out["split_error"] = "IndexError"
    except Timeout:
        out["status"] = "timeout"
        signal.alarm(0)
        out["pops"] = np.array(pops, dtype=np.float64).reshape(-1, 11)
        return out
    except AttributeError as e:  # no path: rs_path is None at path_planner.py:104
        out["status"] = "AttributeError"
        signal.alarm(0)
        out["pops"] = np.array(pops, dtype=np.float64).reshape(-1, 11)
        out["n_checks"] = n_checks[0]
        out["rs_calls"] = rs_calls[0]
        if pl is not None:
            out["n_closed"] = len(pl.planner.closed_list)
            out["n_open"] = len(pl.planner.open_list.queue)
        return out
    signal.alarm(0)
    astar = pl.planner
    out["pops"] = np.array(pops, dtype=np.float64).reshape(-1, 11)
    out["t_init"] = t_init
    out["t_search"] = t_search
    out["n_checks"] = n_checks[0]
    out["rs_calls"] = rs_calls[0]
    out["n_closed"] = len(astar.closed_list)
    out["n_open"] = len(astar.open_list.queue)
    out["global_index"] = astar.global_index
    out["final_path"] = np.array(final_path, dtype=np.float64).reshape(-1, 3)
    out["astar_path"] = np.array(astar_path, dtype=np.float64).reshape(-1, 3)
    out["rs_lengths"] = np.array(rs_path.lengths, dtype=np.float64)
    out["rs_types"] = np.array([TYPE_CODE[c] for c in rs_path.ctypes], dtype=np.int8)
    out["rs_L"] = float(rs_path.L)
    out["rs_xyyaw"] = np.stack([rs_path.x, rs_path.y, rs_path.yaw], 1).astype(np.float64)
    out["rs_dir"] = np.array(rs_path.directions, dtype=np.int8)
    out["split_concat"] = np.array(sum(split, []), dtype=np.float64).reshape(-1, 3)
    out["split_len"] = np.array([len(s) for s in split], dtype=np.int32)
    out["change_gear"] = change_gear
    dj = astar.heuristic
    out["h_queries"] = np.array(dj._queries, dtype=np.float64).reshape(-1, 4)
    out["h_closed_id"] = np.array([g.grid_id for g in dj.closedlist], dtype=np.int64)
    out["h_closed_dist"] = np.array([g.distance for g in dj.closedlist], dtype=np.int64)
    # closed/open list snapshot at termination (poses + flags), order preserved
    out["closed_nodes"] = np.array([(n.index, n.x, n.y, n.theta) for n in astar.closed_list], dtype=np.float64).reshape(-1, 4)
    out["open_nodes"] = np.array([(n.index, n.x, n.y, n.theta, n.f) for n in astar.open_list.queue], dtype=np.float64).reshape(-1, 5)
    if record_checks:
        out["checks"] = np.array(checks, dtype=np.float64).reshape(-1, 4)
    return out


def save(name, d):
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, name), **d)
    print("wrote", name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in d.items() if k in ("pops", "status", "final_path")}, flush=True)


# ----------------------------------------------------------------------------------------------
def gen_micro(groups=(1, 2, 3, 4)):
    cfg = config()
    veh = ref_costmap.Vehicle()
    rng = np.random.default_rng(20260927)
    # G1: costmaps of all 20 cases
    g1 = {}
    maps = {}
    for k in range(1, 21):
        m = load_map(os.path.join(CASES, f"Case{k}.csv"), cfg)
        maps[k] = m
        for kk, v in map_arrays(m).items():
            g1[f"c{k}_{kk}"] = v
        g1[f"c{k}_obs_n"] = np.array([len(o) for o in m.case.obs], dtype=np.int32)
        g1[f"c{k}_obs_xy"] = np.concatenate(m.case.obs, 0)
    g1["vehicle"] = np.array([veh.lw, veh.lf, veh.lr, veh.lb, veh.max_steering_angle, veh.max_v, veh.min_radius_turn])
    g1["steer_tan"] = np.tan(np.linspace(-veh.max_steering_angle, veh.max_steering_angle, cfg["steering_angle_num"]))
    if 1 in groups:
        save("g1_costmaps.npz", g1)
    if 2 in groups:
        gen_micro_g2(maps, rng)
    if 3 in groups:
        gen_micro_g3(maps, rng, cfg, veh)
    if 4 in groups:
        gen_micro_g4(np.random.default_rng(44), veh)


def gen_micro_g2(maps, rng):

    # G2: index maths on 3 cases
    g2 = {}
    for k in (1, 13, 19):
        m = maps[k]
        b = m.boundary
        xs = rng.uniform(b[0], b[1], 10000)
        ys = rng.uniform(b[2], b[3], 10000)
        ids = np.array([m.convert_position_to_index(x, y) for x, y in zip(xs, ys)], dtype=np.int64)
        dj = ref_h.Dijkstra(m)
        # is_obstacle also probes one pitch outside the bounds (negative-index wrap / clamp)
        xo = rng.uniform(b[0] - m._discrete_x, b[1] + m._discrete_x, 10000)
        yo = rng.uniform(b[2] - m._discrete_y, b[3] + m._discrete_y, 10000)
        ob = np.array([dj.is_obstacle(x, y) for x, y in zip(xo, yo)], dtype=np.uint8)
        g2.update({f"c{k}_x": xs, f"c{k}_y": ys, f"c{k}_id": ids, f"c{k}_xo": xo, f"c{k}_yo": yo, f"c{k}_obst": ob})
    save("g2_index.npz", g2)


def gen_micro_g3(maps, rng, cfg, veh, cases=(1, 4, 5, 13, 19, 20), name="g3_collision.npz", n_rand=3000):
    # G3: collision booleans
    g3 = {}
    for k in cases:
        m = maps[k]
        b = m.boundary
        n = n_rand if k != 19 else n_rand // 2
        poses = np.stack([rng.uniform(b[0] + 2, b[1] - 2, n), rng.uniform(b[2] + 2, b[3] - 2, n),
                          rng.uniform(-np.pi, np.pi, n)], 1)
        # exactly axis-aligned headings (k = +-inf -> NaN distances)
        ax = poses[:200].copy()
        ax[:, 2] = np.tile([0.0, np.pi / 2, -np.pi / 2, np.pi, -np.pi], 40)
        # poses placed so that an obstacle point is near/inside the footprint
        ix, iy = np.where(m.cost_map == 255)
        sel = rng.integers(0, len(ix), 400)
        near = np.stack([m.map_position[0][ix[sel]] + rng.uniform(-3, 3, 400),
                         m.map_position[1][iy[sel]] + rng.uniform(-3, 3, 400), rng.uniform(-np.pi, np.pi, 400)], 1)
        near_ax = near[:100].copy()
        near_ax[:, 2] = np.tile([0.0, np.pi / 2, -np.pi / 2, np.pi], 25)
        allp = np.concatenate([poses, ax, near, near_ax], 0)
        dc = ref_cc.distance_checker(map=m, vehicle=veh, config=cfg)
        tc = ref_cc.two_circle_checker(map=m, vehicle=veh, config=cfg)
        rd = np.zeros(len(allp), np.uint8)
        rc = np.zeros(len(allp), np.uint8)
        nn = np.zeros(len(allp), np.int32)
        for i, (x, y, t) in enumerate(allp):
            rd[i] = bool(dc.check(x, y, t))
            rc[i] = bool(tc.check(x, y, t))
            nn[i] = len(dc.get_near_obstacles(x, y, t)[0][0])
        g3.update({f"c{k}_poses": allp, f"c{k}_dist": rd, f"c{k}_circ": rc, f"c{k}_near": nn})
        print("G3 case", k, "collide frac", rd.mean(), rc.mean(), flush=True)
    # footprint corners for 500 poses (A4)
    pp = np.stack([rng.uniform(-30, 30, 500), rng.uniform(-30, 30, 500), rng.uniform(-np.pi, np.pi, 500)], 1)
    g3["corner_poses"] = pp
    g3["corners"] = np.array([veh.create_anticlockpoint(x, y, t, cfg).reshape(5, 2) for x, y, t in pp])
    save(name, g3)


def gen_g3_rest():
    """G3 for the 14 BenchmarkCases the first fixture does not hold (SURVEY 8c asks for all 20)."""
    cfg = config()
    veh = ref_costmap.Vehicle()
    rest = [k for k in range(1, 21) if k not in (1, 4, 5, 13, 19, 20)]
    maps = {k: load_map(os.path.join(CASES, f"Case{k}.csv"), cfg) for k in rest}
    gen_micro_g3(maps, np.random.default_rng(20260927 + 3), cfg, veh, cases=rest, name="g3_collision_rest.npz", n_rand=1200)


def gen_micro_g4(rng, veh):
    # G4: Reeds-Shepp
    maxc = 1 / veh.min_radius_turn
    n = 20000
    q0 = np.stack([rng.uniform(-25, 25, n), rng.uniform(-25, 25, n), rng.uniform(-np.pi, np.pi, n)], 1)
    q1 = np.stack([rng.uniform(-25, 25, n), rng.uniform(-25, 25, n), rng.uniform(-np.pi, np.pi, n)], 1)
    # structured: close poses (short paths, many CCC/CCCC winners)
    q1[n // 2:, :2] = q0[n // 2:, :2] + rng.uniform(-6, 6, (n - n // 2, 2))
    MAXP = 320
    NSAMP = 2500   # sampled way-points are stored for the first NSAMP queries only (fixture size)
    L = np.zeros(n)
    types = np.full((n, 5), -1, np.int8)
    lens = np.zeros((n, 5))
    npts = np.zeros(n, np.int32)
    pts = np.zeros((n, MAXP, 3))
    dirs = np.zeros((n, MAXP), np.int8)
    ncand = np.zeros(n, np.int32)
    cand_types = np.full((2000, 12, 5), -1, np.int8)
    cand_lens = np.zeros((2000, 12, 5))
    for i in range(n):
        p = ref_rs.calc_optimal_path(*q0[i], *q1[i], maxc)
        L[i] = p.L
        nl = len(p.lengths)
        types[i, :nl] = [TYPE_CODE[c] for c in p.ctypes]
        lens[i, :nl] = p.lengths
        npts[i] = len(p.x)
        assert npts[i] <= MAXP
        pts[i, :npts[i], 0] = p.x
        pts[i, :npts[i], 1] = p.y
        pts[i, :npts[i], 2] = p.yaw
        dirs[i, :npts[i]] = p.directions
        if i < 2000:
            c = ref_rs.generate_path(list(q0[i]), list(q1[i]), maxc)
            ncand[i] = len(c)
            for j, pc in enumerate(c):
                cand_types[i, j, :len(pc.ctypes)] = [TYPE_CODE[t] for t in pc.ctypes]
                cand_lens[i, j, :len(pc.lengths)] = pc.lengths
    mp = int(npts[:NSAMP].max())
    save("g4_rs.npz", dict(maxc=maxc, q0=q0, q1=q1, L=L, types=types, lens=lens, npts=npts, pts=pts[:NSAMP, :mp],
                           dirs=dirs[:NSAMP, :mp], ncand=ncand[:2000], cand_types=cand_types, cand_lens=cand_lens))
    # pi_2_pi / M
    th = rng.uniform(-20, 20, 5000)
    save("g4_angles.npz", dict(th=th, pi2pi=np.array([ref_rs.pi_2_pi(t) for t in th]), M=np.array([ref_rs.M(t) for t in th])))


def gen_hfield(cases=(1, 4), random_goals=None):
    """G5: heuristic fields. Query the far corner region so that the sweep closes (almost) the whole map.
    random_goals=(first, count): instead of the case's own goal, `count` random free goals (index first..) on
    each case, written as g5_hfield_c<k>_r<i>.npz (8 random goals on Case1)."""
    cfg = config()
    rng = np.random.default_rng(5)
    for k in cases:
        csv = os.path.join(CASES, f"Case{k}.csv")
        m = load_map(csv, cfg)
        goals = [(m.case.xf, m.case.yf)]
        tag = "g"
        if random_goals is not None:
            first, count = random_goals
            rr = np.random.default_rng(50 + k)
            allg = sampling.sample_free_poses(m.boundary, m.case.obs, 64, rr, margin=6.0)
            goals = [(allg[i][0], allg[i][1]) for i in range(first, first + count)]
            tag = "r"
            rng = np.random.default_rng(500 + 10 * k + first)
        elif k == 1:
            for _ in range(2):
                p = sampling.sample_free_poses(m.boundary, m.case.obs, 1, rng, margin=6.0)[0]
                goals.append((p[0], p[1]))
        for gi, (gx, gy) in enumerate(goals):
            if random_goals is not None:
                gi = random_goals[0] + gi
            m.case.xf, m.case.yf = float(gx), float(gy)
            dj = ref_h.Dijkstra(m)
            b = m.boundary
            # a resumable sequence of queries: a few interior points, then the 4 corners (inset)
            qs = [(rng.uniform(b[0] + 6, b[1] - 6), rng.uniform(b[2] + 6, b[3] - 6)) for _ in range(6)]
            qs += [(b[0] + 0.55, b[2] + 0.55), (b[1] - 0.55, b[3] - 0.55), (b[0] + 0.55, b[3] - 0.55), (b[1] - 0.55, b[2] + 0.55)]
            res = []
            t0 = time.time()
            for (x, y) in qs:
                signal.signal(signal.SIGALRM, _alarm)
                signal.alarm(1500)
                try:
                    d, cl = dj.compute_path(x, y)
                    res.append((x, y, d, len(cl)))
                except Timeout:
                    res.append((x, y, -1, len(dj.closedlist)))
                    break
                finally:
                    signal.alarm(0)
            save(f"g5_hfield_c{k}_{tag}{gi}.npz", dict(goal=np.array([gx, gy]), queries=np.array(res, dtype=np.float64),
                                                     closed_id=np.array([g.grid_id for g in dj.closedlist], dtype=np.int64),
                                                     closed_dist=np.array([g.distance for g in dj.closedlist], dtype=np.int64),
                                                     closed_x=np.array([g.grid_x for g in dj.closedlist]),
                                                     closed_y=np.array([g.grid_y for g in dj.closedlist]),
                                                     case=k))
            print("G5", k, gi, "cells", len(dj.closedlist), "t", time.time() - t0, flush=True)


def gen_trace(case_ids, timeout=5400):
    cfg = config()
    for k in case_ids:
        t0 = time.time()
        d = run_plan(os.path.join(CASES, f"Case{k}.csv"), cfg, timeout=timeout, record_checks=(k in (1, 20)))
        d["case"] = k
        save(f"g6_trace_case{k}.npz", d)
        print("trace case", k, d["status"], "pops", len(d["pops"]), "t", time.time() - t0, flush=True)


def gen_random(case_id, n, seed_off=0, timeout=1200):
    cfg = config()
    csv = os.path.join(CASES, f"Case{case_id}.csv")
    m0 = load_map(csv, cfg)
    veh = ref_costmap.Vehicle()
    dc = ref_cc.distance_checker(map=m0, vehicle=veh, config=cfg)
    rng = np.random.default_rng(20260927 + case_id + 1000 * seed_off)
    poses = sampling.sample_free_poses(m0.boundary, m0.case.obs, 2 * n, rng, margin=6.0, check=dc.check)
    for i in range(n):
        st, go = poses[2 * i], poses[2 * i + 1]
        if os.path.exists(os.path.join(GOLD, f"g7_random_case{case_id}_s{seed_off}_{i}.npz")):
            continue                      # the sampler is sequential: a longer run extends an earlier one
        t0 = time.time()
        d = run_plan(csv, cfg, start=st, goal=go, timeout=timeout)
        d["case"] = case_id
        d["start"] = st
        d["goal"] = go
        save(f"g7_random_case{case_id}_s{seed_off}_{i}.npz", d)
        print("random", case_id, i, d["status"], "pops", len(d["pops"]), "t", time.time() - t0, flush=True)


def gen_irregular():
    """G11: a goal whose goal-anchored lattice is NOT regular with respect to the id grid -- xf one ulp below a cell
    border (nextafter(b0 + 171 * dx, -inf) on the Case1 map), so the accumulated lattice positions xf +- k*dx
    (compute_h.py:58-66,89-186) step over grid-id columns: some id columns are never generated. The device refuses such
    goals (AVP_PLAN_LATTICE); this fixture records what the reference does there, for tests/test_gpu_limits.py: it never
    returns from PathPlanner.__init__ (the start's heuristic query is for an id the sweep never produces: the
    `while` of compute_h.py:77 spins on), recorded as status "timeout" with no pops after 300 s -- or whatever exception
    it ends with."""
    cfg = config()
    csv = os.path.join(CASES, "Case1.csv")
    m0 = load_map(csv, cfg)
    gx = float(np.nextafter(float(m0.boundary[0]) + 171 * float(m0._discrete_x), -np.inf))
    go = np.array([gx, m0.case.yf, m0.case.thetaf])
    st = np.array([m0.case.x0, m0.case.y0, m0.case.theta0])
    t0 = time.time()
    try:
        d = run_plan(csv, cfg, start=st, goal=go, timeout=300)
    except Exception as e:                                   # anything but Timeout / AttributeError
        signal.alarm(0)
        d = {"status": type(e).__name__, "pops": np.zeros((0, 11))}
        d.update({"map_" + k: v for k, v in map_arrays(m0).items()})
    d["case"] = 1
    d["start"] = st
    d["goal"] = go
    d["seconds"] = time.time() - t0
    save("g11_irregular_lattice_case1.npz", d)
    print("irregular", d["status"], "pops", len(d["pops"]), "t", time.time() - t0, flush=True)


VARIANTS = {
    "steer7_r6": {"steering_angle_num": 7, "flag_radius": 6.0},
    "steer3": {"steering_angle_num": 3},
    "dt08": {"dt": 0.8, "trajectory_dt": 0.2, "cost_gear": 3, "cost_heading_change": 1.5},
    "circle": {"collision_check": "circle"},
    "margins_rsall": {"flag_radius": 1e9, "safe_side_dis": 0.05, "safe_fr_dis": 0.2},
    # round 6: motion-primitive sets beyond rounds 1 - 5's device limits (16 steering angles, 4 sub-steps)
    "steer17": {"steering_angle_num": 17},                       # 34 children per expansion (hybrid_a_star.py:81-83,133)
    "dt10_ddt02": {"dt": 1.0, "trajectory_dt": 0.2},             # ceil(dt / trajectory_dt) = 5 sub-steps (:185)
}


def gen_variants(case_id=4, n_random=3, timeout=600, names=None):
    """G10: the reference under other config.yaml values (motion-primitive sets, time steps, costs, checker,
    margins, flag radius): the case's own start/goal plus n_random random pairs per variant. The overrides
    travel in the fixture as JSON (`cfg_json`)."""
    import json
    csv = os.path.join(CASES, f"Case{case_id}.csv")
    for name, over in VARIANTS.items():
        if names and name not in names:
            continue
        cfg = config()
        cfg.update(over)
        m0 = load_map(csv, cfg)
        veh = ref_costmap.Vehicle()
        dc = ref_cc.distance_checker(map=m0, vehicle=veh, config=cfg)
        rng = np.random.default_rng(20260927 + 77 * case_id)
        poses = sampling.sample_free_poses(m0.boundary, m0.case.obs, 2 * n_random, rng, margin=6.0, check=dc.check)
        probs = [(None, None)] + [(poses[2 * i], poses[2 * i + 1]) for i in range(n_random)]
        for i, (st, go) in enumerate(probs):
            t0 = time.time()
            d = run_plan(csv, cfg, start=st, goal=go, timeout=timeout)
            d["case"] = case_id
            d["cfg_json"] = json.dumps(over)
            if st is not None:
                d["start"] = st
                d["goal"] = go
            save(f"g10_variant_{name}_case{case_id}_{i}.npz", d)
            print("variant", name, case_id, i, d["status"], "pops", len(d["pops"]), "t", time.time() - t0, flush=True)


def gen_synth():
    """G8: small synthetic maps through the reference (C4-style polygons, C5-style parking row)."""
    cfg = config()
    os.makedirs("/tmp/avp_synth", exist_ok=True)
    veh = ref_costmap.Vehicle()
    # C4-like: 24 m x 24 m polygons; the CSV start/goal only fix the bounds (+-12 m around them)
    polys = sampling.synthetic_polygon_map(seed=4)
    csv4 = "/tmp/avp_synth/c4.csv"
    sampling.write_tpcap_csv(csv4, (12.0, 12.0, 0.0), (12.0, 12.0, 0.0), polys)
    m = load_map(csv4, cfg, discrete=0.12)
    d = {"c4_" + k: v for k, v in map_arrays(m).items()}
    d["c4_obs_n"] = np.array([len(o) for o in polys], dtype=np.int32)
    d["c4_obs_xy"] = np.concatenate(polys, 0)
    rng = np.random.default_rng(4)
    poses = sampling.sample_free_poses(m.boundary, polys, 4096, rng, margin=1.0, reject=False)
    dc = ref_cc.distance_checker(map=m, vehicle=veh, config=cfg)
    tc = ref_cc.two_circle_checker(map=m, vehicle=veh, config=cfg)
    d["c4_poses"] = poses
    d["c4_dist"] = np.array([bool(dc.check(*p)) for p in poses], dtype=np.uint8)
    d["c4_circ"] = np.array([bool(tc.check(*p)) for p in poses], dtype=np.uint8)
    save("g8_synth_c4.npz", d)
    # two plans on the C4 map
    free = sampling.sample_free_poses(m.boundary, polys, 8, np.random.default_rng(44), margin=3.0, check=dc.check)
    for i in range(3):
        r = run_plan(csv4, cfg, start=free[2 * i], goal=free[2 * i + 1], discrete=0.12, timeout=1500)
        r["start"], r["goal"] = free[2 * i], free[2 * i + 1]
        save(f"g8_synth_c4_plan{i}.npz", r)
    # C5-like: short parking row (12 bays per row), flag_radius huge so the RS shot runs at every pop
    obs, goal, aisle = sampling.parking_lot_map(n_per_row=12, empty_bay=6)
    csv5 = "/tmp/avp_synth/c5.csv"
    sampling.write_tpcap_csv(csv5, (15.0, 7.4, 0.0), goal, obs)
    cfg5 = dict(cfg)
    cfg5["flag_radius"] = 1e9
    for i, st in enumerate([(6.0, 7.4, 0.0), (24.0, 7.6, np.pi)]):
        r = run_plan(csv5, cfg5, start=st, goal=goal, timeout=2400)
        r["start"], r["goal"] = np.array(st), np.array(goal)
        r["obs_n"] = np.array([len(o) for o in obs], dtype=np.int32)
        r["obs_xy"] = np.concatenate(obs, 0)
        save(f"g8_synth_c5_plan{i}.npz", r)


def gen_corridor(cases=(1, 4, 5, 9, 13)):
    """G9: corridor bounds compute_collision_H (optimization/path_optimazition.py:221-409) on the gear
    segments of finished golden plans. cvxopt is only needed by the QP solve, not by this method: stub it."""
    stub = types.ModuleType("cvxopt")
    stub.matrix = lambda *a, **k: None
    stub.solvers = types.SimpleNamespace(options={}, qp=None)
    sys.modules.setdefault("cvxopt", stub)
    from optimization import path_optimazition as ref_po
    cfg = config()
    veh = ref_costmap.Vehicle()
    out = {"expand_dis": float(cfg["expand_dis"])}
    for k in cases:
        g = np.load(os.path.join(GOLD, f"g6_trace_case{k}.npz"))
        m = load_map(os.path.join(CASES, f"Case{k}.csv"), cfg)
        po = ref_po.path_opti(m, veh, cfg)
        seg_len = g["split_len"]
        pts = g["split_concat"]
        # extra poses: exactly axis-aligned headings and all four heading quadrants on the same positions
        extra = pts[: min(40, len(pts))].copy()
        extra[:, 2] = np.resize([0.0, np.pi / 2, -np.pi / 2, np.pi, -np.pi, 2.5, -2.5, -0.7, 0.7, 1.9], len(extra))
        allp = np.concatenate([pts, extra], 0)
        po.original_path = [list(map(float, q)) for q in allp]
        H, Hs = po.compute_collision_H()
        n = len(allp)
        H = np.asarray(H).reshape(-1)
        out[f"c{k}_poses"] = allp
        out[f"c{k}_Hmax"] = H[: 2 * n].reshape(n, 2)
        out[f"c{k}_Hmin"] = -H[2 * n:].reshape(n, 2)
        out[f"c{k}_slack_shape"] = np.array(np.asarray(Hs).shape)
        out[f"c{k}_slack"] = np.asarray(Hs).reshape(-1)
        print("G9 case", k, n, "points", flush=True)
    save("g9_corridor.npz", out)


if __name__ == "__main__":
    what = sys.argv[1]
    if what == "micro":
        gen_micro(tuple(int(a) for a in sys.argv[2:]) or (1, 2, 3, 4))
    elif what == "hfield":
        gen_hfield(tuple(int(a) for a in sys.argv[2:]) or (1, 4))
    elif what == "trace":
        gen_trace([int(a) for a in sys.argv[2:]])
    elif what == "variants":           # variants <case> [name ...]
        gen_variants(int(sys.argv[2]) if len(sys.argv) > 2 else 4, names=sys.argv[3:] or None)
    elif what == "random":
        gen_random(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0,
                   timeout=int(sys.argv[5]) if len(sys.argv) > 5 else 1200)
    elif what == "hfield_random":          # hfield_random <case> <first> <count>
        gen_hfield((int(sys.argv[2]),), random_goals=(int(sys.argv[3]), int(sys.argv[4])))
    elif what == "g3rest":
        gen_g3_rest()
    elif what == "synth":
        gen_synth()
    elif what == "corridor":
        gen_corridor()
    elif what == "irregular":
        gen_irregular()
