"""The limits the device planner has and the reference does not (include/avp.h): every one is a status code or a raised
error, never a wrong answer, and each is shown here -- together with what still works on the same input and, for the one
DATA-dependent limit, what the reference does (golden G11). The limits of rounds 1 - 5 that are gone are parity tests now:
map_discrete_size below 0.088 m (a footprint wider than 64 map columns), maps of more than 4 095 nodes per axis, and (round 6)
more than 16 steering angles / 4 sub-steps per motion primitive."""
import os

import numpy as np
import pytest

from conftest import CASES, case_map_from_gold, gold

pytestmark = pytest.mark.gpu


def _plan_one(m, vehicle, cfg, **bp_kw):
    from automatedvaletparking_amd import _native, path_planner
    c = m.case
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=50)
    return path_planner.BatchPlanner(dm, max_nodes=4096, **bp_kw).plan([[c.x0, c.y0, c.theta0]], [[c.xf, c.yf, c.thetaf]])


@pytest.mark.parametrize("over", [{"steering_angle_num": 17},                      # 34 children (hybrid_a_star.py:81-83,133 takes any number)
                                  {"dt": 1.0, "trajectory_dt": 0.2},                  # 5 sub-steps (hybrid_a_star.py:185)
                                  {"steering_angle_num": 32},                         # 64 children: every lane of the resolving wave
                                  {"steering_angle_num": 9, "dt": 2.0, "trajectory_dt": 0.1},   # 18 children x 20 sub-steps = 360 poses per expansion
                                  {"steering_angle_num": 6, "dt": 1.0, "trajectory_dt": 0.2},   # 12 children x 5 = 60 poses: what a group of waves still holds itself (64)
                                  {"dt": 1.4, "trajectory_dt": 0.2}])                           # 10 children x 7 = 70 poses: a group hands it to the workgroup form
def test_motion_primitive_sets_beyond_the_old_limits_plan_like_the_reference(over, vehicle, cfg):
    """Until round 5 more than 16 steering angles or 4 sub-steps was an error. Now the workgroup form holds 64 children (32
    steering angles) and any number of sub-steps up to 512 poses per expansion; the group forms hold 16 children / 64 sub-step
    poses and hand anything larger to the workgroup form inside the same call -- so EVERY kernel form plans such a configuration,
    with the same result. Case1's own problem and 12 random pairs against the pinned oracle, every observable field (the
    reference itself on two of these sets: goldens g10_variant_steer17 / g10_variant_dt10_ddt02, tests/test_gpu_plan.py)."""
    import _parity
    from automatedvaletparking_amd import _native, path_planner
    from oracle import oracle
    m = case_map_from_gold(1)
    c2 = dict(cfg)
    c2.update(over)
    cap = 40
    dm = _native.DeviceMap(m, vehicle, c2, max_pops=cap)
    st, go = _pairs(m, dm, 12, 5)
    st = np.concatenate([st, [[m.case.x0, m.case.y0, m.case.theta0]]])
    go = np.concatenate([go, [[m.case.xf, m.case.yf, m.case.thetaf]]])
    o = oracle.Oracle(m, vehicle, c2, max_pops=cap)
    for mode in (1, 2, 3, 4, path_planner.STAGED):
        res = path_planner.BatchPlanner(dm, max_nodes=8192, mode=mode, n_slots=64 if mode in (2, 3, 4) else None).plan(st, go, max_trace=cap)
        bad, _ = _parity.compare_pinned(o, res, st, go, cap)
        assert not bad, (over, mode, bad[:3])


def test_motion_primitive_limits_that_are_left(vehicle, cfg):
    """What is still refused (the reference has no limit): more than 32 steering angles (a child is a lane of the resolving
    wave) and more than 512 sub-step poses per expansion -- ValueError where the parameter block is built, AVP_ERR_ARG from the
    C-ABI for a block edited behind it."""
    import ctypes as C
    from automatedvaletparking_amd import _native
    for over in ({"steering_angle_num": 33}, {"steering_angle_num": 10, "dt": 2.6, "trajectory_dt": 0.1}):
        c2 = dict(cfg)
        c2.update(over)
        with pytest.raises(ValueError, match="steering_angle_num"):
            _plan_one(case_map_from_gold(1), vehicle, c2)
    m = case_map_from_gold(1)
    dm = _native.DeviceMap(m, vehicle, cfg)
    pk = dict(dm.pack)
    for field, val in (("n_steer", 33), ("n_sub", 60)):
        p2 = _native.make_params(cfg, vehicle, 50)
        setattr(p2, field, val)
        h = C.c_void_p()
        bnd = np.ascontiguousarray(pk["boundary"], dtype=np.float64)
        L = _native.lib()
        rc = L.avp_map_create(C.byref(p2), pk["occ"].ctypes.data_as(C.c_void_p), C.c_int32(pk["nx"]), C.c_int32(pk["ny"]),
                              pk["xs"].ctypes.data_as(C.c_void_p), pk["ys"].ctypes.data_as(C.c_void_p), bnd.ctypes.data_as(C.c_void_p),
                              pk["obs_ix"].ctypes.data_as(C.c_void_p), pk["obs_iy"].ctypes.data_as(C.c_void_p), C.c_int32(len(pk["obs_ix"])), C.c_int32(0), C.byref(h))
        if rc == 0:
            st = dm.dev_tensor(np.array([[m.case.x0, m.case.y0, m.case.theta0]]))
            ws = dm.empty(64 << 20, dm.torch.uint8)
            res = dm.empty(4096, dm.torch.uint8)
            rc2 = L.avp_plan_batch(h, C.c_void_p(st.data_ptr()), C.c_void_p(st.data_ptr()), C.c_int64(1), C.c_int32(1), C.c_int32(4096),
                                   C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()), C.c_void_p(res.data_ptr()), None, 0, None, 0)
            assert rc2 == -1 and "motion primitives" in _native.last_error()
            L.avp_map_destroy(h)
        else:
            assert rc == -1                                   # (the map constructor may already refuse the block)


def _pairs(m, dm, n, seed):
    from automatedvaletparking_amd import workloads
    return workloads.sample_pairs(m, dm.check_batch, n, np.random.default_rng(seed), chunk=8 * n)


def test_fine_map_footprint_wider_than_64_columns_plans_like_the_reference(vehicle, cfg):
    """map_discrete_size 0.05 m (config/config.yaml:6 takes any value; the default is 0.1): the inflated rectangle's
    diagonal spans 107 cells, so a footprint's AABB covers up to 107 map columns and 2 - 3 bitmap words of rows. Until
    round 4 avp_plan_batch refused such a map ("footprint diagonal < 61 cells": the planner's collision pass walked
    <= 64 columns per pose); the pass now walks any number of column chunks and words (pl_check_pass, the general walk).
    Every kernel form against the pinned oracle: Case1's own problem run to the end and 48 random pairs, every observable
    field; both checkers on 3 000 random poses."""
    import _parity
    from automatedvaletparking_amd import costmap, _native, path_planner
    from oracle import oracle
    m = costmap.Map(file=os.path.join(CASES, "Case1.csv"), discrete_size=0.05)
    assert m.cost_map.shape[0] > 500
    cap = 200
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=cap)
    o = oracle.Oracle(m, vehicle, cfg, max_pops=cap)
    rng = np.random.default_rng(5)
    b = m.boundary
    poses = np.stack([rng.uniform(b[0] + 3, b[1] - 3, 3000), rng.uniform(b[2] + 3, b[3] - 3, 3000), rng.uniform(-np.pi, np.pi, 3000)], 1)
    for kind in (0, 1):
        want = o.check_batch(poses, kind=kind)
        assert np.array_equal(dm.check_batch(poses, kind=kind), want) and 0 < want.sum() < len(want)
    c = m.case
    st, go = _pairs(m, dm, 48, 11)
    st = np.concatenate([[[c.x0, c.y0, c.theta0]], st])
    go = np.concatenate([[[c.xf, c.yf, c.thetaf]], go])
    ref = None
    for mode in (1, 2, 3, 4):
        res = path_planner.BatchPlanner(dm, max_nodes=8192, mode=mode, n_slots=64 if mode > 1 else None).plan(st, go, max_trace=cap)
        bad, _ = _parity.compare_pinned(o, res, st, go, cap)
        assert not bad, (mode, len(bad), bad[:6])
        sig = [(r.status, r.n_pops, r.counters["n_checks"]) for r in res]
        assert ref is None or sig == ref, mode
        ref = sig
    assert sum(r.status == 0 for r in res) >= 10 and max(r.n_pops for r in res) >= 50
    # the two-circle checker inside the planner on the same map
    c2 = dict(cfg)
    c2["collision_check"] = "circle"
    dm2 = _native.DeviceMap(m, vehicle, c2, max_pops=cap)
    res = path_planner.BatchPlanner(dm2, max_nodes=8192).plan(st[:16], go[:16], max_trace=cap)
    bad, _ = _parity.compare_pinned(oracle.Oracle(m, vehicle, c2, max_pops=cap), res, st[:16], go[:16], cap)
    assert not bad, bad[:6]


def test_map_wider_than_4095_nodes_plans_like_the_reference(vehicle, cfg, tmp_path):
    """A 430 m x 30 m strip at the default 0.1 m: 4 300 x 300 nodes. Until round 4 the planner's collision queue packed cell
    indices in 12 bits (nx, ny <= 4095); it now packs 13 (8 191, the limit of avp_map_create and of the check kernel). The
    strip's own problem (405 m: capped) and 32 random pairs against the pinned oracle in the workgroup and the wave form."""
    import _parity
    from automatedvaletparking_amd import costmap, sampling, _native, path_planner
    from oracle import oracle
    rng = np.random.default_rng(2)
    polys = [np.array([[x, y], [x + 2.0, y], [x + 2.0, y + 1.5], [x, y + 1.5]]) for x, y in zip(rng.uniform(15, 422, 70), rng.uniform(6, 22, 70))]
    csv = tmp_path / "strip.csv"
    sampling.write_tpcap_csv(str(csv), (12.5, 14.0, 0.0), (417.5, 14.0, 0.0), polys)
    m = costmap.Map(file=str(csv), discrete_size=cfg["map_discrete_size"])
    assert m.cost_map.shape[0] > 4095
    cap = 120
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=cap)
    o = oracle.Oracle(m, vehicle, cfg, max_pops=cap)
    b = m.boundary
    # poses in the last 44 m of the strip: starts left of x = 400 m, goals right of x = 411 m -- beyond column 4 095, where a
    # 12-bit index would have wrapped
    n = 6000
    poses = np.stack([rng.uniform(b[0] + 385, b[1] - 3, n), rng.uniform(b[2] + 3, b[3] - 3, n), rng.uniform(-np.pi, np.pi, n)], 1)
    want = o.check_batch(poses, kind=0)
    assert np.array_equal(dm.check_batch(poses, kind=0), want) and 0 < want.sum() < len(want)
    free = np.array([p for p, h in zip(poses, want) if not h and sampling.pose_is_free(p[0], p[1], p[2], m.case.obs)])
    far, near = free[free[:, 0] > b[0] + 411.0], free[free[:, 0] < b[0] + 400.0]
    assert len(far) >= 32 and len(near) >= 32
    c = m.case
    st = np.concatenate([[[c.x0, c.y0, c.theta0]], near[:32]])
    go = np.concatenate([[[c.xf, c.yf, c.thetaf]], far[:32]])
    for mode in (1, 2):
        res = path_planner.BatchPlanner(dm, max_nodes=4096, mode=mode, n_slots=64 if mode > 1 else None).plan(st, go, max_trace=cap)
        bad, _ = _parity.compare_pinned(o, res, st, go, cap)
        assert not bad, (mode, len(bad), bad[:6])
    assert sum(r.status == 0 for r in res) >= 10 and sum(r.status == 4 for r in res) >= 2 and max(r.n_pops for r in res if r.status == 0) >= 50


def test_map_wider_than_8191_nodes_plans_like_the_reference(vehicle, cfg, tmp_path):
    """An 860 m x 30 m strip at the default 0.1 m: 8 600 x 300 nodes. Rounds 1 - 5 refused more than 8 191 nodes per axis (the compacting
    kernels pack cell indices in 13 bits); since round 6 such a map takes the lane-per-pose forms -- the footprint kernel's all-points
    variant, the corridor kernel's lane-per-way-point variant, the planner's lane-per-pose collision pass -- with the same results.
    Poses beyond column 8 191 (x > 819 m): both checkers, corridor bounds, and 24 random pairs + the strip's own problem in every
    kernel form, against the pinned oracle."""
    import _parity
    from automatedvaletparking_amd import costmap, sampling, _native, path_planner
    from oracle import oracle
    rng = np.random.default_rng(3)
    polys = [np.array([[x, y], [x + 2.0, y], [x + 2.0, y + 1.5], [x, y + 1.5]]) for x, y in zip(rng.uniform(15, 852, 140), rng.uniform(6, 22, 140))]
    csv = tmp_path / "strip2.csv"
    sampling.write_tpcap_csv(str(csv), (12.5, 14.0, 0.0), (850.5, 14.0, 0.0), polys)
    m = costmap.Map(file=str(csv), discrete_size=cfg["map_discrete_size"])
    assert m.cost_map.shape[0] > 8191
    cap = 100
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=cap)
    o = oracle.Oracle(m, vehicle, cfg, max_pops=cap)
    b = m.boundary
    n = 6000
    poses = np.stack([rng.uniform(b[0] + 805, b[1] - 3, n), rng.uniform(b[2] + 3, b[3] - 3, n), rng.uniform(-np.pi, np.pi, n)], 1)
    for kind in (0, 1):
        want = o.check_batch(poses, kind=kind)
        assert np.array_equal(dm.check_batch(poses, kind=kind), want) and 0 < want.sum() < len(want), kind
    assert np.array_equal(dm.corridor_batch(poses[:2000], 0.8), o.corridor_batch(poses[:2000], 0.8), equal_nan=True)
    want = o.check_batch(poses, kind=0)
    free = np.array([p for p, h in zip(poses, want) if not h and sampling.pose_is_free(p[0], p[1], p[2], m.case.obs)])
    far, near = free[free[:, 0] > b[0] + 840.0], free[free[:, 0] < b[0] + 832.0]
    assert len(far) >= 24 and len(near) >= 24 and (near[:, 0] > b[0] + 819.2).sum() >= 8
    c = m.case
    st = np.concatenate([[[c.x0, c.y0, c.theta0]], near[:24]])
    go = np.concatenate([[[c.xf, c.yf, c.thetaf]], far[:24]])
    for mode in (1, 2, 3, 4):
        res = path_planner.BatchPlanner(dm, max_nodes=4096, mode=mode, n_slots=64 if mode > 1 else None).plan(st, go, max_trace=cap)
        bad, _ = _parity.compare_pinned(o, res, st, go, cap)
        assert not bad, (mode, len(bad), bad[:6])
    assert sum(r.status == 0 for r in res) >= 6
    # what is still refused: more than AVP_MAX_NODES_PER_AXIS (32 767) nodes on an axis
    import ctypes as C
    pk = dict(dm.pack)
    h = C.c_void_p()
    bnd = np.ascontiguousarray(pk["boundary"], dtype=np.float64)
    rc = _native.lib().avp_map_create(C.byref(dm.params), pk["occ"].ctypes.data_as(C.c_void_p), C.c_int32(32768), C.c_int32(2), pk["xs"].ctypes.data_as(C.c_void_p),
                                      pk["ys"].ctypes.data_as(C.c_void_p), bnd.ctypes.data_as(C.c_void_p), None, None, C.c_int32(0), C.c_int32(0), C.byref(h))
    assert rc == -1


def test_goal_on_a_cell_border_irregular_lattice(vehicle, cfg):
    """The data-dependent limit. compute_h.py:58-66,89-186 accumulates the lattice positions xf +- k * dx in floating point;
    when xf sits one ulp below a cell border (here nextafter(b0 + 171 * dx, -inf) on the Case1 map) the accumulated
    positions fall on either side of the later borders, and every other grid-id column to the right of the goal is never
    generated (65 of 289). The device's sweep keys everything by (column, row) offsets from the goal cell and refuses such
    a goal with AVP_PLAN_LATTICE (6) before the first pop. What the reference does there is recorded in golden G11: it
    never returns -- PathPlanner.__init__ asks for the start's heuristic distance, the start's id is in a column the sweep
    never produces, and compute_h.py:77 blocks on its empty queue (300 s, no pop, < 1 s of CPU). The oracle reports the same
    situation as H_UNREACHABLE (2) for 60 of 60 random starts. So the refusal loses no plan the reference would deliver;
    random goals never are such goals (0 of 20 000 on this map; tests/test_gpu_configs.py asserts no LATTICE status on
    2 560 random problems), and the same goal one ulp higher is an ordinary goal."""
    from automatedvaletparking_amd import _native, path_planner
    from oracle import oracle
    g = gold("g11_irregular_lattice_case1.npz")
    m = case_map_from_gold(1)
    assert str(g["status"]) == "timeout" and len(g["pops"]) == 0 and float(g["seconds"]) >= 300.0
    pk = m.pack()
    assert float(g["goal"][0]) == float(np.nextafter(float(m.boundary[0]) + 171 * pk["dx"], -np.inf))
    w = oracle.Oracle(m, vehicle, cfg, max_pops=100).plan(g["start"], g["goal"], max_trace=1)
    assert w["status"] == 2 and w["n_pops"] == 0
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=1000)
    for mode in (1, 2, 3, 4):
        r = path_planner.BatchPlanner(dm, max_nodes=8192, mode=mode, n_slots=16).plan([g["start"]] * 3, [g["goal"]] * 3)
        assert [x.status for x in r] == [6, 6, 6] and all(x.n_pops == 0 and len(x.final_path) == 0 for x in r)
    # one ulp higher -- exactly on the border -- the lattice is regular and the search is an ordinary one, equal to the oracle's
    import _parity
    go = np.array(g["goal"])
    go[0] = float(m.boundary[0]) + 171 * pk["dx"]
    res = path_planner.BatchPlanner(dm, max_nodes=8192).plan([g["start"]], [go], max_trace=1000)
    bad, _ = _parity.compare_pinned(oracle.Oracle(m, vehicle, cfg, max_pops=1000), res, [np.array(g["start"])], [go], 1000, threads=1)
    assert res[0].status == 0 and res[0].n_pops == 34 and not bad, bad
