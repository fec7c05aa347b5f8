"""The limits the device planner has and the reference does not (include/avp.h): every one is a status code or a raised
error, never a wrong answer, and each is shown here -- together with what still works on the same input (the footprint
kernels have wider limits and fall back to the all-points kernel beyond them) and, for the one DATA-dependent limit, what
the reference does (golden G11)."""
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


@pytest.mark.parametrize("over, what", [({"steering_angle_num": 17}, "steering_angle_num"),          # 34 children > 32 (hybrid_a_star.py:81-83 takes any)
                                        ({"dt": 1.0, "trajectory_dt": 0.2}, "trajectory_dt")])        # 5 sub-steps > 4 (hybrid_a_star.py:185)
def test_too_many_motion_primitives_is_an_error(over, what, vehicle, cfg):
    """Refused where the parameter block is built (_native.make_params: ValueError); a C caller that fills avp_params
    itself gets AVP_ERR_ARG "too many motion primitives" from avp_plan_batch (csrc/avp_capi_plan.inc)."""
    import ctypes as C
    from automatedvaletparking_amd import _native
    c2 = dict(cfg)
    c2.update(over)
    with pytest.raises(ValueError, match=what):
        _plan_one(case_map_from_gold(1), vehicle, c2)
    # the C-ABI guard behind it: a parameter block edited after the Python check
    m = case_map_from_gold(1)
    dm = _native.DeviceMap(m, vehicle, cfg)
    pk = dict(dm.pack)
    p2 = _native.make_params(cfg, vehicle, 50)
    if "steering_angle_num" in over:
        p2.n_steer = 17
    else:
        p2.n_sub = 5
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
    # the largest sets that are accepted: 16 steering angles (32 children: the workgroup form; the group forms hold 16
    # children and hand such a configuration over), 4 sub-steps
    ok = dict(cfg)
    ok.update({"steering_angle_num": 16} if "steering_angle_num" in over else {"dt": 0.8, "trajectory_dt": 0.2})
    assert _plan_one(case_map_from_gold(1), vehicle, ok)[0].status in (0, 4)


def test_footprint_wider_than_the_collision_pass_is_an_error_but_checks_still_work(vehicle, cfg, tmp_path):
    """discrete_size 0.05 m: the inflated rectangle's diagonal spans 107 cells (limit: < 61, the planner's collision pass
    walks <= 64 map columns per pose). avp_plan_batch refuses; check_batch answers through the all-points kernel."""
    from automatedvaletparking_amd import costmap, _native
    from oracle import oracle
    m = costmap.Map(file=os.path.join(CASES, "Case1.csv"), discrete_size=0.05)
    assert m.cost_map.shape[0] > 500
    with pytest.raises(RuntimeError, match="footprint diagonal"):
        _plan_one(m, vehicle, cfg)
    dm = _native.DeviceMap(m, vehicle, cfg)
    rng = np.random.default_rng(5)
    b = m.boundary
    poses = np.stack([rng.uniform(b[0] + 3, b[1] - 3, 3000), rng.uniform(b[2] + 3, b[3] - 3, 3000), rng.uniform(-np.pi, np.pi, 3000)], 1)
    want = oracle.Oracle(m, vehicle, cfg).check_batch(poses, kind=0)
    assert np.array_equal(dm.check_batch(poses, kind=0), want) and 0 < want.sum() < len(want)


def test_map_wider_than_4095_nodes_is_an_error_but_checks_still_work(vehicle, cfg, tmp_path):
    """A 430 m x 30 m strip at the default 0.1 m: 4 300 x 300 nodes (limit of the planner: 4 095 per axis, its collision
    queue packs cell indices in 12 bits; the check kernel's limit is 8 191)."""
    from automatedvaletparking_amd import costmap, sampling, _native
    from oracle import oracle
    rng = np.random.default_rng(2)
    polys = [np.array([[x, y], [x + 2.0, y], [x + 2.0, y + 1.5], [x, y + 1.5]]) for x, y in zip(rng.uniform(15, 400, 60), rng.uniform(12, 16, 60))]
    csv = tmp_path / "strip.csv"
    sampling.write_tpcap_csv(str(csv), (12.5, 14.0, 0.0), (417.5, 14.0, 0.0), polys)
    m = costmap.Map(file=str(csv), discrete_size=cfg["map_discrete_size"])
    assert m.cost_map.shape[0] > 4095
    with pytest.raises(RuntimeError, match="nx, ny <= 4095"):
        _plan_one(m, vehicle, cfg)
    dm = _native.DeviceMap(m, vehicle, cfg)
    b = m.boundary
    poses = np.stack([rng.uniform(b[0] + 3, b[1] - 3, 3000), rng.uniform(b[2] + 3, b[3] - 3, 3000), rng.uniform(-np.pi, np.pi, 3000)], 1)
    want = oracle.Oracle(m, vehicle, cfg).check_batch(poses, kind=0)
    assert np.array_equal(dm.check_batch(poses, kind=0), want) and 0 < want.sum() < len(want)


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
