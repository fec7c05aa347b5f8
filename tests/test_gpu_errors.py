"""Error behaviour of the C-ABI and of the reference-style wrappers on a real device."""
import ctypes as C

import numpy as np
import pytest

from conftest import case_map_from_gold

pytestmark = pytest.mark.gpu


def test_bad_arguments_are_status_codes(vehicle, cfg):
    from automatedvaletparking_amd import _native
    L = _native.lib()
    dm = _native.DeviceMap(case_map_from_gold(1), vehicle, cfg)
    assert L.avp_check_batch(dm.h, C.c_int32(7), C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), C.c_int64(1), C.c_void_p(8), 0) == -1
    assert "kind" in _native.last_error()
    assert L.avp_plan_batch(dm.h, None, None, C.c_int64(3), 1, 64, None, C.c_int64(0), None, None, 0, None, 0) == -1
    ws = dm.empty(1024, dm.torch.uint8)
    st = dm.dev_tensor(np.zeros((1, 3)))
    res = dm.empty(512, dm.torch.uint8)
    rc = L.avp_plan_batch(dm.h, C.c_void_p(st.data_ptr()), C.c_void_p(st.data_ptr()), C.c_int64(1), C.c_int32(1), C.c_int32(4096),
                          C.c_void_p(ws.data_ptr()), C.c_int64(1024), C.c_void_p(res.data_ptr()), None, 0, None, 0)
    assert rc == -4 and "workspace" in _native.last_error()
    # obstacle list that is not the np.where order
    pk = dict(dm.pack)
    bad = pk["obs_ix"][::-1].copy()
    h = C.c_void_p()
    bnd = np.ascontiguousarray(pk["boundary"], dtype=np.float64)
    rc = L.avp_map_create(C.byref(dm.params), pk["occ"].ctypes.data_as(C.c_void_p), C.c_int32(pk["nx"]), C.c_int32(pk["ny"]),
                          pk["xs"].ctypes.data_as(C.c_void_p), pk["ys"].ctypes.data_as(C.c_void_p), bnd.ctypes.data_as(C.c_void_p),
                          bad.ctypes.data_as(C.c_void_p), pk["obs_iy"].ctypes.data_as(C.c_void_p), C.c_int32(len(bad)), C.c_int32(0), C.byref(h))
    assert rc == -1


def test_problem_statuses(vehicle, cfg):
    from automatedvaletparking_amd import _native, path_planner
    m = case_map_from_gold(1)
    c = m.case
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=5)
    bp = path_planner.BatchPlanner(dm, max_nodes=4096)
    starts = np.array([[c.x0, c.y0, c.theta0], [c.x0, c.y0, c.theta0], [c.x0, c.y0, c.theta0], [c.xf, c.yf, c.thetaf]])
    goals = np.array([[c.xf, c.yf, c.thetaf], [m.boundary[1] + 5.0, c.yf, 0.0], [c.xf, c.yf, c.thetaf], [c.xf, c.yf, c.thetaf]])
    r = bp.plan(starts, goals)
    assert r[0].status == 4 and r[0].n_pops == 5            # ITER_LIMIT (Case1 needs 85 pops)
    assert r[1].status == 6                                  # goal outside the map
    assert r[3].status == 3                                  # start == goal: the reference's assertion L >= 0.01
    tiny = path_planner.BatchPlanner(_native.DeviceMap(m, vehicle, cfg), max_nodes=64, n_slots=1)
    assert tiny.plan(starts[:1], goals[:1])[0].status == 5   # node arena exhausted
    # wrapper-level mapping
    pl = path_planner.PathPlanner(config=cfg, map=m, vehicle=vehicle)
    pl._batch = tiny
    # CAPACITY is not an error of the reference (it has no limits): a_star_plan repeats the search with a doubled
    # arena / path buffer until it fits, and returns the reference's Case1 path (85 pops, 30 way-points)
    final_path, astar_path, rs_path = pl.a_star_plan()
    assert len(final_path) == 30 and rs_path.ctypes == ["L", "R", "L", "R"]
    # ... while statuses that have no reference outcome still raise
    pl2 = path_planner.PathPlanner(config=cfg, map=m, vehicle=vehicle)
    pl2._batch = path_planner.BatchPlanner(_native.DeviceMap(m, vehicle, cfg, max_pops=5), n_slots=1)
    with pytest.raises(RuntimeError):
        pl2.a_star_plan()
