"""plan_wave_kernel in both its forms -- one wave per problem (avp_plan_batch_mode mode 2) and a pair of waves per problem
(mode 3), four waves per problem (mode 4) -- against the CPU oracle in
device arithmetic and against plan_kernel (mode 1): every record field, pop trace, counter and way-point identical --
the result never depends on the kernel form that produced it, including the problems the wave form hands back to the
workgroup form (Reeds-Shepp shots longer than its sample buffer)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLD, case_map_from_gold
from test_gpu_plan import _assert_same_as_oracle, _gold_problem

pytestmark = pytest.mark.gpu


def _same_results(a, b):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert (x.status, x.n_pops) == (y.status, y.n_pops), (i, x.status_name, y.status_name, x.n_pops, y.n_pops)
        for k in ("n_checks", "n_rs", "n_closed", "n_open", "h_cells", "h_misses", "global_index", "n_nodes", "in_radius_last", "rs_collision"):
            assert x.counters[k] == y.counters[k], (i, k, x.counters[k], y.counters[k])
        assert np.array_equal(x.final_path, y.final_path) and np.array_equal(x.astar_path, y.astar_path)
        assert np.array_equal(x.rs_xyyaw, y.rs_xyyaw) and np.array_equal(x.rs_dirs, y.rs_dirs)
        assert x.rs_types == y.rs_types and x.rs_lengths == y.rs_lengths and x.rs_L == y.rs_L
        if x.trace is not None:
            assert np.array_equal(x.trace, y.trace, equal_nan=True)


FORMS = [2, 3, 4]       # one wave / a pair of waves / four waves per problem


@pytest.mark.parametrize("form", FORMS)
def test_wave_form_case1_batch_vs_oracle_and_workgroup_form(form, vehicle, cfg):
    from automatedvaletparking_amd import sampling, _native, path_planner
    from oracle import oracle
    m = case_map_from_gold(1)
    cap = 400
    o = oracle.Oracle(m, vehicle, cfg, max_pops=cap)
    rng = np.random.default_rng(20260927)
    poses = sampling.sample_free_poses(m.boundary, m.case.obs, 512, rng, margin=6.0,
                                       check=lambda x, y, t: bool(o.check_batch(np.array([[x, y, t]]))[0]))
    starts = np.concatenate([poses[0::2], poses[0::2]])
    goals = np.concatenate([poses[1::2], np.roll(poses[1::2], 7, axis=0)])
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=cap)
    wave = path_planner.BatchPlanner(dm, max_nodes=8192, mode=form).plan(starts, goals, max_trace=cap)
    wg = path_planner.BatchPlanner(dm, max_nodes=8192, mode=1).plan(starts, goals, max_trace=cap)
    _same_results(wave, wg)
    assert len({r.counters["n_nodes"] for r in wave}) > 20 and sum(r.status == 0 for r in wave) > 300
    from concurrent.futures import ThreadPoolExecutor
    with oracle.device_arithmetic():
        with ThreadPoolExecutor(min(64, os.cpu_count() or 1)) as ex:
            ws = list(ex.map(lambda i: o.plan(starts[i], goals[i], max_trace=cap), range(0, 512, 3)))
    for r, w in zip(wave[0:512:3], ws):
        _assert_same_as_oracle(r, w)
    # a batch smaller than one workgroup of slots, and odd sizes (ragged last workgroup)
    for n in (1, 7, 9, 65):
        _same_results(path_planner.BatchPlanner(dm, max_nodes=8192, mode=form).plan(starts[:n], goals[:n]),
                      path_planner.BatchPlanner(dm, max_nodes=8192, mode=1).plan(starts[:n], goals[:n]))


def test_wave_form_auto_mode_and_slots(vehicle, cfg):
    """mode 0 picks the wave form for batches much larger than the chip; a caller-given slot count is honoured; results
    unchanged."""
    import ctypes as C
    from automatedvaletparking_amd import _native, path_planner
    m = case_map_from_gold(1)
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=40)
    L = _native.lib()
    ncu = int(L.avp_plan_slots(dm.h, C.c_int32(1)))
    assert int(L.avp_plan_slots(dm.h, C.c_int32(2))) == int(L.avp_plan_group(C.c_int32(2))) * ncu
    pick = lambda k: int(L.avp_plan_pick_mode(dm.h, C.c_int64(k * ncu), C.c_int32(0)))
    assert (pick(2), pick(10), pick(11), pick(23), pick(24), pick(79), pick(80), pick(200)) == (1, 1, 4, 4, 3, 3, 2, 2)
    rng = np.random.default_rng(8)
    b = m.boundary
    n = 600
    poses = np.stack([rng.uniform(b[0] + 6, b[1] - 6, 6 * n), rng.uniform(b[2] + 6, b[3] - 6, 6 * n), rng.uniform(-np.pi, np.pi, 6 * n)], 1)
    free = poses[dm.check_batch(poses) == 0]
    assert len(free) >= 2 * n
    st, go = free[0:2 * n:2], free[1:2 * n:2]
    ref = path_planner.BatchPlanner(dm, max_nodes=4096, mode=1).plan(st, go)
    for form in FORMS:
        grp = int(L.avp_plan_group(C.c_int32(form)))
        assert int(L.avp_plan_slots(dm.h, C.c_int32(form))) == grp * ncu
        _same_results(path_planner.BatchPlanner(dm, max_nodes=4096, mode=form).plan(st, go), ref)
        _same_results(path_planner.BatchPlanner(dm, max_nodes=4096, mode=form, n_slots=3 * grp).plan(st, go), ref)   # 3 workgroups
    big = 80 * ncu                                                                                       # auto -> wave form
    sb, gb = np.tile(st, (big // n + 1, 1))[:big], np.tile(go, (big // n + 1, 1))[:big]
    auto = path_planner.BatchPlanner(dm, max_nodes=4096).plan(sb, gb)
    _same_results(auto[:n], ref)
    assert len({r.counters["h_cells"] for r in auto}) > 10


@pytest.mark.parametrize("name", ["g6_trace_case1.npz", "g6_trace_case13.npz", "g6_trace_case5.npz", "g6_trace_case20.npz", "g8_synth_c5_plan0.npz",
                                  "g10_variant_circle_case4_0.npz", "g10_variant_steer7_r6_case4_1.npz", "g10_variant_dt08_case4_0.npz",
                                  "g10_variant_margins_rsall_case4_2.npz", "g10_variant_steer17_case4_0.npz", "g10_variant_dt10_ddt02_case4_0.npz"])
@pytest.mark.parametrize("form", FORMS)
def test_wave_form_golden_problems(form, name, vehicle, cfg):
    """Single golden problems through the wave form (8 slots, 7 idle): long searches (Case13: 5 681 pops), NO_PATH
    (Case20), the two-circle checker, 7 steering angles (14 children), other time steps, RS shot at every pop; round 6: 5 sub-steps
    (10 x 5 = 50 poses per expansion: a group holds 64) and 17 steering angles (34 children: more than a group's 16 -- such a
    problem is handed to the workgroup form inside the same call)."""
    import json
    from automatedvaletparking_amd import path_planner, _native
    from oracle import oracle
    g = np.load(os.path.join(GOLD, name))
    m, st, go = _gold_problem(g)
    cfgp = dict(cfg)
    if "synth_c5" in name:
        cfgp["flag_radius"] = 1e9
    if "cfg_json" in g.files:
        cfgp.update(json.loads(str(g["cfg_json"])))
    cap = 30000
    dm = _native.DeviceMap(m, vehicle, cfgp, max_pops=cap)
    res = path_planner.BatchPlanner(dm, n_slots=16, max_nodes=1 << 19, mode=form).plan(st[None, :], go[None, :], max_trace=cap)[0]
    with oracle.device_arithmetic():
        w = oracle.Oracle(m, vehicle, cfgp, max_pops=cap).plan(st, go, max_trace=cap)
    _assert_same_as_oracle(res, w)


@pytest.mark.parametrize("form", FORMS)
def test_wave_form_hands_long_shots_back(form, vehicle, cfg, tmp_path):
    """A Reeds-Shepp shot of more than 256 samples (> 128 m) does not fit the wave form's sample buffer: the problem is
    planned by the workgroup form inside the same call, with the same result; 16 children (8 steering angles) too."""
    from automatedvaletparking_amd import costmap, sampling, _native, path_planner
    polys = [np.array([[2.0, 2.0], [3.0, 2.0], [3.0, 3.0], [2.0, 3.0]])]
    csv = tmp_path / "long.csv"
    sampling.write_tpcap_csv(str(csv), (10.0, 30.0, 0.0), (190.0, 30.0, 0.0), polys)
    m = costmap.Map(file=str(csv), discrete_size=0.2)
    c2 = dict(cfg)
    c2["flag_radius"] = 1e9
    starts = np.array([[10.0 + 0.5 * k, 30.0 + 0.1 * k, 0.01 * k] for k in range(16)])
    goals = np.array([[190.0 - 0.3 * k, 30.0, 0.0] for k in range(16)])
    goals[8:] = starts[8:] + np.array([6.0, 1.0, 0.3])          # short ones stay in the wave form
    dm = _native.DeviceMap(m, vehicle, c2, max_pops=50)
    wave = path_planner.BatchPlanner(dm, max_nodes=4096, mode=form, max_path=1024).plan(starts, goals, max_trace=50)
    wg = path_planner.BatchPlanner(dm, max_nodes=4096, mode=1, max_path=1024).plan(starts, goals, max_trace=50)
    _same_results(wave, wg)
    assert all(r.status == 0 for r in wave) and max(r.n_rs_pts for r in wave) > 256
    c3 = dict(cfg)
    c3["steering_angle_num"] = 8                                   # 16 children: the wave form's maximum
    dm3 = _native.DeviceMap(m, vehicle, c3, max_pops=40)
    _same_results(path_planner.BatchPlanner(dm3, max_nodes=4096, mode=form).plan(starts[8:], goals[8:], max_trace=40),
                  path_planner.BatchPlanner(dm3, max_nodes=4096, mode=1).plan(starts[8:], goals[8:], max_trace=40))


@pytest.mark.parametrize("form", FORMS)
def test_group_forms_shot_at_every_pop(form, vehicle, cfg):
    """config[4]'s parking lot (dense clutter, flag_radius 1e9: the Reeds-Shepp shot and its collision passes run at every
    pop, the collision queue overflows and ranges are halved): 256 starts through every group form == the workgroup form."""
    from automatedvaletparking_amd import _native, path_planner, workloads
    m, c5, st, go, _ = workloads.c5_problems(cfg, 256, device="cuda")
    dm = _native.DeviceMap(m, vehicle, c5, max_pops=150)
    ref = path_planner.BatchPlanner(dm, max_nodes=8192, mode=1, lookahead=False).plan(st, go, max_trace=150)
    got = path_planner.BatchPlanner(dm, max_nodes=8192, mode=form).plan(st, go, max_trace=150)
    _same_results(got, ref)
    assert all(r.counters["n_rs"] >= r.n_pops for r in ref)


@pytest.mark.parametrize("form", FORMS)
def test_time_sliced_group_forms(form, vehicle, cfg):
    """More problems than the form has groups, a workspace slot per problem: searches are parked after slice_pops pops
    while others wait and resumed by whichever group is free (another CU, another XCD). Records, traces, counters and paths
    equal the unsliced launch's and the workgroup form's, for a short and the default slice."""
    import ctypes as C
    from automatedvaletparking_amd import _native, path_planner
    m = case_map_from_gold(1)
    cap = 120
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=cap)
    L = _native.lib()
    n = int(L.avp_plan_slots(dm.h, C.c_int32(form))) + 150
    rng = np.random.default_rng(31 + form)
    b = m.boundary
    poses = np.stack([rng.uniform(b[0] + 6, b[1] - 6, 6 * n), rng.uniform(b[2] + 6, b[3] - 6, 6 * n), rng.uniform(-np.pi, np.pi, 6 * n)], 1)
    free = poses[dm.check_batch(poses) == 0]
    assert len(free) >= 2 * n
    st, go = free[0:2 * n:2], free[1:2 * n:2]
    plain = path_planner.BatchPlanner(dm, max_nodes=4096, mode=form, time_slice=False)
    ref = plain.plan(st, go, max_trace=cap)
    assert not plain.last_time_sliced
    assert sum(r.n_pops > 64 for r in ref) > 50 and sum(r.status == 0 for r in ref) > n // 2
    for sp in (8, None):
        bp = path_planner.BatchPlanner(dm, max_nodes=4096, mode=form, time_slice=True, slice_pops=sp)
        got = bp.plan(st, go, max_trace=cap)
        assert bp.last_time_sliced
        _same_results(got, ref)
        _same_results(bp.plan(st, go, max_trace=cap), ref)          # again on the same workspace (ring words re-zeroed)
    _same_results(ref[:256], path_planner.BatchPlanner(dm, max_nodes=4096, mode=1, lookahead=False).plan(st[:256], go[:256], max_trace=cap))


def test_time_slicing_soak(vehicle, cfg):
    """The resume ring under load: 1 536 problems in the quad form with slices of 4 pops -- every long search changes groups
    dozens of times, between CUs and XCDs --, 25 consecutive launches, every launch's records and way-points equal to the
    unsliced launch's (scripts/slice_soak.py runs the longer version for profiles/)."""
    import ctypes as C
    import hashlib
    import torch
    from automatedvaletparking_amd import _native, path_planner, workloads
    m = case_map_from_gold(1)
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=200)
    L = _native.lib()
    n = int(L.avp_plan_slots(dm.h, C.c_int32(4))) + 512
    st, go = workloads.sample_pairs(m, dm.check_batch, n, np.random.default_rng(77))
    stt, got = dm.dev_tensor(st), dm.dev_tensor(go)

    def digest(bp):
        r, p, _ = bp.plan_dev(stt, got)
        torch.cuda.synchronize()
        rec = r.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:n]
        h = hashlib.sha256()
        for k in ("status", "n_pops", "n_final", "n_checks", "n_rs", "n_closed", "n_open", "h_cells", "h_misses", "global_index", "n_nodes", "rs_L"):
            h.update(np.ascontiguousarray(rec[k]).tobytes())
        pa = p.cpu().numpy()
        for i in range(n):
            h.update(np.ascontiguousarray(pa[i, :int(rec["n_final"][i])]).tobytes())
        return h.hexdigest(), rec
    want, rec = digest(path_planner.BatchPlanner(dm, max_nodes=8192, max_path=256, mode=4, time_slice=False))
    assert int((rec["n_pops"] > 100).sum()) > 100
    on = path_planner.BatchPlanner(dm, max_nodes=8192, max_path=256, mode=4, time_slice=True, slice_pops=4)
    for k in range(25):
        got_d, _ = digest(on)
        assert on.last_time_sliced and got_d == want, k


def test_default_time_slicing_gives_way_when_memory_is_short(vehicle, cfg):
    """time_slice=None is opportunistic: it wants a workspace slot per problem (2.8 MB each at 16 384 nodes), takes them only
    when they fit half of the free device memory, and plans unsliced otherwise -- here a co-tenant tensor leaves 20 GB free;
    8 192 problems in the wave form want 23 GB of slots sliced, 11.5 GB (one slot per group: 4 096) unsliced: the launch goes
    unsliced, last_time_sliced says so, and the results are those of the sliced launch with the memory to itself."""
    import torch
    from automatedvaletparking_amd import _native, path_planner, workloads
    cap = 60
    dm = _native.DeviceMap(case_map_from_gold(1), vehicle, cfg, max_pops=cap)
    m, st, go = workloads.case1_pairs(cfg, lambda mm: dm.check_batch, 4096)
    st, go = np.concatenate([st, st]), np.concatenate([go, np.roll(go, 7, axis=0)])
    ref_bp = path_planner.BatchPlanner(dm, max_nodes=16384, mode=2, time_slice=True)
    ref = ref_bp.plan(st, go)
    assert ref_bp.last_time_sliced is True
    del ref_bp
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(max(free - (20 << 30), 1 << 20), dtype=torch.uint8, device="cuda")
    try:
        bp = path_planner.BatchPlanner(dm, max_nodes=16384, mode=2)            # time_slice=None
        got = bp.plan(st, go)
        assert bp.last_time_sliced is False
        del bp
    finally:
        del hog
        torch.cuda.empty_cache()
    assert all(a.status == b.status and a.n_pops == b.n_pops and np.array_equal(a.final_path, b.final_path) for a, b in zip(ref, got))
