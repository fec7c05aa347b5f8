"""Empty and ragged inputs through every batched entry point (the reference's scalar API has no notion of them; a
batched drop-in must not launch on nothing, read past a short batch, or depend on the batch a problem travels in)."""
import numpy as np
import pytest

from conftest import case_map_from_gold

pytestmark = pytest.mark.gpu


def test_empty_batches(vehicle, cfg):
    from automatedvaletparking_amd import _native, path_planner
    m = case_map_from_gold(1)
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=50)
    assert len(dm.check_batch(np.zeros((0, 3)))) == 0
    assert len(dm.check_batch(np.zeros((0, 3)), kind=1)) == 0
    assert dm.corridor_batch(np.zeros((0, 3)), 5.0).shape == (0, 4)
    r = dm.rs_optimal_batch(np.zeros((0, 3)), np.zeros((0, 3)))
    assert len(r["status"]) == 0 and r["pts"].shape[0] == 0
    assert path_planner.BatchPlanner(dm, max_nodes=4096).plan(np.zeros((0, 3)), np.zeros((0, 3))) == []
    with pytest.raises(ValueError):
        path_planner.BatchPlanner(dm, max_nodes=4096).plan(np.zeros((2, 3)), np.zeros((1, 3)))
    pl = path_planner.PathPlanner(config=cfg, map=m, vehicle=vehicle)
    assert path_planner.split_path_batch([], cfg, vehicle, pl.collision_checker) == []


@pytest.mark.parametrize("n", [1, 63, 64, 65, 129])
def test_result_does_not_depend_on_the_batch_it_travels_in(vehicle, cfg, n):
    """Element i of a batch of n equals element i planned alone / checked alone (ragged last waves and workgroups)."""
    from automatedvaletparking_amd import _native, path_planner
    m = case_map_from_gold(1)
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=60)
    rng = np.random.default_rng(n)
    b = m.boundary
    poses = np.stack([rng.uniform(b[0] + 6, b[1] - 6, 12 * n + 256), rng.uniform(b[2] + 6, b[3] - 6, 12 * n + 256), rng.uniform(-np.pi, np.pi, 12 * n + 256)], 1)
    hit = dm.check_batch(poses)
    for i in range(0, n, max(1, n // 5)):
        assert dm.check_batch(poses[i:i + 1])[0] == hit[i]
    free = poses[hit == 0]
    assert len(free) >= 2 * n
    st, go = free[0:2 * n:2], free[1:2 * n:2]
    bp = path_planner.BatchPlanner(dm, max_nodes=4096)
    res = bp.plan(st, go)
    for i in sorted({0, n // 2, n - 1}):
        one = path_planner.BatchPlanner(dm, max_nodes=4096).plan(st[i:i + 1], go[i:i + 1])[0]
        assert (one.status, one.n_pops) == (res[i].status, res[i].n_pops)
        assert np.array_equal(one.final_path, res[i].final_path) and one.counters == res[i].counters
    rs = dm.rs_optimal_batch(st, go)
    for i in sorted({0, n // 2, n - 1}):
        r1 = dm.rs_optimal_batch(st[i:i + 1], go[i:i + 1])
        assert r1["L"][0] == rs["L"][i] and np.array_equal(r1["lens"][0], rs["lens"][i]) and r1["npts"][0] == rs["npts"][i]
