"""Expansion lookahead (avp_plan_batch_ex): workgroups without a problem of their own pre-compute node expansions for
the running searches. A record only ever replaces the very computation it was made by, so every result -- record fields,
counters, pop traces, way-points -- must be identical with and without it, and equal to the oracle's."""
import numpy as np
import pytest

from conftest import case_map_from_gold
from test_gpu_plan_wave import _same_results

pytestmark = pytest.mark.gpu


def _pairs(m, dm, n, seed):
    rng = np.random.default_rng(seed)
    b = m.boundary
    poses = np.stack([rng.uniform(b[0] + 6, b[1] - 6, 8 * n), rng.uniform(b[2] + 6, b[3] - 6, 8 * n), rng.uniform(-np.pi, np.pi, 8 * n)], 1)
    free = poses[dm.check_batch(poses) == 0]
    assert len(free) >= 2 * n
    return free[0:2 * n:2], free[1:2 * n:2]


@pytest.mark.parametrize("n", [1, 5, 40, 256, 300, 512, 900, 1600])          # (900, 1 600: helpers only in the batch's tail, scarce: the owners post fewer nodes)
def test_lookahead_changes_no_result(vehicle, cfg, n):
    from automatedvaletparking_amd import _native, path_planner
    m = case_map_from_gold(1)
    cap = 400
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=cap)
    st, go = _pairs(m, dm, n, 77 + n)
    off = path_planner.BatchPlanner(dm, max_nodes=8192, mode=1, lookahead=False)
    on = path_planner.BatchPlanner(dm, max_nodes=8192, mode=1, lookahead=True)
    a = off.plan(st, go, max_trace=cap)
    b = on.plan(st, go, max_trace=cap)
    assert on.last_lookahead and not off.last_lookahead
    _same_results(a, b)
    _same_results(on.plan(st, go, max_trace=cap), a)        # a second call on the same (re-zeroed) lookahead workspace


@pytest.mark.parametrize("log2", [6, 10])
def test_tiny_record_store_changes_no_result(vehicle, cfg, log2):
    """The record store is a direct-mapped table named by pose hash: with 64 or 1 024 entries for 256 searches tags collide, claims are
    refused because an entry's jobs are in flight, complete records are taken over while their owner may be copying them (seqlock), and
    readers find other poses' records under their tag (key check). None of it may change a result: identical to the launch without the
    lookahead, twice in a row on the same workspace, and the store does get used."""
    from automatedvaletparking_amd import _native, path_planner
    m = case_map_from_gold(1)
    cap = 400
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=cap)
    st, go = _pairs(m, dm, 256, 4242)
    off = path_planner.BatchPlanner(dm, max_nodes=8192, mode=1, lookahead=False)
    tiny = path_planner.BatchPlanner(dm, max_nodes=8192, mode=1, lookahead=True, look_entries_log2=log2)
    a = off.plan(st, go, max_trace=cap)
    for _ in range(2):
        b = tiny.plan(st, go, max_trace=cap)
        assert tiny.last_lookahead and tiny._look.numel() < (32 << 20)
        _same_results(a, b)
        c = tiny._look[:1024].cpu().numpy().view(np.uint64)
        assert int(c[0]) > 0 and int(c[8]) > 0 and int(c[78]) > 0          # jobs posted, records used, claims refused (entry busy)
    # back to the default store on the same handle
    big = path_planner.BatchPlanner(dm, max_nodes=8192, mode=1, lookahead=True)
    _same_results(a, big.plan(st, go, max_trace=cap))
    assert big._look.numel() > (128 << 20)


def test_lookahead_vs_oracle(vehicle, cfg):
    import os
    from concurrent.futures import ThreadPoolExecutor
    from automatedvaletparking_amd import _native, path_planner
    from oracle import oracle
    from test_gpu_plan import _assert_same_as_oracle
    m = case_map_from_gold(5)
    cap = 300
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=cap)
    st, go = _pairs(m, dm, 48, 5)
    on = path_planner.BatchPlanner(dm, max_nodes=8192, mode=1, lookahead=True)
    res = on.plan(st, go, max_trace=cap)
    assert on.last_lookahead
    o = oracle.Oracle(m, vehicle, cfg, max_pops=cap)
    with oracle.device_arithmetic():
        with ThreadPoolExecutor(min(48, os.cpu_count() or 1)) as ex:
            ws = list(ex.map(lambda i: o.plan(st[i], go[i], max_trace=cap), range(48)))
    for r, w in zip(res, ws):
        _assert_same_as_oracle(r, w)


def test_lookahead_only_for_batches_of_the_workgroup_form(vehicle, cfg):
    import ctypes as C
    from automatedvaletparking_amd import _native
    m = case_map_from_gold(1)
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=10)
    L = _native.lib()
    ncu = int(L.avp_plan_slots(dm.h, C.c_int32(1)))
    assert int(L.avp_plan_look_bytes(dm.h, C.c_int64(2 * ncu), C.c_int32(4096))) > 0
    # (rounds 2 - 5 stopped at two problems per CU; since round 6 every batch the library plans in the workgroup form gets helpers in its tail)
    assert int(L.avp_plan_look_bytes(dm.h, C.c_int64(2 * ncu + 1), C.c_int32(4096))) > 0
    assert int(L.avp_plan_look_bytes(dm.h, C.c_int64(11 * ncu - 1), C.c_int32(4096))) > 0
    assert int(L.avp_plan_look_bytes(dm.h, C.c_int64(11 * ncu), C.c_int32(4096))) == 0          # quad form: no helpers
    assert int(L.avp_plan_look_bytes(dm.h, C.c_int64(32 * ncu), C.c_int32(4096))) == 0          # wave form: no helpers
    assert int(L.avp_plan_look_bytes(dm.h, C.c_int64(64 * ncu), C.c_int32(4096))) == 0


def test_problem_order_changes_no_result(vehicle, cfg):
    """avp_plan_batch_ex `order`: the problems start in the caller's order, results stay at their index -- both kernel forms."""
    from automatedvaletparking_amd import _native, path_planner
    m = case_map_from_gold(1)
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=150)
    st, go = _pairs(m, dm, 700, 3)
    for mode in (1, 2):
        a = path_planner.BatchPlanner(dm, max_nodes=4096, mode=mode, lookahead=False, n_slots=64).plan(st, go, max_trace=150)
        b = path_planner.BatchPlanner(dm, max_nodes=4096, mode=mode, lookahead=False, n_slots=64, longest_first=True).plan(st, go, max_trace=150)
        _same_results(a, b)


def test_lookahead_soak_300_launches(vehicle, cfg):
    """The inter-workgroup hand-offs (job rings, record store) under repetition: 300 consecutive launches of config[1]'s
    256 problems with the lookahead on the same workspace -- the record store is never zeroed between launches, records are
    accepted by key -- every launch's records and way-points equal to the launch without the lookahead. (Builds with other
    helper sleep / owner wait times, release / acquire atomics and injected wrong keys: scripts/look_soak.py,
    profiles/r03_lookahead_soak.json.)"""
    import torch
    from automatedvaletparking_amd import _native, path_planner, workloads
    m = case_map_from_gold(1)
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=1000)
    st, go = workloads.sample_pairs(m, dm.check_batch, 256, np.random.default_rng(workloads.SEED))
    stt, got = dm.dev_tensor(st), dm.dev_tensor(go)

    def fetch(res, paths):
        rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:256]
        return rec, paths.cpu().numpy()

    off = path_planner.BatchPlanner(dm, max_nodes=16384, max_path=256, mode=1, lookahead=False)
    r, p, _ = off.plan_dev(stt, got)
    torch.cuda.synchronize()
    r0, p0 = fetch(r, p)
    on = path_planner.BatchPlanner(dm, max_nodes=16384, max_path=256, mode=1, lookahead=True)
    fields = [f for f in r0.dtype.names if f not in ("slot", "phase_cycles")]
    for k in range(300):
        r, p, _ = on.plan_dev(stt, got)
        torch.cuda.synchronize()
        assert on.last_lookahead
        rk, pk = fetch(r, p)
        for f in fields:
            assert np.array_equal(rk[f], r0[f]), (k, f)
        assert all(np.array_equal(pk[i, :r0["n_final"][i]], p0[i, :r0["n_final"][i]]) for i in range(256)), k
    used = int(on._look[:1024].cpu().numpy().view(np.uint64)[8])
    assert used > 0.5 * r0["n_pops"].sum()          # most pops were served from a record
