"""Heuristic field on the device vs the oracle Dijkstra (compute_h.py restatement) in exact (distance, id) pop order
(`oracle.exact_dijkstra_order`, the order the bucketed sweep realises): per-query distances, hit/miss classification,
the closed set and every closed distance are IDENTICAL -- no tolerance -- for maps with and without aliased grid ids
and for goals near the map edges (wormhole rows). Exact order vs the reference's stale-key order is bounded on the
CPU on all 20 maps by tests/test_dijkstra_stale_key.py."""
import numpy as np
import pytest

from conftest import gold, case_map_from_gold

pytestmark = pytest.mark.gpu


def _run(k, goal, queries, vehicle, cfg):
    from automatedvaletparking_amd import _native
    from oracle import oracle
    m = case_map_from_gold(k)
    dm = _native.DeviceMap(m, vehicle, cfg)
    o = oracle.Oracle(m, vehicle, cfg)
    with oracle.exact_dijkstra_order():
        return _run_exact(o, dm, goal, queries)


def _run_exact(o, dm, goal, queries):
    dj = o.dijkstra(goal[0], goal[1])
    want_d, want_miss = [], []
    for i, (x, y) in enumerate(queries):
        gid = o.pos_to_index(x, y)
        d = -1 if i == 0 else dj.lookup(gid)
        if d < 0:
            d = dj.compute_path(x, y)
            want_miss.append(1)
        else:
            want_miss.append(0)
        want_d.append(d)
        if d < 0:
            break
    force = np.zeros(len(queries), np.int32)
    force[0] = 1
    r = dm.hfield_queries(goal, queries[:len(want_d)], force[:len(want_d)])
    assert r["info"][0] == 0
    assert list(r["d"]) == want_d
    assert list(r["miss"]) == want_miss
    ids, dist, _, _ = dj.dump()
    # every cell the reference closed has the same distance on the device
    first = {}
    for i_, d_ in zip(ids.tolist(), dist.tolist()):
        first.setdefault(i_, d_)
    goal_id = int(r["info"][3])
    gd = r["dist"]
    bad = [(i_, d_, int(gd[i_])) for i_, d_ in first.items() if i_ != goal_id and int(gd[i_]) != d_]
    assert not bad, bad[:5]
    # and the device's closed set (key <= frontier) is the same set of ids
    dF, idF = int(r["info"][1]), int(r["info"][2])
    seen = np.where(gd != 0x7fffffff)[0]
    if want_d[-1] < 0:
        # unreachable query: the reference swept every reachable cell before blocking (compute_h.py:77)
        closed_dev = {int(i_) for i_ in seen}
        dF, idF = 1 << 40, 0
    else:
        closed_dev = {int(i_) for i_ in seen if (int(gd[i_]), int(i_)) <= (dF, idF)}
    closed_ref = set(first.keys())
    closed_ref.discard(goal_id) if int(gd[goal_id]) == 0x7fffffff or (int(gd[goal_id]), goal_id) > (dF, idF) else None
    diff = closed_dev ^ closed_ref
    assert not diff, (len(diff), sorted(diff)[:6])
    return r


@pytest.mark.parametrize("k", [1, 5, 13, 19, 9])
def test_hfield_query_sequences(k, vehicle, cfg):
    m = case_map_from_gold(k)
    b = m.boundary
    rng = np.random.default_rng(1000 + k)
    goals = [(m.case.xf, m.case.yf), (b[0] + 2.3, b[2] + 2.1), (b[1] - 1.7, b[3] - 2.9), (b[1] - 0.31, 0.5 * (b[2] + b[3]))]
    for goal in goals:
        qs = np.stack([rng.uniform(b[0] + 0.5, b[1] - 0.5, 14), rng.uniform(b[2] + 0.5, b[3] - 0.5, 14)], 1)
        # increasing-then-random distances exercise both hits and misses; last queries at the far corners
        qs = np.concatenate([qs, [[b[0] + 0.55, b[2] + 0.55], [b[1] - 0.55, b[3] - 0.55]]])
        _run(k, goal, qs, vehicle, cfg)


def test_hfield_golden_queries_case1(vehicle, cfg):
    g = gold("g6_trace_case1.npz")
    m = case_map_from_gold(1)
    q = g["h_queries"]
    r = _run(1, (m.case.xf, m.case.yf), q[:, :2], vehicle, cfg)
    assert list(r["d"]) == [int(v) for v in q[:, 2]]
