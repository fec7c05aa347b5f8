"""GPU result vs the CPU oracle in its PINNED mode: the platform's glibc libm and the reference's own heuristic-Dijkstra
pop order (in-place decrease-key, compute_h.py:216-235) -- the mode tests/test_oracle_golden.py pins bit for bit to the 215
golden fixtures captured from the unmodified reference. Test infrastructure.

Everything the reference can observe is compared with no tolerance: status, pop count, the whole pop trace (node index,
parent, grid id, pose, g, h, f, gear), the A* counters, the A* path, the Reeds-Shepp tail and the final path. One internal
counter is not observable and is treated apart: `h_misses` (how many heuristic queries extended the sweep). The device
sweeps in exact (distance, id) order, the reference occasionally pops a cell one step early (stale heap keys), which can
turn a later query's hit into a miss or back -- never a distance (tests/test_dijkstra_stale_key.py). Where the GPU's
h_misses differs from the pinned oracle's, the problem is planned again with the oracle's exact-order switch and must then
agree completely, h_misses included; the number of such problems is returned so that the tests can print / bound it."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

THREADS = min(256, os.cpu_count() or 1)
COUNTERS = ("n_closed", "n_open", "global_index", "n_rs", "n_checks")


def observable_diff(r, w):
    """'' when GPU PlanResult r equals oracle dict w in every observable field, else the name of the first that differs."""
    if r.status != w["status"]:
        return "status %s != %s" % (r.status, w["status"])
    if r.n_pops != w["n_pops"]:
        return "n_pops %d != %d" % (r.n_pops, w["n_pops"])
    t, wt = r.trace, w["trace"]
    if len(t) and not np.array_equal(t[:, :10], wt[:len(t), :10]):
        k = int(np.where((t[:, :10] != wt[:len(t), :10]).any(axis=1))[0][0])
        return "trace row %d" % k
    c = r.counters
    for k in COUNTERS:
        if c[k] != w[k]:
            return "%s %d != %d" % (k, c[k], w[k])
    if r.status in (0, 1):
        if not np.array_equal(np.asarray(r.astar_path), np.asarray(w["astar_path"])):
            return "astar_path"
        if not np.array_equal(np.asarray(r.final_path), np.asarray(w["final_path"])):
            return "final_path"
        if not (np.array_equal(r.rs_xyyaw, w["rs_xyyaw"]) and np.array_equal(r.rs_dirs, w["rs_dir"])):
            return "rs samples"
        if r.rs_L != w["rs_L"] or list(r.rs_lengths) != list(w["rs_lengths"]):
            return "rs lengths"
    return ""


def compare_pinned(o, res, starts, goals, cap, threads=THREADS):
    """o: oracle.Oracle bound to the same map / vehicle / config / cap. -> (bad, h_diff): bad = [(index, what)] problems
    that differ from the pinned oracle in an observable field or, after the exact-order re-run, in h_misses; h_diff =
    indices whose h_misses differs from the reference-order oracle's (explained by the stale-key pops, see above)."""
    from oracle import oracle
    assert oracle.lib().orc_get_dij_reheap() == 0 and oracle.lib().orc_get_restated_libm() == 0, "not the pinned mode"
    with ThreadPoolExecutor(threads) as ex:
        ws = list(ex.map(lambda sg: o.plan(sg[0], sg[1], max_trace=cap), zip(starts, goals)))
    bad, h_diff = [], []
    for i, (r, w) in enumerate(zip(res, ws)):
        d = observable_diff(r, w)
        if d:
            bad.append((i, d))
        elif r.counters["h_misses"] != w["n_dij_calls"]:
            h_diff.append(i)
    if h_diff:
        with oracle.exact_dijkstra_order():
            with ThreadPoolExecutor(threads) as ex:
                w2 = list(ex.map(lambda i: o.plan(starts[i], goals[i], max_trace=cap), h_diff))
        for i, w in zip(h_diff, w2):
            d = observable_diff(res[i], w)
            if d or res[i].counters["h_misses"] != w["n_dij_calls"]:
                bad.append((i, "exact-order re-run: " + (d or "h_misses")))
    return bad, h_diff
