"""How far the reference's stale-key Dijkstra (in-place decrease-key without re-heapify, compute_h.py:216-235) and the exact
(distance, id) pop order the device realises can differ -- a larger CPU sweep than tests/test_dijkstra_stale_key.py runs every round
(test infrastructure: it drives the oracle). For every BenchmarkCase map, --goals goals (the case's own + random ones), a sequence
of --queries random heuristic queries each, ending with the map's far corners so that the whole reachable map is swept:

  * query distances that differ between the two orders (the claim: none, ever);
  * cells closed in both runs whose distance differs (none);
  * size of the symmetric difference of the closed sets (a few frontier cells);
  * queries whose hit / miss classification flips (the one thing the order can change: the device's `h_misses` counter).

    python scripts/dijkstra_stale_key_sweep.py --goals 24 --queries 40 > profiles/r05_dijkstra_stale_key_sweep.json
"""
import argparse
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_map(args):
    k, n_goals, n_q = args
    from automatedvaletparking_amd import costmap, config
    from oracle import oracle
    from conftest import case_map_from_gold
    cfg = config.default_config()
    m = case_map_from_gold(k)
    o = oracle.Oracle(m, costmap.Vehicle(), cfg)
    b = m.boundary
    rng = np.random.default_rng(9000 + k)

    def run(goal, qs):
        dj = o.dijkstra(goal[0], goal[1])
        ds, miss = [], []
        for i, (x, y) in enumerate(qs):
            d = -1 if i == 0 else dj.lookup(o.pos_to_index(x, y))
            if d < 0:
                d = dj.compute_path(x, y)
                miss.append(1)
            else:
                miss.append(0)
            ds.append(d)
            if d < 0:
                break
        ids, dist, _, _ = dj.dump()
        first = {}
        for i_, d_ in zip(ids.tolist(), dist.tolist()):
            first.setdefault(i_, d_)
        return ds, miss, first

    out = dict(case=k, goals=0, queries=0, query_distance_diffs=0, closed_distance_diffs=0, closed_cells=0, closed_set_xor_max=0, closed_set_xor_sum=0,
               classification_flips=0, goals_with_a_flip=0, unreachable_goals=0)
    goals = [(m.case.xf, m.case.yf)] + [(rng.uniform(b[0] + 2, b[1] - 2), rng.uniform(b[2] + 2, b[3] - 2)) for _ in range(n_goals - 1)]
    for goal in goals:
        qs = np.stack([rng.uniform(b[0] + 0.5, b[1] - 0.5, n_q), rng.uniform(b[2] + 0.5, b[3] - 0.5, n_q)], 1)
        # (a) the random queries alone: the closed sets are partial here -- the frontier is where the two orders can differ
        d_ref, miss_ref, c_ref = run(goal, qs)
        with oracle.exact_dijkstra_order():
            d_ex, miss_ex, c_ex = run(goal, qs)
        x = len(c_ref.keys() ^ c_ex.keys())
        out["closed_set_xor_max"] = max(out["closed_set_xor_max"], x)
        out["closed_set_xor_sum"] += x
        # (b) the same queries followed by the map's four corners: everything reachable is closed, every cell's distance compared
        qs = np.concatenate([qs, [[b[0] + 0.55, b[2] + 0.55], [b[1] - 0.55, b[3] - 0.55], [b[0] + 0.55, b[3] - 0.55], [b[1] - 0.55, b[2] + 0.55]]])
        d_ref, miss_ref, c_ref = run(goal, qs)
        with oracle.exact_dijkstra_order():
            d_ex, miss_ex, c_ex = run(goal, qs)
        out["goals"] += 1
        out["queries"] += len(d_ref)
        if len(d_ref) != len(d_ex):
            out["query_distance_diffs"] += abs(len(d_ref) - len(d_ex))
        out["query_distance_diffs"] += sum(a != b_ for a, b_ in zip(d_ref, d_ex))
        both = c_ref.keys() & c_ex.keys()
        out["closed_cells"] += len(both)
        out["closed_distance_diffs"] += sum(c_ref[i] != c_ex[i] for i in both)
        fl = sum(a != b_ for a, b_ in zip(miss_ref, miss_ex))
        out["classification_flips"] += fl
        out["goals_with_a_flip"] += fl > 0
        out["unreachable_goals"] += d_ref[0] < 0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--goals", type=int, default=24)
    ap.add_argument("--queries", type=int, default=40)
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    with ProcessPoolExecutor(a.procs) as ex:
        rows = list(ex.map(run_map, [(k, a.goals, a.queries) for k in range(1, 21)]))
    tot = {k: sum(r[k] for r in rows) for k in rows[0] if k not in ("case", "closed_set_xor_max")}
    tot["closed_set_xor_max"] = max(r["closed_set_xor_max"] for r in rows)
    print(json.dumps({"what": "reference pop order (stale keys) vs exact (distance, id) order of the heuristic Dijkstra, CPU oracle, 20 BenchmarkCases maps",
                      "goals_per_map": a.goals, "queries_per_goal": a.queries + 4, "total": tot, "per_map": rows}, indent=1))


if __name__ == "__main__":
    main()
