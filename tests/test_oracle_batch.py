"""orc_plan_batch (oracle/avp_oracle.c): the all-core CPU baseline of bench.py -- pthreads, one problem per thread from an
atomic ticket counter, no Python in the loop. It must be the same planner as orc_plan: per-problem status and pop count
equal to the single-call results whatever the thread count, and the steady-state mode's totals must be consistent."""
import os

import numpy as np

from automatedvaletparking_amd import workloads


def _setup(vehicle, cfg, n=24, cap=60):
    from oracle import oracle
    m = workloads.case_map(1, cfg)
    o0 = oracle.Oracle(m, vehicle, cfg)
    st, go = workloads.sample_pairs(m, lambda p: np.asarray(o0.check_batch(p, kind=0)).astype(bool), n, np.random.default_rng(7))
    return oracle.Oracle(m, vehicle, cfg, max_pops=cap), st, go


def test_one_pass_equals_single_calls(vehicle, cfg):
    o, st, go = _setup(vehicle, cfg)
    want = [o.plan(s, g, max_trace=1) for s, g in zip(st, go)]
    for threads in (1, 3, 8):
        b = o.plan_batch(st, go, threads=threads)
        assert b["threads"] == threads and b["plans"] == len(st)
        assert list(b["status"]) == [w["status"] for w in want]
        assert list(b["n_pops"]) == [w["n_pops"] for w in want]
        assert b["completed"] == sum(w["status"] in (0, 1) for w in want) and b["pops"] == sum(w["n_pops"] for w in want)
        assert b["seconds"] > 0.0


def test_steady_state_cycles_the_set(vehicle, cfg):
    o, st, go = _setup(vehicle, cfg, n=8, cap=20)
    order = np.random.default_rng(0).permutation(len(st)).astype(np.int32)
    b = o.plan_batch(st, go, threads=min(4, os.cpu_count() or 1), min_seconds=0.5, order=order)
    assert b["seconds"] >= 0.5 and b["plans"] >= len(st) and (b["status"] >= 0).all()
    one = o.plan_batch(st, go, threads=1)
    assert list(b["status"]) == list(one["status"]) and list(b["n_pops"]) == list(one["n_pops"])
    # totals are whole multiples of the per-problem figures only up to the plans in flight at the deadline
    assert b["pops"] >= int(one["n_pops"].sum()) * (b["plans"] // len(st))
