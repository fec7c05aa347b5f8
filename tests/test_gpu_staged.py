"""The staged call (avp_plan_batch_staged; BatchPlanner mode STAGED): every problem in the wave form for a few pops, the
searches still running then planned again from scratch in the pair / workgroup form. A restarted search is the same
search: every record field, counter, trace row and way-point equals the one-kernel result (and, through
tests/test_gpu_plan_wave.py / test_gpu_plan.py, the CPU oracle's)."""
import numpy as np
import pytest

from conftest import case_map_from_gold
from test_gpu_plan_wave import _same_results

pytestmark = pytest.mark.gpu


def _case1_problems(vehicle, cfg, n_pairs=256, cap=400):
    from automatedvaletparking_amd import _native, workloads
    m = case_map_from_gold(1)
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=cap)
    st, go = workloads.sample_pairs(m, dm.check_batch, n_pairs, np.random.default_rng(20260927))
    return m, dm, st, go


@pytest.mark.parametrize("stage_pops", [1, 16, 64])
def test_staged_equals_one_kernel(stage_pops, vehicle, cfg):
    from automatedvaletparking_amd import path_planner
    cap = 400
    m, dm, st, go = _case1_problems(vehicle, cfg, 256, cap)
    starts = np.concatenate([st, st, st])                      # 768 problems: more unfinished searches than compute units -> the pair form runs
    goals = np.concatenate([go, np.roll(go, 7, axis=0), np.roll(go, 31, axis=0)])
    ref = path_planner.BatchPlanner(dm, max_nodes=8192, mode=1, lookahead=False).plan(starts, goals, max_trace=cap)
    got = path_planner.BatchPlanner(dm, max_nodes=8192, mode=path_planner.STAGED, stage_pops=stage_pops).plan(starts, goals, max_trace=cap)
    _same_results(got, ref)
    assert sum(r.status == 4 for r in ref) > 30 and sum(r.n_pops > 64 for r in ref) > 60
    for n in (1, 5, 33):                                      # fewer unfinished searches than compute units -> plan_kernel finishes them
        _same_results(path_planner.BatchPlanner(dm, max_nodes=8192, mode=path_planner.STAGED, stage_pops=stage_pops).plan(starts[:n], goals[:n], max_trace=cap), ref[:n])


def test_first_stage_alone_defers_the_long_searches(vehicle, cfg):
    import torch
    from automatedvaletparking_amd import path_planner
    cap = 400
    m, dm, st, go = _case1_problems(vehicle, cfg, 128, cap)
    ref = path_planner.BatchPlanner(dm, max_nodes=8192, mode=1, lookahead=False).plan(st, go)
    bp = path_planner.BatchPlanner(dm, max_nodes=8192, mode=path_planner.STAGED, stage_pops=16)
    res, paths, _ = bp.plan_dev(dm.dev_tensor(st), dm.dev_tensor(go), first_stage_only=True)
    torch.cuda.synchronize()
    rec = res.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:len(st)]
    for i, r in enumerate(ref):
        if r.n_pops <= 16:                                     # finished within the stage: final
            assert rec["status"][i] == r.status and rec["n_pops"][i] == r.n_pops and rec["n_final"][i] == len(r.final_path)
        else:
            assert rec["status"][i] == 100                    # AVP_PLAN_DEFERRED
    assert (rec["status"] == 100).sum() > 10


def test_staged_shot_at_every_pop(vehicle, cfg):
    """config[4]'s map (RS shot at every pop, flag_radius 1e9), 300 starts."""
    from automatedvaletparking_amd import _native, path_planner, workloads
    m, c5, st, go, _ = workloads.c5_problems(cfg, 300, device="cuda")
    dm = _native.DeviceMap(m, vehicle, c5, max_pops=120)
    ref = path_planner.BatchPlanner(dm, max_nodes=8192, mode=1, lookahead=False).plan(st, go, max_trace=120)
    got = path_planner.BatchPlanner(dm, max_nodes=8192, mode=path_planner.STAGED, stage_pops=8).plan(st, go, max_trace=120)
    _same_results(got, ref)
