"""The bench's problem sets (automatedvaletparking_amd/workloads.py) are rejection-sampled with a footprint check: bench.py
and the GPU tests use the HIP kernel, the CPU parity-chain test (tests/test_parity_chain_workloads.py) the oracle. The two
checks agree bit for bit, so the sets must be the same arrays either way."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gpu(veh, cfg, cap):
    from automatedvaletparking_amd import _native
    return lambda m: _native.DeviceMap(m, veh, cfg, max_pops=cap).check_batch


def _cpu(veh, cfg):
    from oracle import oracle

    def make(m):
        o = oracle.Oracle(m, veh, cfg)
        return lambda poses: np.asarray(o.check_batch(poses, kind=0)).astype(bool)
    return make


def test_problem_sets_do_not_depend_on_the_checker(vehicle, cfg):
    from automatedvaletparking_amd import workloads
    for n in (256, 4096):
        _, sg, gg = workloads.case1_pairs(cfg, _gpu(vehicle, cfg, 1000), n)
        _, sc, gc = workloads.case1_pairs(cfg, _cpu(vehicle, cfg), n)
        assert np.array_equal(sg, sc) and np.array_equal(gg, gc)
    for k in (3, 13, 19):
        mg, sg, gg = workloads.c3_map_pairs(k, cfg, _gpu(vehicle, cfg, 300), 128, device="cuda")
        mc, sc, gc = workloads.c3_map_pairs(k, cfg, _cpu(vehicle, cfg), 128)
        assert np.array_equal(mg.cost_map, mc.cost_map) and np.array_equal(sg, sc) and np.array_equal(gg, gc)
    mg, _ = workloads.c4_map(device="cuda")
    mc, _ = workloads.c4_map()
    assert np.array_equal(mg.cost_map, mc.cost_map)
    sg, gg = workloads.c4_plan_pairs(mg, _gpu(vehicle, cfg, 300), 256)
    sc, gc = workloads.c4_plan_pairs(mc, _cpu(vehicle, cfg), 256)
    assert np.array_equal(sg, sc) and np.array_equal(gg, gc)
