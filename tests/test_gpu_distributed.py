"""The multi-GPU data path WITH THE REAL HIP PLANNER under world_size 2, on one GPU box.

tests/test_distributed_gloo.py runs automatedvaletparking_amd.distributed with the CPU oracle standing in for the planner;
bench.py --gpus N runs it with the kernels, but the driver's scaling run is not the builder's to launch. Here two
processes share cuda:0 (RCCL refuses two ranks on one device, so the collectives are gloo on CPU staging copies -- the
helpers in distributed.py are backend agnostic and move whatever tensors the stage callables return) and run exactly
what `bench.py --gpus 2` runs per step:

  * the weak step (`plan_weak`): every rank plans its own contiguous block with BatchPlanner in the bench's headline
    configuration (workgroup form + expansion lookahead -- two such launches share the device here), ONE gather to rank 0;
  * the two-stage deal (`two_stage_plan`): BatchPlanner(mode=STAGED, first_stage_only) on the index slice, the records
    all-gathered, the searches still running dealt round-robin and planned in `long_search_mode`'s form, way-points
    gathered to rank 0 only -- on 513 problems, so that one first-stage shard is padded and the second stage is uneven
    or not as the statuses fall.

The gathered result must equal the single-process result of the same planner record for record and way-point for
way-point, and that one the pinned oracle (tests/_parity.py) on every observable field."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CAP = 300
N = 513
MAX_NODES = 8192
MAX_PATH = 256
STAGE_POPS = 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problems(cfg, dm_check):
    from automatedvaletparking_amd import workloads
    m = workloads.case_map(1, cfg)
    st, go = workloads.sample_pairs(m, dm_check(m), N, np.random.default_rng(20260928), chunk=4096)
    return m, st, go


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from automatedvaletparking_amd import _native, config, costmap, path_planner, distributed as avd
    cfg = config.default_config()
    veh = costmap.Vehicle()
    # rank 0 builds the map and samples the problems; the others receive both (the bench's untimed set-up)
    if rank == 0:
        m, st, go = _problems(cfg, lambda mm: _native.DeviceMap(mm, veh, cfg, max_pops=CAP).check_batch)
        prob = torch.as_tensor(np.concatenate([st, go], 1))
    else:
        m, prob = None, torch.empty((N, 6), dtype=torch.float64)
    m = avd.broadcast_map(m, src=0, device="cpu")
    dist.broadcast(prob, src=0)
    pr = prob.numpy()
    st, go = pr[:, :3].copy(), pr[:, 3:].copy()
    dm = _native.DeviceMap(m, veh, cfg, device=0, max_pops=CAP)

    def staged(t):          # device tensors -> the CPU staging copies the gloo collectives move
        return t.cpu()

    # ---- weak step: this rank's block of the first 512, the bench's headline planner, one gather to rank 0 ----------------
    bp_w = path_planner.BatchPlanner(dm, max_nodes=MAX_NODES, max_path=MAX_PATH)
    used = {}

    def block(s_l, g_l):
        r, p, _ = bp_w.plan_dev(dm.dev_tensor(s_l), dm.dev_tensor(g_l), want_paths=True)
        used["lookahead"] = bool(bp_w.last_lookahead)
        return staged(r), staged(p)

    nw = (N // world) * world
    recw, pathw = avd.plan_weak(block, st[:nw], go[:nw], rank, world, dst=0)
    assert (recw is None) == (rank != 0)
    # ---- two-stage deal over all 513 -----------------------------------------------------------------------------------------
    bp1 = path_planner.BatchPlanner(dm, max_nodes=MAX_NODES, max_path=MAX_PATH, mode=path_planner.STAGED, stage_pops=STAGE_POPS)
    bp2 = {}
    seen = {"stage2_n": 0, "stage2_mode": 0}

    def stage1(s_l, g_l):
        r, p, _ = bp1.plan_dev(dm.dev_tensor(s_l), dm.dev_tensor(g_l), want_paths=True, first_stage_only=True)
        return staged(r), staged(p)

    def stage2(s_l, g_l):
        mode2 = path_planner.long_search_mode(dm, len(s_l))
        if mode2 not in bp2:
            bp2[mode2] = path_planner.BatchPlanner(dm, max_nodes=MAX_NODES, max_path=MAX_PATH, mode=mode2)
        seen["stage2_n"], seen["stage2_mode"] = len(s_l), mode2
        r, p, _ = bp2[mode2].plan_dev(dm.dev_tensor(s_l), dm.dev_tensor(g_l), want_paths=True)
        return staged(r), staged(p)

    rec2, path2, deferred = avd.two_stage_plan(stage1, stage2, st, go, rank, world, paths_to=0)
    assert (path2 is None) == (rank != 0) and len(rec2) == N
    # every rank holds the same records after the all-gathers
    chk = [torch.zeros_like(rec2) for _ in range(world)]
    dist.all_gather(chk, rec2.contiguous())
    assert all(torch.equal(c, rec2) for c in chk)
    if rank == 0:
        q.put(dict(st=st, go=go, blob=avd.pack_map_blob(m), recw=recw.numpy().copy(), pathw=pathw.numpy().copy(), rec2=rec2.numpy().copy(),
                   path2=path2.numpy().copy(), deferred=np.asarray(deferred).copy(), lookahead=used.get("lookahead"), **seen))
    dist.barrier()
    dist.destroy_process_group()


def _same(ra, pa, rb, pb):
    names = [f for f in ra.dtype.names if f not in ("slot", "phase_cycles")]
    for f in names:
        if not np.array_equal(ra[f], rb[f]):
            return "records differ in %s at %s" % (f, np.where(ra[f] != rb[f])[0][:5])
    for i in range(len(ra)):
        k = int(ra["n_final"][i])
        if not np.array_equal(pa[i, :k], pb[i, :k]):
            return "way-points of problem %d differ" % i
    return ""


@pytest.mark.timeout(900)
def test_world2_real_planner_on_one_gpu(vehicle, cfg):
    import torch
    import torch.multiprocessing as mp
    import _parity
    from automatedvaletparking_amd import _native, path_planner, distributed as avd
    from oracle import oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _queue
    import time
    got, t0 = None, time.time()
    try:
        while got is None:
            try:
                got = q.get(timeout=5)
            except _queue.Empty:
                assert all(p.exitcode in (None, 0) for p in procs), "a rank died: exit codes %s" % [p.exitcode for p in procs]
                assert time.time() - t0 < 600, "no result from rank 0 within 600 s"
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()                                # (the exact processes this test started)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    # ---- the same planner in ONE process ---------------------------------------------------------------------------------------
    m, st, go = _problems(cfg, lambda mm: _native.DeviceMap(mm, vehicle, cfg, max_pops=CAP).check_batch)
    assert np.array_equal(st, got["st"]) and np.array_equal(go, got["go"]) and np.array_equal(avd.pack_map_blob(m), got["blob"])
    dm = _native.DeviceMap(m, vehicle, cfg, max_pops=CAP)
    bp = path_planner.BatchPlanner(dm, max_nodes=MAX_NODES, max_path=MAX_PATH)
    r1, p1, _ = bp.plan_dev(dm.dev_tensor(st), dm.dev_tensor(go), want_paths=True)
    torch.cuda.synchronize()
    one = r1.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:N]
    one_p = p1.cpu().numpy()
    # the weak step: 2 x 256 problems, gathered in problem order
    nw = (N // world) * world
    recw = got["recw"].view(path_planner.RESULT_DTYPE).reshape(-1)
    assert len(recw) == nw and got["lookahead"] is True, "the weak step is meant to run the headline form (workgroup form + lookahead)"
    d = _same(recw, got["pathw"], one[:nw], one_p[:nw])
    assert not d, "weak step: " + d
    # the two-stage deal: all 513, one padded first-stage shard (513 is odd), a non-empty second stage
    rec2 = got["rec2"].view(path_planner.RESULT_DTYPE).reshape(-1)
    assert len(rec2) == N and got["stage2_n"] > 0 and 0 < len(got["deferred"]) < N
    assert got["stage2_n"] == (len(got["deferred"]) + world - 1) // world
    d = _same(rec2, got["path2"], one, one_p)
    assert not d, "two-stage deal: " + d
    long_ones = set(np.where(one["n_pops"] > STAGE_POPS)[0].tolist())
    assert long_ones <= set(got["deferred"].tolist()), "every search longer than the first stage's budget went through stage 2"
    # ---- and the single-process result against the pinned oracle, every observable field --------------------------------------
    res = bp.plan(st, go, max_trace=CAP)
    bad, h_diff = _parity.compare_pinned(oracle.Oracle(m, vehicle, cfg, max_pops=CAP), res, st, go, CAP)
    assert not bad, (len(bad), bad[:6])
    assert [r.status for r in res] == one["status"].tolist() and [r.n_pops for r in res] == one["n_pops"].tolist()
