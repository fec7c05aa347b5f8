#!/usr/bin/env python3
"""bench.py -- hybrid-A* plans/s on batched poses (BASELINE.json metric), one process per GPU.

A "step" = one pass of the hot path over one batch of problems, inputs resident in HBM.

  N = 1 : headline = BASELINE config[1] (Case1 map, 256 random start/goal pairs, pop cap 1000); extras on the same GPU:
          `batch4096` (north_star's 4 096-pose target workload, every kernel form and the staged call), `scale_point` (the
          same 4 096 set as the N > 1 runs plan it, with the predicted rank loads of both deals), `c3` (20 BenchmarkCases x
          128), `c5` (dense clutter, RS shot at every pop), `saturating_batch`, `cap_sweep`, `cases20` (the 20
          BenchmarkCases' own problems, run to termination), `single_plan_latency_ms`, the footprint kernel alone, and
          the CPU port timed on the host.
  N > 1 : SURVEY 8(e), WEAK scaling (the default): the global set holds N x 256 problems on the Case1 map -- config[1]'s
          sampler and seed, so that N = 1 is the headline set --, rank 0 samples them, the map blob and the problems are
          broadcast (RCCL, untimed set-up), every rank plans its own block of 256 exactly as the N = 1 headline does, and
          ONE gather of the records and way-points to rank 0 closes the timed step
          (automatedvaletparking_amd.distributed.plan_weak): no collective between the launches, per-GPU work fixed.
          Rank 0 then plans the whole set alone and checks that the gathered result is identical (`shard_invariant`).
          Two more points ride on the line (`strong_scaling_4096`, `throughput`): the fixed 4 096-problem set split over
          the ranks by the two-stage deal (distributed.two_stage_plan: first stage on the index slice, records
          all-gathered, the searches still running dealt evenly, way-points gathered to rank 0; with the in-run 1-GPU
          time, `speedup_vs_1gpu`), and N x 16 384 problems, one time-sliced launch per rank, one gather -- the
          saturated regime. `--workload batch4096` makes the strong-scaling set the headline instead.
          AVP_BENCH_FORCE_DIST=1 runs exactly these code paths with world size 1 (through RCCL).
  --workload c3 with N > 1: every rank holds all 20 maps, each map's 128 problems are dealt by index slice.

`value` counts COMPLETED searches only (status OK / NO_PATH); problems stopped by the pop cap (ITER_LIMIT; the reference
has no cap and needs hours on them, DESIGN.md "Workload") are excluded from the numerator and reported separately.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# config[2] is 20 launches on 20 HIP streams. The ROCm runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and
# launches that share a queue run one after the other: with 4, four launches are in flight, each down to its ~40 capped searches after a
# millisecond (164 of 256 CUs busy) until the slowest is done -- 77 ms; with a queue per stream every launch's workgroups are resident as CUs
# fall free -- 51 ms (8 queues: 55, 16: 54). A setting of the HIP runtime of THIS process, read when it initialises (before torch is imported);
# single-process runs only (the multi-rank runs have no multi-stream extra and keep the runtime's default beside RCCL's own streams).
if os.environ.get("WORLD_SIZE", "1") == "1" and os.environ.get("AVP_BENCH_FORCE_DIST") != "1":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, ROOT)

POP_CAP = 1000
MAX_NODES = 16384
MAX_PATH = 256
STAGE_POPS = 16                 # first-stage budget of the staged call / the two-stage deal
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
N_SIMD = 256 * 4                # SIMDs on the chip
CLOCK_GHZ = 2.4                 # max shader clock
FP64_PEAK_TFLOPS = 78.6         # vector fp64: 256 CUs x 128 flop/clk x 2.4 GHz
PMC_FILE = "r06_pmc_summary.json"
CASES = os.path.join(ROOT, "data", "BenchmarkCases")
# the reference's own wall-clock per case (unmodified reference imported in the build container,
# one Python thread): seconds for PathPlanner() + a_star_plan(); None = did not finish / raises
REFERENCE_SECONDS = {1: 52.3, 2: 198.4, 3: 81.7, 4: 70.0, 5: 267.9, 6: 170.9, 7: None, 8: None, 9: 544.7, 10: 1266.6, 11: 1191.7, 12: 728.2, 13: 805.1,
                     14: 110.7, 15: 105.7, 16: 47.6, 17: 12.45, 18: 205.4, 19: None, 20: None}
FORM_NAMES = {1: "one workgroup per problem", 2: "one wave per problem", 3: "a pair of waves per problem", 4: "four waves per problem",
              16: "staged (wave form for %d pops, then the form that suits the number of searches left)" % STAGE_POPS}


def source_hash():
    """sha256 over the kernel sources: stamps PMC summaries so that a stale one is refused (profiles/README.md)."""
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, "automatedvaletparking_amd", "csrc"), os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".h", ".hip", ".inc")):
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_quota():
    """CPUs this process may use: the cgroup's cpu.max quota (a container on a 256-thread host may be held to 16 CPUs' worth of
    time: more runnable threads than that only queue), the affinity mask, or the host's thread count -- whichever is smallest."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class Group:
    """One map + its problems on this rank's GPU."""

    def __init__(self, m, veh, cfg, starts, goals, local, cap=POP_CAP, mode=0, lookahead=None, max_nodes=MAX_NODES, time_slice=None):
        import ctypes as C
        from automatedvaletparking_amd import _native, path_planner
        self.m = m
        self.dm = _native.DeviceMap(m, veh, cfg, device=local, max_pops=cap)
        self.bp = path_planner.BatchPlanner(self.dm, max_nodes=max_nodes, max_path=MAX_PATH, mode=mode, lookahead=lookahead, stage_pops=STAGE_POPS, time_slice=time_slice)
        self.set_problems(starts, goals)
        L = _native.lib()
        # the kernel form that runs: 1 workgroup / 2 wave / 3 pair of waves / 4 four waves per problem, STAGED
        self.mode = mode if mode == path_planner.STAGED else int(L.avp_plan_pick_mode(self.dm.h, C.c_int64(self.n), C.c_int32(mode)))
        self.slots = int(L.avp_plan_slots(self.dm.h, C.c_int32(2 if self.mode == path_planner.STAGED else self.mode)))

    def set_problems(self, starts, goals):
        self.starts, self.goals = np.ascontiguousarray(starts), np.ascontiguousarray(goals)
        self.n = len(starts)
        self.st_t, self.go_t = self.dm.dev_tensor(self.starts), self.dm.dev_tensor(self.goals)

    def plan(self, **kw):
        return self.bp.plan_dev(self.st_t, self.go_t, want_paths=True, **kw)


def plan_groups(groups):
    """One pass over all groups. Several maps (C3: 20 BenchmarkCases x 128 problems) are independent launches of half a
    chip each: every group gets its own HIP stream so that the launches overlap; the caller's stream waits for all."""
    import torch
    if len(groups) == 1:
        return [groups[0].plan()]
    cur = torch.cuda.current_stream()
    outs = []
    for g in groups:
        if not hasattr(g, "stream"):
            g.stream = torch.cuda.Stream()
            # the expansion lookahead parks helper workgroups on every CU its launch leaves free: right for one launch
            # that has the device to itself, wrong beside 19 other launches that want those CUs (measured: 133 vs 83 ms)
            g.bp.lookahead = False
        g.stream.wait_stream(cur)
        with torch.cuda.stream(g.stream):
            outs.append(g.plan())
    for g in groups:
        cur.wait_stream(g.stream)
    return outs


def records(res_t, n):
    from automatedvaletparking_amd import path_planner
    return res_t.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:n]


def summarize(recs, n_slots, elapsed_per_step, time_sliced=False):
    """Throughput figures of one step over the given record arrays (one per group). n_slots: problem slots of the kernel
    form that ran, or a list, one per group. time_sliced: the launch moved searches between groups (a record's slot is
    then the problem's own workspace slot, not the group that ran it): no slot utilisation."""
    if not isinstance(n_slots, (list, tuple)):
        n_slots = [n_slots] * len(recs)
    rec = np.concatenate(recs)
    done = (rec["status"] == 0) | (rec["status"] == 1)
    pops = int(rec["n_pops"].sum())
    # slot utilisation: pops / (slots x pops of the busiest slot), per launch, pops-weighted over the launches
    util_num = util_den = 0
    for r, ns in zip(recs, n_slots):
        per_slot = np.bincount(r["slot"], weights=r["n_pops"], minlength=1)
        slots = min(ns, len(r))
        util_num += r["n_pops"].sum()
        util_den += slots * per_slot.max()
    return {"plans_per_s": float(done.sum()) / elapsed_per_step, "all_problems_per_s": len(rec) / elapsed_per_step,
            "expansions_per_s": pops / elapsed_per_step, "problems": int(len(rec)), "completed": int(done.sum()),
            "solved_frac": float((rec["status"] == 0).mean()), "iter_limit_frac": float((rec["status"] == 4).mean()),
            "capacity_frac": float((rec["status"] == 5).mean()),
            "slot_utilisation": None if time_sliced else float(util_num / max(util_den, 1)), "ms_per_step": elapsed_per_step * 1e3}


def same_results(ra, pa, rb, pb):
    """Two (records, paths) results of the same problem set are identical: every record field but the slot that ran the
    problem and the diagnostics, and every way-point."""
    ok = all(np.array_equal(ra[f], rb[f]) for f in ra.dtype.names if f not in ("slot", "phase_cycles"))
    return bool(ok and all(np.array_equal(pa[i, :ra["n_final"][i]], pb[i, :ra["n_final"][i]]) for i in range(len(ra))))


def collective_proof(torch, dist, rank, local, world, dev):
    """What RCCL itself saw: every rank's (rank, local device index, device uuid, PCI address, host name, device name) travels
    through ONE all-gather on the nccl group -- the line then shows the backend, the world size the process group reports and
    the number of DISTINCT devices among the ranks (which must equal the world size: a rank per GPU), not just the --gpus
    argument echoed back."""
    import socket
    pr = torch.cuda.get_device_properties(local)
    pci = "%04x:%02x:%02x" % (int(getattr(pr, "pci_domain_id", 0)), int(getattr(pr, "pci_bus_id", 0)), int(getattr(pr, "pci_device_id", 0)))
    ident = "|".join([str(rank), str(local), str(getattr(pr, "uuid", "")), pci, socket.gethostname(), str(pr.name)]).encode()[:256]
    mine = torch.zeros(256, dtype=torch.uint8, device=dev)
    mine[:len(ident)] = torch.tensor(list(ident), dtype=torch.uint8, device=dev)
    allr = torch.empty(world * 256, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(allr, mine)
    rows = [bytes(allr[i * 256:(i + 1) * 256].cpu().numpy().tolist()).rstrip(b"\0").decode() for i in range(world)]
    devices = [dict(zip(("rank", "local_device", "uuid", "pci", "host", "name"), r.split("|"))) for r in rows]
    distinct = len({(d["host"], d["uuid"], d["pci"]) for d in devices})          # (uuid AND PCI address: a runtime that reports one uuid for every device still tells them apart)
    return {"backend": str(dist.get_backend()), "world_size": int(dist.get_world_size()), "devices": devices, "distinct_devices": distinct,
            "ranks_in_order": [int(d["rank"]) for d in devices] == list(range(world)),
            "note": "gathered through all_gather_into_tensor on the process group the bench's collectives use"}


class Bench:
    """What every part of the run needs: the process's place in the job, the default config / vehicle, the timing helpers and the
    problem-set builders. (Until round 5 all of this lived in closures of one 600-line main().)"""

    def __init__(self, a):
        import torch
        import torch.distributed as dist
        from automatedvaletparking_amd import costmap, config
        self.a, self.torch, self.dist = a, torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
        torch.cuda.set_device(self.local)
        self.dev = f"cuda:{self.local}"
        self.use_dist = self.world > 1 or os.environ.get("AVP_BENCH_FORCE_DIST") == "1"     # the env var exercises RCCL with world 1
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local))
        assert self.world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={self.world}"
        self.proof = collective_proof(torch, dist, self.rank, self.local, self.world, self.dev) if self.use_dist else None
        if self.proof is not None:
            assert self.proof["world_size"] == self.world and self.proof["distinct_devices"] == self.world and self.proof["ranks_in_order"], f"RCCL saw {self.proof}"
        self.workload = a.workload if a.workload != "auto" else "c2"
        self.weak = self.use_dist and self.workload in ("c2", "c5")      # N > 1: every rank its own block (weak scaling); else the two-stage deal of one set
        self.cfg = config.default_config()
        self.veh = costmap.Vehicle()
        self.ev_k = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps + a.warmup)]
        self.ev_i = 0

    # ---- problem sets (rank 0 builds and samples; the others receive) ------------------------------------------------
    def checker(self, cap):
        from automatedvaletparking_amd import _native
        return lambda m: _native.DeviceMap(m, self.veh, self.cfg, device=self.local, max_pops=cap).check_batch

    def build(self, name):
        """-> (label, cfg, cap, [(Map, starts, goals)]) (automatedvaletparking_amd/workloads.py holds the definitions)."""
        from automatedvaletparking_amd import workloads
        cfg, world = self.cfg, self.world
        if name in ("c2", "batch4096"):
            n = 256 * (world if self.use_dist else 1) if name == "c2" else 4096
            m, st, go = workloads.case1_pairs(cfg, self.checker(POP_CAP), n)
            label = (("Case1 map, 256 random start/goal pairs (config[1]), pop cap 1000" if n == 256 else
                      "Case1 map, %d x 256 random start/goal pairs (config[1]'s sampler; a block of 256 per rank), pop cap 1000" % world) if name == "c2" else
                     "Case1 map, 4096 random start/goal pairs (north_star target batch), pop cap 1000")
            return label, cfg, POP_CAP, [(m, st, go)]
        if name == "c3":
            maps20 = workloads.case_maps(range(1, 21), cfg, device="cuda")          # batched ingest: one rasteriser launch for the 20 files
            out = [workloads.c3_map_pairs(k, cfg, self.checker(300), 128, m=maps20[k - 1]) for k in range(1, 21)]
            return "all 20 BenchmarkCases x 128 random pairs (config[2]), pop cap 300", cfg, 300, out
        if name == "c5":
            m, c5, starts, goals, _ = workloads.c5_problems(cfg, 1024, device="cuda")
            return "parking lot, 120 obstacles, 1024 starts, RS shot at every pop (config[4]), pop cap 300", c5, 300, [(m, starts, goals)]
        raise ValueError(name)

    def bcast_sets(self, bundle):
        """rank 0's (label, cfg, cap, [(Map, starts, goals)]) on every rank: RCCL broadcasts of the packed costmaps and of
        the problem sets (untimed set-up)."""
        if not self.use_dist:
            return bundle
        from automatedvaletparking_amd import distributed as avd
        torch, dist, rank, dev = self.torch, self.dist, self.rank, self.dev
        lab_, cfg_, cap_, sets_ = bundle if rank == 0 else (None, None, None, None)
        meta = [lab_, cfg_, cap_, len(sets_) if rank == 0 else 0]
        dist.broadcast_object_list(meta, src=0)
        lab_, cfg_, cap_, nsets = meta
        full = []
        for g in range(nsets):
            m = avd.broadcast_map(sets_[g][0] if rank == 0 else None, src=0)         # RCCL broadcast of the packed costmap
            prob = torch.as_tensor(np.concatenate([sets_[g][1], sets_[g][2]], 1), device=dev) if rank == 0 else None
            nprob = torch.tensor([len(sets_[g][1]) if rank == 0 else 0], dtype=torch.int64, device=dev)
            dist.broadcast(nprob, src=0)
            if rank != 0:
                prob = torch.empty((int(nprob.item()), 6), dtype=torch.float64, device=dev)
            dist.broadcast(prob, src=0)                                              # ... and of the problem set
            pr = prob.cpu().numpy()
            full.append((m, pr[:, :3].copy(), pr[:, 3:].copy()))
        return lab_, cfg_, cap_, full

    # ---- timing ------------------------------------------------------------------------------------------------------------
    def timed_steps(self, step_fn, steps, warmup):
        """The contract's timed region: W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize, MAX over ranks."""
        torch, dist = self.torch, self.dist
        for _ in range(warmup):
            step_fn()
        torch.cuda.synchronize()
        if self.use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step_fn()
        torch.cuda.synchronize()
        if self.use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if self.use_dist:
            tt = torch.tensor([el], dtype=torch.float64, device=self.dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, out

    def time_group(self, g, reps=2, **kw):
        """(seconds per pass, last outputs) of one group alone on this GPU."""
        torch = self.torch
        g.plan(**kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            o = g.plan(**kw)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, o

    def events(self):
        """The next pair of HIP events of the headline's timed steps (recorded on torch's current stream = the launch stream)."""
        e = self.ev_k[self.ev_i % len(self.ev_k)]
        self.ev_i += 1
        return e

    def group(self, m, xcfg, st, go, cap, **kw):
        return Group(m, self.veh, xcfg, st, go, self.local, cap, **kw)

    # ---- the multi-GPU steps ---------------------------------------------------------------------------------------------
    def run_weak(self, full, xcfg, xcap, steps, warmup, events=False, **group_kw):
        """Weak-scaling steps over a world x per problem set (one map): this rank plans its contiguous block, one gather of
        records + way-points to rank 0 inside the step. -> (elapsed, [records] and [paths] on rank 0, this rank's Group)."""
        from automatedvaletparking_amd import distributed as avd, path_planner
        m, st, go = full[0]
        per = len(st) // self.world
        blk = slice(self.rank * per, (self.rank + 1) * per)
        g = self.group(m, xcfg, st[blk], go[blk], xcap, **group_kw)

        def step():
            if events:
                e0, e1 = self.events()
                e0.record()
            o = avd.plan_weak(lambda s_, g_: g.plan()[:2], st, go, self.rank, self.world, dst=0)
            if events:
                e1.record()
            return o

        el, (rec_t, path_t) = self.timed_steps(step, steps, warmup)
        if self.rank != 0:
            return el, None, None, g
        return el, [rec_t.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)], [path_t.cpu().numpy()], g

    def run_strong(self, full, xcfg, xcap, steps, warmup, events=False):
        """Strong-scaling steps: every set of `full` split over the ranks by the two-stage deal (way-points to rank 0)."""
        from automatedvaletparking_amd import distributed as avd, path_planner, _native
        planners = []
        for (m, st, go) in full:
            dm = _native.DeviceMap(m, self.veh, xcfg, device=self.local, max_pops=xcap)
            planners.append((dm, path_planner.BatchPlanner(dm, max_nodes=MAX_NODES, max_path=MAX_PATH, mode=path_planner.STAGED, stage_pops=STAGE_POPS), {}))

        def two_stage(k):
            dm, bp1, bp2 = planners[k]
            m, st, go = full[k]

            def stage1(s_l, g_l):
                r, p, _ = bp1.plan_dev(dm.dev_tensor(s_l), dm.dev_tensor(g_l), want_paths=True, first_stage_only=True)
                return r, p

            def stage2(s_l, g_l):
                # every search of the second stage is a long one: the form whose slots hold them all at once
                mode2 = path_planner.long_search_mode(dm, len(s_l))
                if mode2 not in bp2:
                    bp2[mode2] = path_planner.BatchPlanner(dm, max_nodes=MAX_NODES, max_path=MAX_PATH, mode=mode2)
                r, p, _ = bp2[mode2].plan_dev(dm.dev_tensor(s_l), dm.dev_tensor(g_l), want_paths=True)
                return r, p

            return avd.two_stage_plan(stage1, stage2, st, go, self.rank, self.world, paths_to=0)

        def step():
            if events:
                e0, e1 = self.events()
                e0.record()
            o = [two_stage(k) for k in range(len(full))]
            if events:
                e1.record()
            return o

        el, o = self.timed_steps(step, steps, warmup)
        return el, o, planners

    def strong_point(self, full, xcfg, xcap, el, o):
        """rank 0: the sharded result against the whole set planned on this GPU alone (staged call), timed."""
        from automatedvaletparking_amd import distributed as avd, path_planner
        info, ok, t1, slots = {}, True, 0.0, []
        recs_ = [rec_t.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1) for (rec_t, _, _) in o]
        for k, (m, st, go) in enumerate(full):
            ref = self.group(m, xcfg, st, go, xcap, mode=path_planner.STAGED)
            sec, (r_res, r_paths, _) = self.time_group(ref, reps=2)
            t1 += sec
            ok &= same_results(recs_[k], o[k][1].cpu().numpy(), records(r_res, ref.n), r_paths.cpu().numpy())
            slots.append(ref.slots * self.world)
            if k == 0:
                rr = records(r_res, ref.n)
                info["deal_simulation"] = avd.simulate_deals(rr["n_pops"], rr["status"], st, go, STAGE_POPS)
                info["deferred_after_stage1"] = int(len(o[k][2]))
                info["deferred_note"] = "searches still running after the first stage's 16 pops, plus any the wave form cannot hold (same status value)"
            del ref
        assert ok, "sharded result differs from the single-GPU result"
        info.update({"one_gpu_ms_per_step": t1 * 1e3, "speedup_vs_1gpu": t1 / el, "parallel_efficiency": t1 / el / self.world,
                     "one_gpu_note": "the whole set planned by rank 0 alone (staged call) right after the timed steps, same process, same GPU"})
        return info, ok, recs_, slots


# ---- the other two points of the multi-GPU picture (N > 1 / AVP_BENCH_FORCE_DIST), on every rank (collectives inside) -------
def dist_extras(b):
    from automatedvaletparking_amd import path_planner
    dist, rank, world = b.dist, b.rank, b.world
    extra = {}
    b4 = b.bcast_sets(b.build("batch4096") if rank == 0 else None)
    el4, o4, _ = b.run_strong(b4[3], b4[1], b4[2], 2, 1)
    if rank == 0:
        info, ok4, r4, sl4 = b.strong_point(b4[3], b4[1], b4[2], el4 / 2, o4)
        x = summarize(r4, sl4, el4 / 2)
        x.update(info)
        x.update({"workload": b4[0], "scaling": "strong", "shard_invariant": ok4,
                  "note": "the fixed 4 096-problem set split over the ranks by the two-stage deal (records all-gathered after each stage, way-points gathered to rank 0)"})
        extra["strong_scaling_4096"] = x
    dist.barrier()
    # N x 16 384 problems (the 4 096 starts against rotations of the goals), one time-sliced launch per rank, one gather
    if rank == 0:
        mm, st4, go4 = b4[3][0]
        st16 = np.concatenate([st4] * (4 * world))
        go16 = np.concatenate([np.roll(go4, 17 * k, axis=0) for k in range(4 * world)])
        tb = ("Case1 map, %d x 16384 problems (the 4096 starts against rotations of the goals), pop cap 1000" % world, b4[1], b4[2], [(mm, st16, go16)])
    else:
        tb = None
    tb = b.bcast_sets(tb)
    elt, rt, _, gt = b.run_weak(tb[3], tb[1], tb[2], 2, 1)
    if rank == 0:
        x = summarize(rt, [gt.slots * world], elt / 2, time_sliced=bool(gt.bp.last_time_sliced))
        # the N = 1 in-run value of the same per-GPU work: this rank's 16 384 problems alone, no collective
        sec1, o1 = b.time_group(gt, reps=2)
        x1 = summarize([records(o1[0], gt.n)], [gt.slots], sec1, time_sliced=bool(gt.bp.last_time_sliced))
        x.update({"one_gpu_block_ms_without_gather": sec1 * 1e3, "one_gpu_expansions_per_s": x1["expansions_per_s"], "one_gpu_plans_per_s": x1["plans_per_s"],
                  "weak_scaling_efficiency_in_run": x["expansions_per_s"] / (world * x1["expansions_per_s"]) if x1["expansions_per_s"] else None,
                  "one_gpu_note": "rank 0's own 16 384-problem block planned alone right after the timed steps (no gather): the per-GPU rate the %d-rank figure is to be held against" % world})
        x.update({"workload": tb[0], "scaling": "weak", "kernel_form": FORM_NAMES.get(gt.mode), "time_sliced": bool(gt.bp.last_time_sliced), "lookahead": bool(gt.bp.last_lookahead),
                  "note": "every rank plans its own 16 384 problems in one launch (long searches time-sliced), records + way-points gathered to rank 0 at the end of the step"})
        extra["throughput"] = x
    del gt
    dist.barrier()
    return extra


# ---- the headline's roofline object -----------------------------------------------------------------------------------------
def load_pmc():
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        return pj if pj.get("source_hash") == source_hash() else {"stale": True}
    except Exception:
        return None


def headline_roofline(b, recs, kernel_ms, P, pmc):
    """`roofline` of the dominant kernel (plan_kernel). NOT bandwidth shaped (the map lives in LDS): `frac` = the share of SIMD cycles
    with a VALU instruction active, from the stamped PMC passes; SURVEY 8(d)'s algorithmic bytes per launch over the launch time measured
    with HIP events in THIS run are `frac_hbm_algorithmic`; the physical HBM bytes are `traffic`."""
    rec = np.concatenate(recs)
    B_cc = 16 * P + 25
    # algorithmic bytes of the launch(es): U2 per pop (checks + the reference's linear list scans + 680 B)
    # + U3 per heuristic sweep. The list-scan term is what the reference READS; the kernel replaces those scans by an
    # O(1) pose hash and never moves these bytes -- it is reported separately and left out of `frac_without_list_scan`.
    t_checks = float(rec["n_checks"].sum()) * B_cc
    t_scan = 240.0 * float((rec["n_pops"].astype(np.float64) * (rec["n_closed"] + rec["n_open"]) / 2).sum())
    t_pop = 680.0 * float(rec["n_pops"].sum())
    t_h = 16.0 * float(rec["h_cells"].sum())
    scale = (1.0 / b.world) if b.use_dist else 1.0           # per launch on ONE GPU
    alg = (t_checks + t_scan + t_pop + t_h) * scale
    rl = {"kernel": "plan_kernel", "launch_ms": kernel_ms,
          "algorithmic_bytes_per_launch": alg,
          "algorithmic_terms": {"footprint_checks": t_checks * scale, "reference_list_scans": t_scan * scale,
                                "per_pop_state": t_pop * scale, "heuristic_field": t_h * scale},
          "hbm_algorithmic_GBps": alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms else None,
          "frac_hbm_algorithmic": alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if kernel_ms else None,
          "frac_hbm_without_list_scan": (alg - t_scan * scale) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if kernel_ms else None,
          "frac_hbm_note": "SURVEY 8(d)'s algorithmic bytes (what the REFERENCE formulation reads) per launch time: a throughput-equivalence figure. "
                           "The kernel keeps the map in LDS and replaces the list scans by a pose hash; its physical HBM traffic is `traffic`, "
                           "its bound is fp64 VALU issue (`frac`, `fp64_flops_frac`)",
          "hbm_roofline": "n/a (LDS-resident map): `frac` = `frac_valu_busy`, the share of SIMD cycles with a VALU instruction active; the SURVEY 8(d) byte figure is `frac_hbm_algorithmic`",
          "bound": "valu", "unit": "G SIMD-cycles/s", "peak": N_SIMD * CLOCK_GHZ, "achieved": None, "frac": None, "frac_valu_busy": None, "traffic": None}
    if pmc and not pmc.get("stale") and b.workload == "c2" and "plan_kernel" in pmc:
        pk = pmc["plan_kernel"]
        # VALU-busy SIMD cycles per launch (SQ_ACTIVE_INST_VALU counts quad-cycles) over THIS run's launch time
        rl["achieved"] = pk["valu_active_simd_cycles_per_launch"] / (kernel_ms * 1e-3) / 1e9
        rl["frac"] = rl["frac_valu_busy"] = rl["achieved"] / rl["peak"]
        rl["traffic"] = pk.get("hbm_bytes_per_launch_corrected")
        rl["wait_frac"] = pk.get("wait_any_frac")
        if pk.get("f64_valu_wave_insts_per_launch"):
            # upper bound: every counted f64 VALU wave-instruction as 64 live lanes (x 2 flops for an FMA)
            fl = pk["f64_valu_wave_insts_per_launch"]
            rl["fp64_flops_frac"] = 64.0 * (2 * fl.get("fma", 0) + fl.get("mul", 0) + fl.get("add", 0)) / (kernel_ms * 1e-3) / (FP64_PEAK_TFLOPS * 1e12)
            rl["fp64_flops_note"] = "upper bound (64 live lanes per counted wave-instruction) of the fp64 vector peak %.1f TFLOP/s" % FP64_PEAK_TFLOPS
        if pk.get("valu_lane_utilisation") is not None:
            rl["valu_lane_utilisation"] = pk["valu_lane_utilisation"]
        if pk.get("valu_insts_per_launch") is not None:
            rl["valu_wave_insts_per_pop"] = pk["valu_insts_per_launch"] / max(float(rec["n_pops"].sum()) * scale, 1.0)
        rl["pmc_source"] = "profiles/%s (source hash %s)" % (PMC_FILE, pmc["source_hash"])
    elif pmc and pmc.get("stale"):
        rl["pmc_source"] = "profiles/%s is STALE (kernel sources changed): PMC-derived fields left null" % PMC_FILE
    return rl


# ---- extras on the same GPU (N = 1): each one a function of (b, out-so-far) -> the entry it adds ----------------------------
def extra_without_lookahead(b, g0, wcfg, cap, recs, outs):
    """The same step without the expansion lookahead (idle CUs stay idle): what the helpers buy, and that they change no result."""
    gn = b.group(g0.m, wcfg, g0.starts, g0.goals, cap, mode=1, lookahead=False)
    sec, o0 = b.time_group(gn, reps=3)
    x0 = summarize([records(o0[0], g0.n)], [g0.slots], sec)
    x0["lookahead"] = bool(gn.bp.last_lookahead)
    x0["identical_results"] = same_results(recs[0], outs[0][1].cpu().numpy(), records(o0[0], g0.n), o0[1].cpu().numpy())
    return x0


def extra_lookahead_batch_sizes(b):
    """The lookahead on batches LARGER than the chip (workgroup form, config[1]'s sampler, pop cap 1000): helpers exist only in the
    batch's tail, and while they are scarce the owners post only the nodes they pop next (PL_LOOK_TOP_BUSY / PL_LOOK_BACKLOG2). Per size: ms with and
    without, identical results."""
    from automatedvaletparking_amd import workloads
    out = {"workload": "Case1 map, n random start/goal pairs (config[1]'s sampler), pop cap %d, one workgroup per problem" % POP_CAP, "sizes": {}}
    for n in (512, 1024, 1536, 2560):
        m, st, go = workloads.case1_pairs(b.cfg, b.checker(POP_CAP), n)
        e, keep = {}, {}
        for key, look in (("without", False), ("with", None)):
            g = b.group(m, b.cfg, st, go, POP_CAP, mode=1, lookahead=look)
            sec, o = b.time_group(g, reps=2)
            keep[key] = (records(o[0], g.n), o[1].cpu().numpy())
            e["ms_" + key] = sec * 1e3
            e["lookahead_" + key] = bool(g.bp.last_lookahead)
            del g
        e["identical_results"] = same_results(keep["without"][0], keep["without"][1], keep["with"][0], keep["with"][1])
        e["pops"] = int(keep["with"][0]["n_pops"].sum())
        out["sizes"][str(n)] = e
    return out


def extra_workload(b, name):
    """Another BASELINE workload (c3 = config[2], c5 = config[4]) on this GPU."""
    torch = b.torch
    lab, xcfg, xcap, xsets = b.build(name)
    xg = [b.group(m, xcfg, st, go, xcap) for (m, st, go) in xsets]
    plan_groups(xg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        xo = plan_groups(xg)
    torch.cuda.synchronize()
    xe = (time.perf_counter() - t0) / 2
    xs = summarize([records(o[0], g.n) for o, g in zip(xo, xg)], [g.slots for g in xg], xe, time_sliced=any(g.bp.last_time_sliced for g in xg))
    xs["workload"] = lab
    xs["kernel_form"] = FORM_NAMES.get(xg[0].mode)
    xs["lookahead"] = any(bool(g.bp.last_lookahead) for g in xg)
    if len(xg) > 1:
        xs["hip_hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)")
        xs["lookahead_note"] = "%d launches on %d streams share the device: each one's helpers would hold compute units the others' problems wait for (measured 133 vs 83 ms), so the lookahead is off here" % (len(xg), len(xg))
    return xs


def extra_batch4096(b, out):
    """north_star's 4 096-pose batch: every kernel form and the staged call, identical results; the `scale_point`; the saturating
    batches (4 x and 8 x the set, goals re-paired). Adds batch4096, scale_point, saturating_batch."""
    from automatedvaletparking_amd import path_planner, distributed as avd
    lab, xcfg, xcap, xsets = b.build("batch4096")
    mm, st4, go4 = xsets[0]
    forms, ref_rp = {}, None
    for mode in (1, 2, 3, 4, path_planner.STAGED):
        g4 = b.group(mm, xcfg, st4, go4, xcap, mode=mode)
        sec, o4 = b.time_group(g4, reps=2)
        r4, p4 = records(o4[0], g4.n), o4[1].cpu().numpy()
        forms[mode] = summarize([r4], [g4.slots], sec, time_sliced=bool(g4.bp.last_time_sliced))
        forms[mode]["kernel_form"] = FORM_NAMES[mode]
        forms[mode]["time_sliced"] = bool(g4.bp.last_time_sliced)
        forms[mode]["lookahead"] = bool(g4.bp.last_lookahead)
        if ref_rp is None:
            ref_rp = (r4, p4)
        else:
            forms[mode]["identical_to_workgroup_form"] = same_results(ref_rp[0], ref_rp[1], r4, p4)
        del g4
    best = min(forms, key=lambda k: forms[k]["ms_per_step"])
    b4 = dict(forms[best])
    b4["workload"] = lab
    b4["forms_ms_per_step"] = {FORM_NAMES[k]: forms[k]["ms_per_step"] for k in forms}
    b4["forms_identical"] = all(forms[k].get("identical_to_workgroup_form", True) for k in forms)
    out["batch4096"] = b4
    # the point of the 1 -> 8 curve this GPU contributes: the same set, planned the way `--gpus N` plans it
    sp = dict(forms[path_planner.STAGED])
    sp["workload"] = lab
    sp["note"] = "the set `bench.py --gpus N` shards (strong scaling): at N = 1 the two-stage deal is the staged call"
    sp["deal_simulation"] = avd.simulate_deals(ref_rp[0]["n_pops"], ref_rp[0]["status"], st4, go4, STAGE_POPS)
    sp["deal_simulation_note"] = ("predicted load of the busiest rank over the mean, minus 1, from this run's per-problem pop counts: deal by decreasing "
                                  "start-goal distance (round 2) vs the two-stage deal; two_stage_extra_pops_frac = the first-stage pops of the searches planned again")
    out["scale_point"] = sp
    # ---- a saturating batch (4 x the 4 096 set, goals re-paired): the chip's sustained expansion rate per kernel form
    st16 = np.concatenate([st4] * 4)
    go16 = np.concatenate([np.roll(go4, 17 * k, axis=0) for k in range(4)])
    sat = {"workload": "Case1 map, 16384 problems (the 4096 starts against 4 rotations of the goals), pop cap 1000"}
    # (group forms: a workspace slot per problem, long searches time-sliced -- it pays where they outnumber the groups)
    for mode, key, ts in ((1, "workgroup_per_problem", None), (2, "wave_per_problem", None), (3, "pair_per_problem", None), (4, "quad_per_problem", None),
                          (path_planner.STAGED, "staged", None), (2, "wave_per_problem_unsliced", False), (3, "pair_per_problem_unsliced", False)):
        g16 = b.group(mm, xcfg, st16, go16, xcap, mode=mode, time_slice=ts)
        sec, o16 = b.time_group(g16, reps=1)
        sat[key] = summarize([records(o16[0], g16.n)], [g16.slots], sec, time_sliced=bool(g16.bp.last_time_sliced))
        sat[key]["time_sliced"] = bool(g16.bp.last_time_sliced)
        sat[key]["lookahead"] = bool(g16.bp.last_lookahead)
        del g16
    st32, go32 = np.concatenate([st16] * 2), np.concatenate([go16, np.roll(go16, 5, axis=0)])
    for key, ts in (("wave_per_problem", None), ("wave_per_problem_unsliced", False)):
        g32 = b.group(mm, xcfg, st32, go32, xcap, mode=2, time_slice=ts)
        sec, o32 = b.time_group(g32, reps=1)
        sat["n32768_" + key] = summarize([records(o32[0], g32.n)], [g32.slots], sec, time_sliced=bool(g32.bp.last_time_sliced))
        sat["n32768_" + key]["time_sliced"] = bool(g32.bp.last_time_sliced)
        sat["n32768_" + key]["lookahead"] = bool(g32.bp.last_lookahead)
        del g32
    out["saturating_batch"] = sat


def extra_cap_sweep(b):
    """Cap sensitivity: the headline set and config[4] at pop caps 300 / 1000 / 3000 (/ 10 000). Every entry says whether the
    expansion lookahead was on (its record store has a fixed size since round 6: the node arena of a larger cap no longer crowds it out)."""
    sweep = {}
    for wname, caps in (("c2", (300, 1000, 3000)), ("c5", (300, 1000, 3000, 10000))):
        lab_s, scfg, _, ssets = b.build(wname)
        ms_, st_, go_ = ssets[0]
        sweep[wname] = {}
        for cap_s in caps:
            # (the node arena grows with the cap: a search makes up to 10 nodes per pop)
            gs = b.group(ms_, scfg, st_, go_, cap_s, max_nodes=max(MAX_NODES, 12 * cap_s))
            sec, os_ = b.time_group(gs, reps=1)
            rs_ = records(os_[0], gs.n)
            x = summarize([rs_], [gs.slots], sec, time_sliced=bool(gs.bp.last_time_sliced))
            sweep[wname][str(cap_s)] = {k: x[k] for k in ("plans_per_s", "expansions_per_s", "completed", "problems", "iter_limit_frac", "capacity_frac", "ms_per_step")}
            sweep[wname][str(cap_s)].update({"lookahead": bool(gs.bp.last_lookahead), "us_per_pop_of_the_longest_search": sec * 1e6 / max(int(rs_["n_pops"].max()), 1),
                                             "kernel_form": FORM_NAMES.get(gs.mode), "max_nodes": gs.bp.max_nodes})
            del gs
    sweep["note"] = ("completed plans/s is a function of the cap only through the searches the cap stops: on Case1 a fifth of the random pairs never "
                     "connects (the reference would not terminate); on config[4] most searches need thousands of pops")
    return sweep


def extra_cases20(b):
    """The 20 BenchmarkCases' own problems, run to termination (cap 30 000), one problem per launch."""
    from automatedvaletparking_amd import workloads, path_planner
    cfg = b.cfg
    c20 = {}
    for k in range(1, 21):
        mk = workloads.case_map(k, cfg, device="cuda")
        ck = mk.case
        gk = b.group(mk, cfg, np.array([[ck.x0, ck.y0, ck.theta0]]), np.array([[ck.xf, ck.yf, ck.thetaf]]), 30000, mode=1, lookahead=True)
        gk.bp.max_nodes = 1 << 19
        sec, ok_ = b.time_group(gk, reps=1)
        rk = records(ok_[0], 1)[0]
        c20[f"Case{k}"] = {"status": path_planner.STATUS_NAMES.get(int(rk["status"]), int(rk["status"])), "pops": int(rk["n_pops"]),
                           "ms": sec * 1e3, "way_points": int(rk["n_final"]), "reference_s": REFERENCE_SECONDS.get(k), "lookahead": bool(gk.bp.last_lookahead)}
        del gk
    tot = sum(v["ms"] for v in c20.values())
    return {"cases": c20, "total_ms_one_after_the_other": tot, "plans_per_s": 20.0 / (tot * 1e-3),
            "note": "each BenchmarkCase's own start / goal, pop cap 30 000 (never reached), one problem per launch with the expansion lookahead; "
                    "reference_s: the unmodified Python reference on one host thread (BASELINE.md)"}


def extra_single_plan_latency(b):
    """One plan through the reference's API (config[0]): PathPlanner(...).path_planning() on Case1, host call to split path."""
    from automatedvaletparking_amd import workloads, path_planner
    m1 = workloads.case_map(1, b.cfg)
    pl = path_planner.PathPlanner(config=b.cfg, map=m1, vehicle=b.veh)
    pl.path_planning()
    b.torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        pl.path_planning()
    return (time.perf_counter() - t0) / 5 * 1e3


def _time_check(b, dm, t, co, reps=10):
    torch = b.torch
    dm.check_batch_dev(t[0], t[1], t[2], out=co)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dm.check_batch_dev(t[0], t[1], t[2], out=co)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def extra_check_kernel(b, g0, pmc):
    """`roofline_check`: the footprint-collision kernel alone -- 2^20 random poses, the collision-free ones only, and the near misses
    (the free poses with the most obstacle points under the footprint's AABB: every near point goes through the exact test)."""
    torch = b.torch
    dm, m = g0.dm, g0.m
    P = dm.P
    B_cc = 16 * P + 25
    rng = np.random.default_rng(1)
    n_chk = 1 << 20
    cp = np.stack([rng.uniform(m.boundary[0] + 6, m.boundary[1] - 6, n_chk), rng.uniform(m.boundary[2] + 6, m.boundary[3] - 6, n_chk),
                   rng.uniform(-np.pi, np.pi, n_chk)], 0)
    ct = dm.dev_tensor(cp)
    co = dm.empty(n_chk, torch.uint8)
    cms = _time_check(b, dm, ct, co)
    hit_frac = float(co.float().mean().item())
    # the work-heavy case: collision-free poses only (no early exit: every near point of every pose is tested)
    free = cp[:, (co.cpu().numpy() == 0)]
    free = np.ascontiguousarray(np.tile(free, (1, n_chk // max(free.shape[1], 1) + 1))[:, :n_chk])
    fms = _time_check(b, dm, dm.dev_tensor(free), co)
    assert float(co.float().sum().item()) == 0.0
    # near misses: the collision-free poses whose footprint AABB holds the most obstacle points (the reference's "near" set,
    # collision_check.py:55-69): every one of them goes through the exact point test and none ends the pose early
    pp_ = dm.params
    fr = free[:, :min(free.shape[1], 200000)]
    pk_ = m.pack()
    ox_, oy_ = np.asarray(pk_["obs_x"]), np.asarray(pk_["obs_y"])
    near_n = np.zeros(fr.shape[1], np.int32)
    for c0_ in range(0, fr.shape[1], 20000):
        x_, y_, t_ = fr[0, c0_:c0_ + 20000], fr[1, c0_:c0_ + 20000], fr[2, c0_:c0_ + 20000]
        cs_, sn_ = np.cos(t_), np.sin(t_)
        cxs = np.stack([cs_ * lx - sn_ * ly + x_ for lx in (pp_.fp_xr, pp_.fp_xf) for ly in (pp_.fp_yr, pp_.fp_yl)])
        cys = np.stack([sn_ * lx + cs_ * ly + y_ for lx in (pp_.fp_xr, pp_.fp_xf) for ly in (pp_.fp_yr, pp_.fp_yl)])
        inx = (ox_[None, :] >= cxs.min(0)[:, None]) & (ox_[None, :] <= cxs.max(0)[:, None])
        iny = (oy_[None, :] >= cys.min(0)[:, None]) & (oy_[None, :] <= cys.max(0)[:, None])
        near_n[c0_:c0_ + 20000] = (inx & iny).sum(1)
    thr_n = max(15, int(np.quantile(near_n, 0.98)))          # the 2 % of the free poses with the most near points
    nm = fr[:, near_n >= thr_n]
    near_miss = None
    if nm.shape[1] >= 1000:
        mean_near = float(near_n[near_n >= thr_n].mean())
        nm = np.ascontiguousarray(np.tile(nm, (1, n_chk // nm.shape[1] + 1))[:, :n_chk])
        nms = _time_check(b, dm, dm.dev_tensor(nm), co)
        assert float(co.float().sum().item()) == 0.0
        near_miss = {"checks_per_s": n_chk / (nms * 1e-3), "launch_ms": nms, "mean_near_points_per_pose": mean_near,
                     "point_tests_per_s": n_chk * mean_near / (nms * 1e-3),
                     "min_near_points": thr_n,
                     "note": "the 2 % of the random set's collision-free poses with the most obstacle points under the footprint's AABB (collision_check.py:55-69's near set): the point test runs on every one of them and no pose ends early"}
    rc = {"kernel": "check_distance_kernel", "launch_ms": cms, "checks_per_s": n_chk / (cms * 1e-3), "bytes_per_check_reference": B_cc,
          "colliding_frac": hit_frac,
          "free_poses_only": {"checks_per_s": n_chk / (fms * 1e-3), "launch_ms": fms,
                              "note": "the same kernel on collision-free poses only (the random set's free poses, repeated): no early exit, but few near points (free poses are far from obstacles)"},
          "near_miss_poses": near_miss,
          "hbm_algorithmic_GBps": n_chk * B_cc / (cms * 1e-3) / 1e9,
          "note": "the reference formulation reads every obstacle point per check (16P+25 B); the kernel keeps the map in LDS and moves 25 B/check of HBM, so its bound is VALU/LDS issue, not HBM",
          "hbm_traffic_GBps": n_chk * 25 / (cms * 1e-3) / 1e9, "frac_hbm_traffic": n_chk * 25 / (cms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
          "hbm_roofline": "n/a (LDS-resident map)",
          "bound": "valu", "unit": "G SIMD-cycles/s", "peak": N_SIMD * CLOCK_GHZ, "achieved": None, "frac": None, "frac_valu_busy": None, "traffic": None}
    if pmc and not pmc.get("stale") and "check_distance_kernel" in pmc:
        ck = pmc["check_distance_kernel"]
        rc["achieved"] = ck["valu_active_simd_cycles_per_launch"] / (cms * 1e-3) / 1e9
        rc["frac"] = rc["frac_valu_busy"] = rc["achieved"] / rc["peak"]
        rc["traffic"] = ck.get("hbm_bytes_per_launch_corrected")
        rc["lds_busy_frac"] = ck.get("lds_busy_frac")
        rc["lds_bank_conflict_frac"] = ck.get("lds_bank_conflict_frac")
        if ck.get("f64_valu_wave_insts_per_launch"):
            fl = ck["f64_valu_wave_insts_per_launch"]
            rc["fp64_flops_frac"] = 64.0 * (2 * fl.get("fma", 0) + fl.get("mul", 0) + fl.get("add", 0)) / (cms * 1e-3) / (FP64_PEAK_TFLOPS * 1e12)
        if near_miss and ck.get("valu_lane_utilisation") is not None:
            # the yardstick of this kernel is point tests per second on near misses; its ceiling is VALU issue: at the measured VALU-busy share
            # and lane utilisation a fully busy, fully live chip would do point_tests_per_s / (frac x lane utilisation)
            near_miss["valu_ceiling_point_tests_per_s"] = near_miss["point_tests_per_s"] / max(rc["frac"] * ck["valu_lane_utilisation"], 1e-9)
            near_miss["valu_ceiling_note"] = "point_tests_per_s / (VALU-busy share x live-lane share of the random-pose PMC pass): what the same instruction stream would do on a chip whose every SIMD cycle issued a full-wave VALU instruction"
    return rc


def extra_cpu_baseline(b, g0, wcfg, cap, head, out):
    """`cpu_baseline` (one core) and `cpu_baseline_all_cores` (pthreads, steady state): the oracle's C restatement on this box's host."""
    from oracle import oracle
    o = oracle.Oracle(g0.m, b.veh, wcfg, max_pops=cap)
    nb = min(g0.n, 256)
    t1 = time.perf_counter()
    pops_cpu = done_cpu = n_cpu = passes_cpu = 0
    while time.perf_counter() - t1 < 10.0:        # whole passes over the set until ~10 s of one core are spent
        for s_, g_ in zip(g0.starts[:nb], g0.goals[:nb]):
            w = o.plan(s_, g_, max_trace=1)
            pops_cpu += w["n_pops"]
            done_cpu += w["status"] in (0, 1)
            n_cpu += 1
        passes_cpu += 1
    tc = time.perf_counter() - t1
    out["cpu_baseline"] = {"value": done_cpu / tc, "unit": "plans/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
                           "sample": f"the first {nb} problems of the headline workload, {passes_cpu} passes, pop cap {cap}, C restatement (oracle/avp_oracle.c, glibc libm), {tc:.1f} s; completed searches only, like `value`",
                           "all_problems_per_s": n_cpu / tc, "expansions_per_s": pops_cpu / tc}
    # all host cores, steady state, NO Python in the loop: orc_plan_batch (oracle/avp_oracle.c) -- pthreads, one problem
    # per thread from an atomic ticket counter cycling over the same problems (shuffled once); in-flight plans are
    # finished and counted, the clock stops when the last one ends. The same loop on ONE thread gives the scaling.
    ncore, nquota = os.cpu_count() or 1, cpu_quota()
    order = np.random.default_rng(0).permutation(nb).astype(np.int32)
    one = o.plan_batch(g0.starts[:nb], g0.goals[:nb], threads=1, min_seconds=5.0, order=order)
    per_thread = {}
    # (the container's CPU quota is what "all cores" means here: thread counts around it, and the host's full count for the record)
    for nt, secs in ((nquota, 8.0), (min(ncore, 2 * nquota), 5.0), (max(1, nquota // 2), 4.0), (ncore, 4.0)):
        if nt in per_thread:
            continue
        bb = o.plan_batch(g0.starts[:nb], g0.goals[:nb], threads=nt, min_seconds=secs, order=order)
        per_thread[nt] = {"plans_per_s": bb["completed"] / bb["seconds"], "all_problems_per_s": bb["plans"] / bb["seconds"],
                          "expansions_per_s": bb["pops"] / bb["seconds"], "seconds": bb["seconds"], "plans": bb["plans"]}
    best_nt = max(per_thread, key=lambda k: per_thread[k]["expansions_per_s"])
    bt = per_thread[best_nt]
    one_exp = one["pops"] / one["seconds"]
    out["cpu_baseline_all_cores"] = {"value": bt["plans_per_s"], "unit": "plans/s", "cores": best_nt, "kind": "port", "cpu_model": cpu_model(),
                                     "all_problems_per_s": bt["all_problems_per_s"], "expansions_per_s": bt["expansions_per_s"],
                                     "one_thread_same_loop": {"plans_per_s": one["completed"] / one["seconds"], "expansions_per_s": one_exp, "seconds": one["seconds"]},
                                     "scaling_vs_1core": bt["expansions_per_s"] / one_exp if one_exp else None,
                                     "by_thread_count": {str(k): v for k, v in per_thread.items()},
                                     "gpu_over_cpu_all_cores_expansions": head["expansions_per_s"] / bt["expansions_per_s"] if bt["expansions_per_s"] else None,
                                     "cpu_quota": nquota, "host_hardware_threads": ncore,
                                     "sample": f"the same {nb} problems, shuffled once and cycled by an atomic ticket counter over {best_nt} pthreads (orc_plan_batch, no Python in the loop) for {bt['seconds']:.1f} s ({bt['plans']} plans): steady state; this process may use {nquota} CPUs (cgroup cpu.max / affinity) of the host's {ncore} hardware threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["auto", "c2", "batch4096", "c3", "c5"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only")
    ap.add_argument("--pmc-mode", action="store_true", help="headline + the footprint kernel only (the rocprofv3 PMC passes: per-launch counters)")
    a = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result: everything else that writes to file descriptor 1 (RCCL's version
    # banner, library chatter) is sent to stderr for the lifetime of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    b = Bench(a)
    from automatedvaletparking_amd import path_planner
    torch, dist, rank, world, use_dist, weak, workload = b.torch, b.dist, b.rank, b.world, b.use_dist, b.weak, b.workload
    extra_dist = {}

    # ---- headline -----------------------------------------------------------------------------------------------------
    label, wcfg, cap, groups_full = b.bcast_sets(b.build(workload) if rank == 0 else None)
    outs = None
    if not use_dist:
        groups = [b.group(m, wcfg, st, go, cap) for (m, st, go) in groups_full]

        def step():
            e0, e1 = b.events()
            e0.record()
            o = plan_groups(groups)
            e1.record()
            return o

        elapsed, outs = b.timed_steps(step, a.steps, a.warmup)
        recs = [records(o[0], g.n) for o, g in zip(outs, groups)]
        shard_invariant = None
        slots_all = [g.slots for g in groups]
    elif weak:
        # ---- weak scaling: every rank its own block of the world x per set, one gather to rank 0 per step ----------------
        elapsed, recs, paths_w, gw = b.run_weak(groups_full, wcfg, cap, a.steps, a.warmup, events=True)
        groups = [gw]
        shard_invariant = None
        slots_all = [gw.slots * world]
        if rank == 0:
            m, st, go = groups_full[0]
            # this rank's block alone, without the gather (the N = 1 time of the same per-GPU work), and the whole set on this
            # GPU alone: the gathered result must be identical
            sec_blk, _ = b.time_group(gw, reps=3)
            ref = b.group(m, wcfg, st, go, cap)
            sec_all, (r_res, r_paths, _) = b.time_group(ref, reps=1)
            shard_invariant = same_results(recs[0], paths_w[0], records(r_res, ref.n), r_paths.cpu().numpy())
            assert shard_invariant, "sharded result differs from the single-GPU result"
            extra_dist.update({"block_ms_without_gather": sec_blk * 1e3, "weak_scaling_efficiency_in_run": sec_blk / (elapsed / a.steps),
                               "weak_scaling_note": "one rank's 256-problem block planned alone, no collective, over the time of a step with %d ranks and the gather" % world,
                               "whole_set_on_one_gpu_ms": sec_all * 1e3})
            del ref
        dist.barrier()
    else:
        elapsed, out_s, planners = b.run_strong(groups_full, wcfg, cap, a.steps, a.warmup, events=True)
        recs, shard_invariant, slots_all = [], None, []
        if rank == 0:
            info, shard_invariant, recs, slots_all = b.strong_point(groups_full, wcfg, cap, elapsed / a.steps, out_s)
            extra_dist.update(info)
        dist.barrier()

        class _G:          # what the report below reads of a group
            pass
        groups = []
        for k, (m, st, go) in enumerate(groups_full):
            g = _G()
            g.dm, g.m, g.mode, g.bp, g.n = planners[k][0], m, path_planner.STAGED, planners[k][1], len(st)
            groups.append(g)

    if use_dist and weak and not a.no_extras:
        extra_dist.update(dist_extras(b))

    if rank == 0:
        head = summarize(recs, slots_all, elapsed / a.steps, time_sliced=any(g.bp.last_time_sliced for g in groups))
        kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in b.ev_k[a.warmup:a.warmup + a.steps]])) if a.steps else 0.0
        P = groups[0].dm.P
        pmc = load_pmc()
        out = {
            "metric": "hybrid-A* plans/sec, batched poses (completed searches at pop cap %d; node expansions/sec in expansions_per_s)" % cap, "value": head["plans_per_s"], "unit": "plans/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if (use_dist and not weak) else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": label, "problems": head["problems"], "pop_cap": cap, "obstacle_points": P,
                       "kernel_form": FORM_NAMES.get(groups[0].mode, str(groups[0].mode)) if (not use_dist or weak) else
                       "two-stage deal: wave form for %d pops on the index slice, the rest dealt evenly (library's choice of form)" % STAGE_POPS,
                       "expansion_lookahead": bool(groups[0].bp.last_lookahead),
                       "parallelism": f"shard{world}" + ((" (every rank plans its own block; records + way-points gathered to rank 0 inside the timed step)" if weak else
                                                         " (records all-gathered after each stage, way-points gathered to rank 0, inside the timed step)") if use_dist else "")},
            "value_counts": "completed searches (status OK or NO_PATH); ITER_LIMIT problems are excluded",
            "all_problems_per_s": head["all_problems_per_s"], "expansions_per_s": head["expansions_per_s"],
            "solved_frac": head["solved_frac"], "iter_limit_frac": head["iter_limit_frac"],
            "slot_utilisation": head["slot_utilisation"], "shard_invariant": shard_invariant,
            "roofline": headline_roofline(b, recs, kernel_ms, P, pmc),
        }
        out.update(extra_dist)
        if b.proof is not None:
            out["collective"] = b.proof
        if "strong_scaling_4096" in extra_dist:
            # the strong-scaling point of the fixed 4 096 set beside the weak headline, at the top level (comparable with earlier rounds' multi-GPU lines)
            out["speedup_vs_1gpu"] = extra_dist["strong_scaling_4096"].get("speedup_vs_1gpu")
            out["parallel_efficiency"] = extra_dist["strong_scaling_4096"].get("parallel_efficiency")
            out["speedup_note"] = "strong scaling of the fixed 4096-problem set (strong_scaling_4096); the headline `value` is the weak-scaling step"

        single = world == 1 and not use_dist and not a.no_extras
        if single and not a.pmc_mode and groups[0].bp.last_lookahead:
            out["without_lookahead"] = extra_without_lookahead(b, groups[0], wcfg, cap, recs, outs)
        if single:
            if not a.pmc_mode:
                for name in ("c3", "c5"):
                    if name != workload:
                        out[name] = extra_workload(b, name)
                extra_batch4096(b, out)
                out["cap_sweep"] = extra_cap_sweep(b)
                out["lookahead_batch_sizes"] = extra_lookahead_batch_sizes(b)
                out["cases20"] = extra_cases20(b)
                out["single_plan_latency_ms"] = extra_single_plan_latency(b)
                out["single_plan_note"] = "PathPlanner.path_planning() on BenchmarkCases/Case1.csv, host call to split path (uploads, launch, download, split_path), mean of 5; reference: 52 s"
            out["roofline_check"] = extra_check_kernel(b, groups[0], pmc)
            if not a.no_cpu_baseline and not a.pmc_mode:
                extra_cpu_baseline(b, groups[0], wcfg, cap, head, out)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
