#!/usr/bin/env python3
"""bench.py -- hybrid-A* plans/s on batched poses (BASELINE.json metric), one process per GPU.

A "step" = one pass of the hot path over one batch of problems, inputs resident in HBM.

  N = 1 : headline = BASELINE config[1] (Case1 map, 256 random start/goal pairs, pop cap 1000); extras on the same GPU:
          `batch4096` (north_star's 4 096-pose target workload), `c3` (20 BenchmarkCases x 128), `c5` (dense clutter, RS
          shot at every pop), the footprint kernel alone, and the CPU port timed on the host.
  N > 1 : SURVEY 8(e), strong scaling of ONE fixed 4 096-problem set on the Case1 map: rank 0 samples the problems,
          the map blob and the problems are broadcast (RCCL, untimed set-up), every rank plans the shard that
          `shard_indices` deals it, and the timed step ends with the all-gather of the result records AND the solved
          paths (fixed stride max_path x 4 doubles) -- the only data-path collectives. Rank 0 then checks the gathered
          result against its own single-GPU run of the whole set (shard invariance).
          AVP_BENCH_FORCE_DIST=1 runs exactly this code path with world size 1 (through RCCL).
  --workload c3 with N > 1: every rank holds all 20 maps, each map's 128 problems are dealt the same way.

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
sys.path.insert(0, ROOT)

POP_CAP = 1000
MAX_NODES = 16384
MAX_PATH = 256
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
N_SIMD = 256 * 4                # SIMDs on the chip
CLOCK_GHZ = 2.4                 # max shader clock
CASES = os.path.join(ROOT, "data", "BenchmarkCases")


def source_hash():
    """sha256 over the kernel sources: stamps PMC summaries so that a stale one is refused (profiles/README.md)."""
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, "automatedvaletparking_amd", "csrc"), os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".h", ".hip", ".inc")):
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


class Group:
    """One map + its problems on this rank's GPU."""

    def __init__(self, m, veh, cfg, starts, goals, local, cap=POP_CAP, mode=0):
        import ctypes as C
        from automatedvaletparking_amd import _native, path_planner
        self.m = m
        self.dm = _native.DeviceMap(m, veh, cfg, device=local, max_pops=cap)
        self.bp = path_planner.BatchPlanner(self.dm, max_nodes=MAX_NODES, max_path=MAX_PATH, mode=mode)
        self.set_problems(starts, goals)
        L = _native.lib()
        self.mode = int(L.avp_plan_pick_mode(self.dm.h, C.c_int64(self.n), C.c_int32(mode)))      # 1 workgroup / 2 wave per problem
        self.slots = int(L.avp_plan_slots(self.dm.h, C.c_int32(self.mode)))

    def set_problems(self, starts, goals):
        self.starts, self.goals = np.ascontiguousarray(starts), np.ascontiguousarray(goals)
        self.n = len(starts)
        self.st_t, self.go_t = self.dm.dev_tensor(self.starts), self.dm.dev_tensor(self.goals)


def plan_groups(groups):
    """One pass over all groups. Several maps (C3: 20 BenchmarkCases x 128 problems) are independent launches of half a
    chip each: every group gets its own HIP stream so that the launches overlap; the caller's stream waits for all."""
    import torch
    if len(groups) == 1:
        g = groups[0]
        return [g.bp.plan_dev(g.st_t, g.go_t, want_paths=True)]
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
            outs.append(g.bp.plan_dev(g.st_t, g.go_t, want_paths=True))
    for g in groups:
        cur.wait_stream(g.stream)
    return outs


def sample_pairs(m, dm, n_pairs, rng):
    """SURVEY 8(d) sampler: footprint-free (HIP check kernel) poses outside every obstacle polygon, paired up."""
    from automatedvaletparking_amd import sampling
    free = []
    while len(free) < 2 * n_pairs:
        cand = sampling.sample_free_poses(m.boundary, m.case.obs, 8 * min(n_pairs, 512), rng, margin=6.0, reject=False)
        hit = dm.check_batch(cand)
        free += [p for p, h in zip(cand, hit) if not h and sampling.pose_is_free(p[0], p[1], p[2], m.case.obs)]
    poses = np.array(free[:2 * n_pairs])
    return poses[0::2], poses[1::2]


def records(res_t, n):
    from automatedvaletparking_amd import path_planner
    return res_t.cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)[:n]


def summarize(recs, n_slots, elapsed_per_step):
    """Throughput figures of one step over the given record arrays (one per group). n_slots: problem slots of the kernel
    form that ran (CUs for one workgroup per problem, 8 x CUs for one wave per problem), or a list, one per group."""
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
            "slot_utilisation": float(util_num / max(util_den, 1)), "ms_per_step": elapsed_per_step * 1e3}


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

    import torch
    import torch.distributed as dist
    from automatedvaletparking_amd import costmap, config, sampling, _native, path_planner, distributed as avd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    use_dist = world > 1 or os.environ.get("AVP_BENCH_FORCE_DIST") == "1"     # the env var exercises RCCL with world 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    workload = a.workload if a.workload != "auto" else ("batch4096" if use_dist else "c2")

    cfg = config.default_config()
    veh = costmap.Vehicle()

    # ---- problem sets (rank 0 builds and samples; the others receive) ------------------------------------------------
    def build(name):
        """-> (label, cfg, cap, [(Map, starts, goals)]) on rank 0; maps None elsewhere."""
        if name in ("c2", "batch4096"):
            n = 256 if name == "c2" else 4096
            m = costmap.Map(file=os.path.join(CASES, "Case1.csv"), discrete_size=cfg["map_discrete_size"])
            dm = _native.DeviceMap(m, veh, cfg, device=local, max_pops=POP_CAP)
            st, go = sample_pairs(m, dm, n, np.random.default_rng(20260927))
            label = ("Case1 map, 256 random start/goal pairs (config[1]), pop cap 1000" if name == "c2" else
                     "Case1 map, 4096 random start/goal pairs (north_star target batch), pop cap 1000")
            return label, cfg, POP_CAP, [(m, st, go)]
        if name == "c3":
            out = []
            for k in range(1, 21):
                m = costmap.Map(file=os.path.join(CASES, f"Case{k}.csv"), discrete_size=cfg["map_discrete_size"], device="cuda")
                dm = _native.DeviceMap(m, veh, cfg, device=local, max_pops=300)
                st, go = sample_pairs(m, dm, 128, np.random.default_rng(20260927 + k))
                out.append((m, st, go))
            return "all 20 BenchmarkCases x 128 random pairs (config[2]), pop cap 300", cfg, 300, out
        if name == "c5":
            import tempfile
            obs, goal, aisle = sampling.parking_lot_map()
            with tempfile.TemporaryDirectory() as td:
                pth = os.path.join(td, "c5.csv")
                sampling.write_tpcap_csv(pth, (aisle[0] + 8.0, 0.5 * (aisle[2] + aisle[3]), 0.0), goal, obs)
                m = costmap.Map(file=pth, discrete_size=cfg["map_discrete_size"], device="cuda")
            c5 = dict(cfg)
            c5["flag_radius"] = 1e9
            rng = np.random.default_rng(5)
            starts = np.stack([rng.uniform(m.boundary[0] + 4, m.boundary[1] - 4, 1024), rng.uniform(aisle[2] + 1.2, aisle[3] - 1.2, 1024),
                               rng.choice([0.0, np.pi], 1024) + rng.normal(0, 0.05, 1024)], 1)
            return "parking lot, 120 obstacles, 1024 starts, RS shot at every pop (config[4]), pop cap 300", c5, 300, [(m, starts, np.tile(np.array(goal), (1024, 1)))]
        raise ValueError(name)

    def timed_steps(step_fn, steps, warmup):
        for _ in range(warmup):
            step_fn()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step_fn()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, out

    # ---- headline -----------------------------------------------------------------------------------------------------
    label, wcfg, cap, sets = build(workload) if rank == 0 else (None, None, None, None)
    if use_dist:
        meta = [label, wcfg, cap, len(sets) if rank == 0 else 0]
        dist.broadcast_object_list(meta, src=0)
        label, wcfg, cap, nsets = meta
        groups_full = []
        for g in range(nsets):
            m = avd.broadcast_map(sets[g][0] if rank == 0 else None, src=0)          # RCCL broadcast of the packed costmap
            prob = torch.as_tensor(np.concatenate([sets[g][1], sets[g][2]], 1), device=dev) if rank == 0 else None
            nprob = torch.tensor([len(sets[g][1]) if rank == 0 else 0], dtype=torch.int64, device=dev)
            dist.broadcast(nprob, src=0)
            if rank != 0:
                prob = torch.empty((int(nprob.item()), 6), dtype=torch.float64, device=dev)
            dist.broadcast(prob, src=0)                                               # ... and of the problem set
            pr = prob.cpu().numpy()
            groups_full.append((m, pr[:, :3].copy(), pr[:, 3:].copy()))
    else:
        groups_full = sets

    ev_k = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps + a.warmup)]
    ev_i = [0]

    if not use_dist:
        groups = [Group(m, veh, wcfg, st, go, local, cap) for (m, st, go) in groups_full]

        def step():
            outs = []
            e0, e1 = ev_k[ev_i[0] % len(ev_k)]
            ev_i[0] += 1
            e0.record()
            outs = plan_groups(groups)
            e1.record()
            return outs

        elapsed, outs = timed_steps(step, a.steps, a.warmup)
        recs = [records(o[0], g.n) for o, g in zip(outs, groups)]
        shard_invariant = None
        for g in groups:
            g.kernel_form = g.mode
    else:
        # shard: problems dealt by decreasing start-goal distance; equal shard size (padded with start == goal problems)
        groups, idxs, pers = [], [], []
        for (m, st, go) in groups_full:
            s_l, g_l, idx_pad, per = avd.shard_problems(st, go, rank, world)
            groups.append(Group(m, veh, wcfg, s_l, g_l, local, cap))
            idxs.append(idx_pad)
            pers.append(per)
        rec_stride = path_planner.RESULT_DTYPE.itemsize
        gat_r = [torch.empty((world, per, rec_stride), dtype=torch.uint8, device=dev) for per in pers]
        gat_p = [torch.empty((world, per, MAX_PATH, 4), dtype=torch.float64, device=dev) for per in pers]
        idx_t = [avd.all_gather_rows(torch.as_tensor(ix, device=dev)).cpu().numpy().reshape(-1) for ix in idxs]

        def step():
            e0, e1 = ev_k[ev_i[0] % len(ev_k)]
            ev_i[0] += 1
            e0.record()
            for k, g in enumerate(groups):
                res, paths, _ = g.bp.plan_dev(g.st_t, g.go_t, want_paths=True)
                if k == len(groups) - 1:
                    e1.record()
                avd.all_gather_rows(res, out=gat_r[k])       # final gather of the records ...
                avd.all_gather_rows(paths, out=gat_p[k])     # ... and of the solved paths (RCCL all-gather over xGMI)
            return None

        elapsed, _ = timed_steps(step, a.steps, a.warmup)
        recs, shard_invariant = [], None
        slots_per_gpu = int(_native.lib().avp_plan_default_slots(groups[0].dm.h))
        if rank == 0:
            shard_invariant = True
            for k, (m, st, go) in enumerate(groups_full):
                flat_r = gat_r[k].reshape(-1, rec_stride).cpu().numpy().view(path_planner.RESULT_DTYPE).reshape(-1)
                flat_p = gat_p[k].reshape(-1, MAX_PATH, 4).cpu().numpy()
                flat_r = flat_r.copy()
                flat_r["slot"] += (np.arange(len(flat_r)) // pers[k]).astype(np.int32) * slots_per_gpu     # slot ids are per GPU
                rec_all, path_all = avd.unshard_rows(idx_t[k], flat_r, flat_p)
                recs.append(rec_all)
                # the same set on this GPU alone (untimed): the sharded result must be identical
                ref = Group(m, veh, wcfg, st, go, local, cap)
                r_res, r_paths, _ = ref.bp.plan_dev(ref.st_t, ref.go_t, want_paths=True)
                r_rec, r_paths = records(r_res, ref.n), r_paths.cpu().numpy()
                for name in ("status", "n_pops", "n_astar", "n_final", "n_checks", "n_rs", "n_closed", "n_open", "rs_L"):
                    shard_invariant &= bool(np.array_equal(rec_all[name], r_rec[name]))
                for i in range(ref.n):
                    nf = int(r_rec["n_final"][i])
                    shard_invariant &= bool(np.array_equal(path_all[i, :nf], r_paths[i, :nf]))
            assert shard_invariant, "sharded result differs from the single-GPU result"

    if rank == 0:
        n_slots = int(_native.lib().avp_plan_default_slots(groups[0].dm.h))
        head = summarize(recs, [g.slots * (world if use_dist else 1) for g in groups], elapsed / a.steps)
        kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev_k[a.warmup:a.warmup + a.steps]])) if a.steps else 0.0
        rec = np.concatenate(recs)
        P = groups[0].dm.P
        B_cc = 16 * P + 25
        # algorithmic bytes of the launch(es): U2 per pop (checks + the reference's linear list scans + 680 B)
        # + U3 per heuristic sweep. The list-scan term is what the reference READS; the kernel replaces those scans by an
        # O(1) pose hash and never moves these bytes -- it is reported separately and left out of `frac_without_list_scan`.
        t_checks = float(rec["n_checks"].sum()) * B_cc
        t_scan = 240.0 * float((rec["n_pops"].astype(np.float64) * (rec["n_closed"] + rec["n_open"]) / 2).sum())
        t_pop = 680.0 * float(rec["n_pops"].sum())
        t_h = 16.0 * float(rec["h_cells"].sum())
        scale = (1.0 / world) if use_dist else 1.0           # per launch on ONE GPU
        alg = (t_checks + t_scan + t_pop + t_h) * scale
        pmc = None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_summary.json")))
            pmc = pj if pj.get("source_hash") == source_hash() else {"stale": True}
        except Exception:
            pass
        rl = {"kernel": "plan_kernel", "launch_ms": kernel_ms,
              "algorithmic_bytes_per_launch": alg,
              "algorithmic_terms": {"footprint_checks": t_checks * scale, "reference_list_scans": t_scan * scale,
                                    "per_pop_state": t_pop * scale, "heuristic_field": t_h * scale},
              "hbm_algorithmic_GBps": alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms else None,
              "frac_hbm_algorithmic": alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if kernel_ms else None,
              "frac_hbm_without_list_scan": (alg - t_scan * scale) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if kernel_ms else None,
              "bound": "valu", "unit": "G SIMD-cycles/s", "peak": N_SIMD * CLOCK_GHZ, "achieved": None, "frac": None, "traffic": None}
        if pmc and not pmc.get("stale") and workload == "c2" and "plan_kernel" in pmc:
            pk = pmc["plan_kernel"]
            # VALU-busy SIMD cycles per launch (SQ_ACTIVE_INST_VALU counts quad-cycles) over THIS run's launch time
            rl["achieved"] = pk["valu_active_simd_cycles_per_launch"] / (kernel_ms * 1e-3) / 1e9
            rl["frac"] = rl["achieved"] / rl["peak"]
            rl["traffic"] = pk.get("hbm_bytes_per_launch_corrected")
            rl["wait_frac"] = pk.get("wait_any_frac")
            try:      # per-pop critical path of the capped problems, from the instrumented kernel (scripts/variant_bench.py)
                rl["cycles_per_pop"] = json.load(open(os.path.join(ROOT, "profiles", "r02_plan_kernel_phase_cycles.json")))["cyc_per_pop"]
                rl["cycles_per_pop_note"] = "the long way (no expansion record), instrumented kernel without lookahead"
            except Exception:
                rl["cycles_per_pop"] = None
            try:      # ... and with the expansion lookahead (scripts/look_bench.py): record pops of the capped searches
                lk = json.load(open(os.path.join(ROOT, "profiles", "r02_lookahead.json")))
                if lk.get("source_hash") == source_hash():
                    rp = lk["record_pops_of_capped_problems"]
                    rl["record_pop_frac"] = rp["record_pop_frac"]
                    rl["cycles_per_record_pop"] = max(rp["cycles_since_pop_start_per_wave"]["end_of_pop"])
                    # average over record pops and long pops: the long-way figure scaled by the measured launch-time ratio
                    if rl["cycles_per_pop"]:
                        rl["cycles_per_pop_with_lookahead"] = rl["cycles_per_pop"] * lk["with_lookahead"]["ms_best"] / lk["without_lookahead"]["ms_best"]
            except Exception:
                pass
            rl["pmc_source"] = "profiles/r02_pmc_summary.json (source hash %s)" % pmc["source_hash"]
        elif pmc and pmc.get("stale"):
            rl["pmc_source"] = "profiles/r02_pmc_summary.json is STALE (kernel sources changed): PMC-derived fields left null"
        out = {
            "metric": "hybrid-A* plans/sec, batched poses", "value": head["plans_per_s"], "unit": "plans/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if use_dist else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": label, "problems": head["problems"], "pop_cap": cap, "obstacle_points": P,
                       "kernel_form": "one workgroup per problem" if groups[0].mode == 1 else "one wave per problem",
                       "expansion_lookahead": bool(groups[0].bp.last_lookahead),
                       "parallelism": f"shard{world}" + (" (records + paths all-gathered in the timed step)" if use_dist else "")},
            "value_counts": "completed searches (status OK or NO_PATH); ITER_LIMIT problems are excluded",
            "all_problems_per_s": head["all_problems_per_s"], "expansions_per_s": head["expansions_per_s"],
            "solved_frac": head["solved_frac"], "iter_limit_frac": head["iter_limit_frac"],
            "slot_utilisation": head["slot_utilisation"], "shard_invariant": shard_invariant,
            "roofline": rl,
        }

        if world == 1 and not use_dist and not a.no_extras and not a.pmc_mode and groups[0].bp.last_lookahead:
            # ---- the same step without the expansion lookahead (idle CUs stay idle): what the helpers buy -----------------
            g0 = groups[0]
            bp0 = path_planner.BatchPlanner(g0.dm, max_nodes=MAX_NODES, max_path=MAX_PATH, mode=1, lookahead=False)
            o0 = bp0.plan_dev(g0.st_t, g0.go_t, want_paths=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                o0 = bp0.plan_dev(g0.st_t, g0.go_t, want_paths=True)
            torch.cuda.synchronize()
            x0 = summarize([records(o0[0], g0.n)], [g0.slots], (time.perf_counter() - t0) / 3)
            r_on, r_off = recs[0], records(o0[0], g0.n)
            p_on, p_off = outs[0][1].cpu().numpy(), o0[1].cpu().numpy()
            x0["identical_results"] = bool(all(np.array_equal(r_on[f], r_off[f]) for f in r_on.dtype.names if f not in ("slot", "phase_cycles"))
                                           and all(np.array_equal(p_on[i, :r_on["n_final"][i]], p_off[i, :r_on["n_final"][i]]) for i in range(g0.n)))
            out["without_lookahead"] = x0
            del bp0
        if world == 1 and not use_dist and not a.no_extras:
            # ---- extras on the same GPU: the other BASELINE workloads ------------------------------------------------
            for name in ("batch4096", "c3", "c5"):
                if name == workload or a.pmc_mode:
                    continue
                lab, xcfg, xcap, xsets = build(name)
                xg = [Group(m, veh, xcfg, st, go, local, xcap) for (m, st, go) in xsets]

                def xstep():
                    return plan_groups(xg)

                xstep()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    xo = xstep()
                torch.cuda.synchronize()
                xe = (time.perf_counter() - t0) / 2
                xs = summarize([records(o[0], g.n) for o, g in zip(xo, xg)], [g.slots for g in xg], xe)
                xs["workload"] = lab
                xs["kernel_form"] = "one workgroup per problem" if xg[0].mode == 1 else "one wave per problem"
                out[name] = xs
                del xg
            if not a.pmc_mode:
                # ---- a saturating batch (4 x the 4 096 set, goals re-paired): the chip's sustained expansion rate in both
                # kernel forms (avp_plan_batch_mode): the workgroup form keeps 256 problems in flight, the wave form 2 048
                lab, xcfg, xcap, xsets = build("batch4096")
                mm, st4, go4 = xsets[0]
                st16 = np.concatenate([st4] * 4)
                go16 = np.concatenate([np.roll(go4, 17 * k, axis=0) for k in range(4)])
                sat = {"workload": "Case1 map, 16384 problems (the 4096 starts against 4 rotations of the goals), pop cap 1000"}
                for mode, key in ((1, "workgroup_per_problem"), (2, "wave_per_problem")):
                    g16 = Group(mm, veh, xcfg, st16, go16, local, xcap, mode=mode)
                    g16.bp.plan_dev(g16.st_t, g16.go_t, want_paths=True)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    o16 = g16.bp.plan_dev(g16.st_t, g16.go_t, want_paths=True)
                    torch.cuda.synchronize()
                    sat[key] = summarize([records(o16[0], g16.n)], [g16.slots], time.perf_counter() - t0)
                    del g16
                out["saturating_batch"] = sat
            # ---- the footprint-collision kernel alone ----------------------------------------------------------------
            dm = groups[0].dm
            m = groups[0].m
            rng = np.random.default_rng(1)
            n_chk = 1 << 20
            cp = np.stack([rng.uniform(m.boundary[0] + 6, m.boundary[1] - 6, n_chk), rng.uniform(m.boundary[2] + 6, m.boundary[3] - 6, n_chk),
                           rng.uniform(-np.pi, np.pi, n_chk)], 0)
            ct = dm.dev_tensor(cp)
            co = dm.empty(n_chk, torch.uint8)
            dm.check_batch_dev(ct[0], ct[1], ct[2], out=co)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dm.check_batch_dev(ct[0], ct[1], ct[2], out=co)
            e1.record()
            torch.cuda.synchronize()
            cms = e0.elapsed_time(e1) / 10
            rc = {"kernel": "check_distance_kernel", "launch_ms": cms, "checks_per_s": n_chk / (cms * 1e-3), "bytes_per_check_reference": B_cc,
                  "hbm_algorithmic_GBps": n_chk * B_cc / (cms * 1e-3) / 1e9,
                  "note": "the reference formulation reads every obstacle point per check (16P+25 B); the kernel keeps the map in LDS and moves 25 B/check of HBM, so its bound is VALU/LDS issue, not HBM",
                  "hbm_traffic_GBps": n_chk * 25 / (cms * 1e-3) / 1e9, "frac_hbm_traffic": n_chk * 25 / (cms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                  "bound": "valu", "unit": "G SIMD-cycles/s", "peak": N_SIMD * CLOCK_GHZ, "achieved": None, "frac": None, "traffic": None}
            if pmc and not pmc.get("stale") and "check_distance_kernel" in pmc:
                ck = pmc["check_distance_kernel"]
                rc["achieved"] = ck["valu_active_simd_cycles_per_launch"] / (cms * 1e-3) / 1e9
                rc["frac"] = rc["achieved"] / rc["peak"]
                rc["traffic"] = ck.get("hbm_bytes_per_launch_corrected")
                rc["lds_busy_frac"] = ck.get("lds_busy_frac")
                rc["lds_bank_conflict_frac"] = ck.get("lds_bank_conflict_frac")
            out["roofline_check"] = rc
            if not a.no_cpu_baseline and not a.pmc_mode:
                from oracle import oracle
                g0 = groups[0]
                o = oracle.Oracle(g0.m, veh, wcfg, max_pops=cap)
                nb = min(g0.n, 256)
                t1 = time.perf_counter()
                pops_cpu = done_cpu = 0
                for s_, g_ in zip(g0.starts[:nb], g0.goals[:nb]):
                    w = o.plan(s_, g_, max_trace=1)
                    pops_cpu += w["n_pops"]
                    done_cpu += w["status"] in (0, 1)
                tc = time.perf_counter() - t1
                out["cpu_baseline"] = {"value": done_cpu / tc, "unit": "plans/s", "cores": 1, "kind": "port",
                                       "sample": f"the first {nb} problems of the headline workload, pop cap {cap}, C restatement (oracle/avp_oracle.c, glibc libm), {tc:.1f} s; completed searches only, like `value`",
                                       "all_problems_per_s": nb / tc, "expansions_per_s": pops_cpu / tc}
                from concurrent.futures import ThreadPoolExecutor
                ncore = os.cpu_count() or 1
                t2 = time.perf_counter()
                with ThreadPoolExecutor(max_workers=ncore) as ex:
                    st_all = list(ex.map(lambda sg: o.plan(sg[0], sg[1], max_trace=1)["status"], zip(g0.starts[:nb], g0.goals[:nb])))
                tm = time.perf_counter() - t2
                out["cpu_baseline_all_cores"] = {"value": sum(s in (0, 1) for s in st_all) / tm, "unit": "plans/s", "cores": ncore, "kind": "port",
                                                 "sample": f"the same {nb} problems, one per thread, {tm:.1f} s"}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
