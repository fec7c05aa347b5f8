"""Fold the rocprofv3 PMC passes of `bench.py --pmc-mode` into profiles/r06_pmc_summary.json.

    python scripts/pmc_summary.py <dir with one sub-directory per pass> > profiles/r06_pmc_summary.json

Passes (each its own run, --kernel-trace --pmc only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes):
  sq    SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS
  lds   SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU
  lane  SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
        (lane utilisation of the VALU work = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): rocprofv3's own VALUUtilization expression)
  fetch FETCH_SIZE     write WRITE_SIZE    (KB; gfx950 FETCH_SIZE counts 64 B per 128-B request: HBM bytes = (2 FETCH + WRITE) * 1024)
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (4 shader cycles). Counter values are summed over all
instances of a dispatch; per kernel the launches of the largest grid are averaged. The file is stamped with the hash of
the kernel sources it was measured on; bench.py refuses a stale one."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_hash  # noqa: E402


def collect(root):
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))      # kernel -> dispatch -> counter -> value
    grid = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            d = (f, r["Dispatch_Id"])
            per[k][d][r["Counter_Name"]] += float(r["Counter_Value"])
            grid[(k, d)] = int(r.get("Grid_Size", 0) or 0)
    out = {}
    for k, disp in per.items():
        gmax = max(grid[(k, d)] for d in disp)
        sel = [v for d, v in disp.items() if grid[(k, d)] == gmax]
        out[k] = {c: sum(v[c] for v in sel) / len(sel) for c in sel[0]}
        out[k]["launches"] = len(sel)
        out[k]["grid_size"] = gmax
    return out


def durations(root):
    """Average launch duration [ms] per kernel (largest launches only: the top half by duration) from the kernel traces."""
    per = defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            per[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    out = {}
    for k, v in per.items():
        big = [x for x in v if x >= 0.5 * max(v)]
        out[k] = {"avg_ms": sum(big) / len(big), "launches": len(big)}
    return out


def main():
    base = sys.argv[1]
    passes = {name: collect(os.path.join(base, name)) for name in ("sq", "lds", "lane", "fetch", "write") if os.path.isdir(os.path.join(base, name))}
    res = {"source_hash": source_hash(),
           "command": "rocprofv3 --kernel-trace --pmc <pass counters> -- python bench.py --steps 3 --warmup 1 --pmc-mode  (one run per pass)",
           "units": "SQ_* cycle counters in quad-cycles; *_simd_cycles in shader cycles; bytes per launch"}
    dur = durations(os.path.join(base, "sq"))
    # the headline's planner instantiation: plan_kernel<STAGE, PROFILE, LOOK> with the most time in the sq pass
    plan_names = [k for k in dur if k.startswith("plan_kernel<")]
    plan_name = max(plan_names, key=lambda k: dur[k]["avg_ms"] * dur[k]["launches"]) if plan_names else "plan_kernel<true, false, true>"
    res["plan_kernel_instantiation"] = plan_name
    for kern, key in ((plan_name, "plan_kernel"), ("check_distance_kernel<true>", "check_distance_kernel"),
                      ("rs_optimal_kernel", "rs_optimal_kernel"), ("corridor_compact_kernel<true>", "corridor_compact_kernel"), ("check_circle_kernel", "check_circle_kernel")):
        e = {}
        sq = passes.get("sq", {}).get(kern)
        if sq:
            e["raw_sq"] = sq
            e["valu_active_simd_cycles_per_launch"] = 4.0 * sq["SQ_ACTIVE_INST_VALU"]
            e["wait_any_frac"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
            e["valu_active_frac_of_wave_cycles"] = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]
            e["valu_insts_per_launch"] = sq["SQ_INSTS_VALU"]
        ld = passes.get("lds", {}).get(kern)
        if ld:
            e["raw_lds"] = ld
            if ld.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_bank_conflict_frac"] = ld["SQ_LDS_BANK_CONFLICT"] / ld["SQ_LDS_IDX_ACTIVE"]
            e["lds_insts_per_launch"] = ld.get("SQ_INSTS_LDS")
            e["f64_valu_insts_per_launch"] = sum(ld.get(c, 0.0) for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64"))
            e["f64_valu_wave_insts_per_launch"] = {"fma": ld.get("SQ_INSTS_VALU_FMA_F64", 0.0), "mul": ld.get("SQ_INSTS_VALU_MUL_F64", 0.0), "add": ld.get("SQ_INSTS_VALU_ADD_F64", 0.0)}
        la = passes.get("lane", {}).get(kern)
        if la:
            e["raw_lane"] = la
            if la.get("SQ_ACTIVE_INST_VALU"):
                e["valu_lane_utilisation"] = la["SQ_THREAD_CYCLES_VALU"] / (64.0 * la["SQ_ACTIVE_INST_VALU"])
        fe, wr = passes.get("fetch", {}).get(kern), passes.get("write", {}).get(kern)
        if fe and wr:
            e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"] = fe["FETCH_SIZE"], wr["WRITE_SIZE"]
            e["hbm_bytes_per_launch_corrected"] = (2.0 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024.0
        du = dur.get(kern)
        if du:
            e["launch_ms_under_rocprof"] = du["avg_ms"]
            if "valu_active_simd_cycles_per_launch" in e:
                # fraction of the chip's SIMD cycles (1024 SIMDs x 2.4 GHz) in which a VALU instruction was active
                e["valu_busy_frac_of_chip"] = e["valu_active_simd_cycles_per_launch"] / (du["avg_ms"] * 1e-3 * 1024 * 2.4e9)
            if "hbm_bytes_per_launch_corrected" in e:
                e["hbm_GBps"] = e["hbm_bytes_per_launch_corrected"] / (du["avg_ms"] * 1e-3) / 1e9
            if "f64_valu_wave_insts_per_launch" in e:
                fl = e["f64_valu_wave_insts_per_launch"]
                # upper bound: every counted wave-instruction as 64 live lanes; x lane utilisation = the estimate
                e["fp64_flops_frac_upper_bound"] = 64.0 * (2 * fl["fma"] + fl["mul"] + fl["add"]) / (du["avg_ms"] * 1e-3) / 78.6e12
        if e:
            res[key] = e
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
