"""SQ counters of the planner kernels on a saturating batch (profiles/r06_pmc_saturating_batch.json).

    python scripts/pmc_sat_summary.py <dir with sub-directories sq/ and lane/, each holding the rocprofv3 csv output of
        rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python scripts/variant_bench.py --big 16384 --big-mode M --no-profile --steps 1>

Per kernel (the launches of the largest grid, averaged): launch time, VALU-busy share of the chip's SIMD cycles
(4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x 2.4 GHz x time)), the shares of the wave cycles spent parked (SQ_WAIT_ANY: s_waitcnt,
barriers, sleep), waiting to issue (SQ_WAIT_INST_ANY) and issuing (SQ_ACTIVE_INST_ANY), instruction counts, lane utilisation."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from pmc_summary import collect, durations  # noqa: E402
from bench import source_hash  # noqa: E402


def main():
    base = sys.argv[1]
    out = {"source_hash": source_hash(), "workload": "Case1 map, 16384 problems, pop cap 1000 (scripts/variant_bench.py --big 16384 --big-mode M)"}
    for sub in sorted(os.listdir(base)):
        d = os.path.join(base, sub)
        if not os.path.isdir(os.path.join(d, "sq")):
            continue
        sq, dur = collect(os.path.join(d, "sq")), durations(os.path.join(d, "sq"))
        lane = collect(os.path.join(d, "lane")) if os.path.isdir(os.path.join(d, "lane")) else {}
        for k, v in sq.items():
            if not k.startswith(("plan_wave_kernel", "plan_kernel")) or k not in dur or dur[k]["avg_ms"] < 5.0:
                continue
            ms = dur[k]["avg_ms"]
            e = {"launch_ms_under_rocprof": ms, "valu_busy_frac_of_chip": 4.0 * v["SQ_ACTIVE_INST_VALU"] / (ms * 1e-3 * 1024 * 2.4e9),
                 "wait_any_frac": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], "wait_inst_any_frac": v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
                 "active_inst_any_frac": v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], "valu_insts": v.get("SQ_INSTS_VALU"), "raw_sq": v}
            la = lane.get(k)
            if la and la.get("SQ_ACTIVE_INST_VALU"):
                e["valu_lane_utilisation"] = la["SQ_THREAD_CYCLES_VALU"] / (64.0 * la["SQ_ACTIVE_INST_VALU"])
                e["raw_lane"] = la
            out[f"{sub}: {k}"] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
