"""Regenerates the round-5 / round-6 table of DESIGN.md section 5 from the committed evidence (profiles/r06_*): the block between the
markers <!-- r06-table-begin --> and <!-- r06-table-end -->. Run after scripts/gpu/collect_r06.sh.   python scripts/fill_design.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def sci(v, d=2):
    e = int(("%e" % v).split("e")[1])
    return ("%." + str(d) + "f·10%s") % (v / 10 ** e, "".join("⁰¹²³⁴⁵⁶⁷⁸⁹"[int(c)] for c in str(e)))


def main():
    d = json.load(open(os.path.join(P, "r06_bench_n1.json")))
    look = json.load(open(os.path.join(P, "r06_lookahead.json")))["with_lookahead"]
    sec = {}
    for line in open(os.path.join(P, "r06_secondary_kernels.jsonl")):
        if line.strip():
            r = json.loads(line)
            sec[(r["kernel"], r.get("variant", 0))] = r
    r = d["roofline"]
    cs = d["cap_sweep"]["c2"]
    sat = d["saturating_batch"]
    rc = d["roofline_check"]
    rows = [
        ("**config[1]** (Case1, 256 pairs, cap 1000; workgroup form + lookahead) ms / step", "18.7", "**%.1f**" % d["ms_per_step"]),
        ("completed plans/s (204 of 256; capped searches excluded) · expansions/s", "10 840 · 2.86·10⁶", "**%s · %s**" % (format(int(round(d["value"], -1)), ",").replace(",", " "), sci(d["expansions_per_s"]))),
        ("same step without the lookahead (identical results: `%s`)" % str(d["without_lookahead"]["identical_results"]).lower(), "32.1", "%.1f" % d["without_lookahead"]["ms_per_step"]),
        ("record pops / all pops · records never used", "0.79 · 45 %", "%.2f · %.0f %%" % (look["record_pop_frac"], 100 * look["records_never_used_frac"])),
        ("VALU wave-instructions per pop · VALU-busy (`roofline.frac`) · live lanes · wave cycles parked · HBM traffic / launch", "57.9 k · 0.279 · 0.37 · 0.71 · 1.95 GB",
         "%.1f k · %.3f · %.2f · %.2f · %.2f GB" % (r["valu_wave_insts_per_pop"] / 1e3, r["frac"], r["valu_lane_utilisation"], r["wait_frac"], r["traffic"] / 1e9)),
        ("`frac_hbm_algorithmic` (SURVEY 8d bytes per launch ÷ `launch_ms` ÷ 8 TB/s) · `launch_ms` (HIP events)", "0.54 · 18.72", "%.2f · %.2f" % (r["frac_hbm_algorithmic"], r["launch_ms"])),
        ("cap sweep config[1]: caps 300 / 1000 / 3000, ms (lookahead)", "7.2 / 18.8 / 93.5 (on / on / **off**)",
         " / ".join("%.1f" % cs[k]["ms_per_step"] for k in ("300", "1000", "3000")) + " (" + " / ".join("on" if cs[k]["lookahead"] else "off" for k in ("300", "1000", "3000")) + "); %.1f µs / pop at 3 000" % cs["3000"]["us_per_pop_of_the_longest_search"]),
        ("4 096-pose batch, best form (all five identical: `%s`) ms" % str(d["batch4096"]["forms_identical"]).lower(), "89.6 (quad)", "%.1f (%s)" % (d["batch4096"]["ms_per_step"], d["batch4096"]["kernel_form"])),
        ("16 384 problems (pair form, time-sliced) · 32 768 (wave form, time-sliced) expansions/s", "16.6·10⁶ · 21.1·10⁶", "%s · %s" % (sci(sat["pair_per_problem"]["expansions_per_s"]), sci(sat["n32768_wave_per_problem"]["expansions_per_s"]))),
        ("C3 (20 maps × 128, 20 streams, no lookahead) · C5 (1 024 starts, cap 300) ms", "78.1 · 65.7", "%.1f · %.1f" % (d["c3"]["ms_per_step"], d["c5"]["ms_per_step"])),
        ("the 20 BenchmarkCases' own problems, one launch each, total ms · one `path_planning()` call on Case1 ms", "1 406 · 3.8", "%s · %.1f" % (format(int(round(d["cases20"]["total_ms_one_after_the_other"])), ",").replace(",", " "), d["single_plan_latency_ms"])),
        ("`check_distance_kernel` checks/s (VALU-busy) · near-miss point tests/s (VALU ceiling)", "2.71·10⁹ (0.44) · 7.4·10¹⁰", "%s (%.2f) · %s (%s)" % (sci(rc["checks_per_s"]), rc["frac"], sci(rc["near_miss_poses"]["point_tests_per_s"], 1), sci(rc["near_miss_poses"]["valu_ceiling_point_tests_per_s"], 1))),
        ("`check_circle_kernel` · `rs_optimal_kernel` · `corridor_compact_kernel` per second", "5.5·10⁹ · 8.2·10⁸ · 4.7·10⁸",
         "%s · %s · **%s**" % (sci(sec[("circle", 0)]["checks_per_s"], 1), sci(sec[("rs_optimal (lengths only)", 0)]["solves_per_s"], 1), sci(sec[("corridor", 0)]["waypoints_per_s"], 1))),
        ("CPU port (oracle): 1 core · %d threads on the container's quota of %d CPUs, plans/s · GPU / CPU in expansions/s" % (d["cpu_baseline_all_cores"]["cores"], d["cpu_baseline_all_cores"].get("cpu_quota", d["cpu_baseline_all_cores"]["cores"])), "63 · 986 · 11.0",
         "%.0f · %.0f · %.1f" % (d["cpu_baseline"]["value"], d["cpu_baseline_all_cores"]["value"], d["cpu_baseline_all_cores"]["gpu_over_cpu_all_cores_expansions"])),
    ]
    tab = "| | round 5 | round 6 |\n|---|---|---|\n" + "\n".join("| %s | %s | %s |" % r_ for r_ in rows) + "\n"
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    s2, n = re.subn(r"<!-- r06-table-begin -->\n.*?<!-- r06-table-end -->", "<!-- r06-table-begin -->\n" + tab + "<!-- r06-table-end -->", s, flags=re.S)
    assert n == 1, "markers not found"
    open(path, "w").write(s2)
    print(tab)


if __name__ == "__main__":
    main()
