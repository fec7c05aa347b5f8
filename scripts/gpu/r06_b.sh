#!/bin/bash
# round 6: the refactored bench line, the N > 1 code paths at world 1, the secondary kernels, the two tests fixed since the last suite run
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py tests/test_gpu_corridor.py -q -m gpu -k "config_variants or golden_problems or corridor" > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
timeout -k 10 300 python scripts/bench_check.py --iters 10 > $O/bench_check.jsonl 2> $O/bench_check.err; cut -c1-400 $O/bench_check.jsonl
timeout -k 10 900 python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; head -c 700 $O/bench_n1.json; echo; tail -3 $O/bench_n1.err
AVP_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --steps 3 --warmup 1 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; head -c 300 $O/bench_force_dist.json; echo; tail -3 $O/bench_force_dist.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06b/bench_n1.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "look", d["config"]["expansion_lookahead"], "wo", d["without_lookahead"]["ms_per_step"], d["without_lookahead"]["identical_results"])
print("cap_sweep c2", {k: (round(v["ms_per_step"], 2), v["lookahead"], round(v["us_per_pop_of_the_longest_search"], 2)) for k, v in d["cap_sweep"]["c2"].items()})
print("c3", d["c3"]["ms_per_step"], d["c3"]["lookahead"], "c5", d["c5"]["ms_per_step"], d["c5"]["lookahead"])
print("batch4096", d["batch4096"]["forms_ms_per_step"], d["batch4096"]["forms_identical"])
print("sat", {k: round(v["ms_per_step"], 1) for k, v in d["saturating_batch"].items() if isinstance(v, dict)})
print("cases20 total", d["cases20"]["total_ms_one_after_the_other"], "single", d["single_plan_latency_ms"])
print("check", d["roofline_check"]["checks_per_s"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline_all_cores"]["value"])
f = json.load(open("gpurun_out/r06b/bench_force_dist.json"))
print("dist", f["ms_per_step"], f["collective"], f["throughput"].get("weak_scaling_efficiency_in_run"), f["strong_scaling_4096"]["ms_per_step"])
PY
