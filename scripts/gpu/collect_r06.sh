#!/bin/bash
# Copies the summaries of the round-6 evidence run (gpurun_out/r06/, scripts/gpu/profile_r06.sh) into profiles/ (tracked).
cd "$(dirname "$0")/../.."
O=gpurun_out/r06; P=profiles
cp $O/bench_n1.json $P/r06_bench_n1.json
cp $O/bench_force_dist.json $P/r06_bench_force_dist_n1.json
cp $O/pmc_summary.json $P/r06_pmc_summary.json
cp $O/pmc_saturating_batch.json $P/r06_pmc_saturating_batch.json
cp $O/bench_headline_under_rocprof.json $P/r06_bench_headline_under_rocprof.json
cp "$(ls -t $(find $O/stats_headline -name '*kernel_stats.csv') | head -1)" $P/r06_bench_headline_kernel_stats.csv
cp "$(ls -t $(find $O/stats_check -name '*kernel_stats.csv') | head -1)" $P/r06_secondary_kernel_stats.csv
cp $O/bench_check.jsonl $P/r06_secondary_kernels.jsonl
cp $O/phase_profile.json $P/r06_plan_kernel_phase_cycles.json
python3 - <<'PY'
import json, glob, os
O, P = "gpurun_out/r06", "profiles"
w = {}
for m in (2, 3, 4):
    f = f"{O}/wave_profile_m{m}.json"
    if os.path.exists(f) and os.path.getsize(f):
        w[{2: "one wave per problem", 3: "a pair of waves per problem", 4: "four waves per problem"}[m]] = json.load(open(f))
json.dump(w, open(f"{P}/r06_group_forms_phase_cycles.json", "w"), indent=1)
s = {}
for f in sorted(glob.glob(f"{O}/soak_*.json")):
    if os.path.getsize(f):
        s[os.path.basename(f)[5:-5]] = json.load(open(f))
json.dump(s, open(f"{P}/r06_lookahead_soak.json", "w"), indent=1)
PY
cp $O/lookahead.json $P/r06_lookahead.json
cp $O/large_maps.json $P/r06_large_maps.json
cp $O/slice_soak.json $P/r06_time_slicing_soak.json
cat $O/slice_*_off.json $O/slice_*_on.json > $P/r06_time_slicing.jsonl
tail -n 6 $O/pytest_gpu.log > $P/r06_pytest_gpu_tail.txt
ls -la $P | grep r06
cp $O/pmc_icache.json $P/r06_pmc_icache.json
cp $O/config_sweep.jsonl $P/r06_config_sweep.jsonl
cp $O/cap_growth.jsonl $P/r06_cap_growth.jsonl
cp $O/lookahead_batch_sizes.jsonl $P/r06_lookahead_batch_sizes.jsonl
cp $O/lookahead_capped_sets.jsonl $P/r06_lookahead_capped_sets.jsonl
