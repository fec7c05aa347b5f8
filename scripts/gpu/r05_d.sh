#!/bin/bash
# round-5 fourth GPU run: whole GPU suite on the kernels as they will ship (round 4's plan_kernel structure + the general collision pass;
# circle checker with the squared-distance first look; batched ingest), the secondary-kernel rates, one bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r05d; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 300 python scripts/bench_check.py --variants 0 2>/dev/null | cut -c1-200 | tee $O/bench_check.jsonl
timeout 300 python scripts/bench_check.py --variants 0 --case 19 --iters 10 2>/dev/null | grep circle | cut -c1-200 | tee -a $O/bench_check.jsonl
timeout -k 10 900 python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; tail -3 $O/bench_n1.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05d/bench_n1.json"))
print(json.dumps({k: d.get(k) for k in ("value", "ms_per_step", "expansions_per_s", "cpu_baseline_all_cores")})[:2500])
print({k: (d[k].get("ms_per_step") if isinstance(d.get(k), dict) else None) for k in ("c3", "c5", "batch4096", "without_lookahead")})
PY
