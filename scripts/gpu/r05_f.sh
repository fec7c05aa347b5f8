#!/bin/bash
# round-5: late adoption of expansion records (PL_LOOK_LATE) -- parity first, then the variants in one box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05f; rm -rf $O; mkdir -p $O
V=automatedvaletparking_amd/variants
timeout -k 10 900 python -m pytest tests/test_gpu_lookahead.py tests/test_gpu_plan.py -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
timeout 300 python scripts/look_bench.py > $O/lookahead.json 2> $O/lookahead.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05f/lookahead.json"))
w = d["with_lookahead"]
print({k: w.get(k) for k in ("ms_best", "jobs_posted", "records_used", "records_adopted_late", "record_pop_frac", "child_lookups")}, d["identical_results"])
PY
for rep in 1 2; do
  for v in default $VARS; do
    L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
    echo "== rep $rep $v"; timeout 300 python scripts/variant_bench.py --no-profile --no-big --steps 8 $L 2>/dev/null | tail -1 | cut -c1-200
  done
done 2>&1 | tee $O/sweep.log
