#!/bin/bash
# round 6, first GPU call: the whole GPU suite on the fixed-size record store + dive prediction, then the lookahead's A/B per variant
# (same box, two alternating rounds): default / nofetch (prediction from prefetched records only) / nopred (no prediction) / top8.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06a; rm -rf $O; mkdir -p $O
V=automatedvaletparking_amd/variants
timeout -k 10 600 python -m pytest tests/test_gpu_lookahead.py -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
for rep in ${REPS:-1 2}; do
  for v in default $VARS; do
    L=""; [ $v != default ] && L="$PWD/$V/libavp_hip_$v.so"
    echo "== rep $rep $v"
    AVP_HIP_LIB=$L timeout 300 python scripts/look_bench.py > $O/look_${v}_$rep.json 2> $O/look_${v}_$rep.err
    python - $O/look_${v}_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
w = d["with_lookahead"]
print({k: w.get(k) for k in ("ms_best", "jobs_posted", "records_used", "records_adopted_late", "record_pop_frac", "children_posted_by_dive_prediction", "children_posted_by_helpers", "claims_refused_entry_busy", "copies_refused_by_seqlock", "next_node_lookups", "misses_by_kind")}, d["identical_results"], d["without_lookahead"]["ms_best"])
print(d.get("record_pops_of_capped_problems", {}).get("record_pop_frac"))
PY
  done
done 2>&1 | tee $O/sweep.log
