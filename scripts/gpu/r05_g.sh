#!/bin/bash
# config[1] ms per batch for the default library and the variants named in $VARS (one box, two alternating rounds) + their lookahead counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05g; rm -rf $O; mkdir -p $O
V=automatedvaletparking_amd/variants
for rep in 1 2; do
  for v in default $VARS; do
    L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
    echo "== rep $rep $v"; timeout 300 python scripts/variant_bench.py --no-profile --no-big --steps 8 $L 2>/dev/null | tail -1 | cut -c1-200
  done
done 2>&1 | tee $O/sweep.log
for v in default $VARS; do
  L=""; [ $v != default ] && export AVP_HIP_LIB=$PWD/$V/libavp_hip_$v.so || unset AVP_HIP_LIB
  echo "== counters $v"; timeout 300 python scripts/look_bench.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); w=d['with_lookahead']
print({k: w.get(k) for k in ('ms_best','jobs_posted','records_used','records_adopted_late','successor_jobs','child_lookups')}, d['identical_results'])"
done 2>&1 | tee $O/counters.log
