#!/bin/bash
# A/B of sweep variants (variants/libavp_hip_<name>.so, $@ = names) against the working build in ONE call: headline batch +
# the 2 048 batch in the workgroup form, two repetitions, then the phase profile of each.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04ab; mkdir -p $O
V=automatedvaletparking_amd/variants
for rep in 1 2; do
  for name in WORK "$@"; do
    lib=""; [ $name != WORK ] && lib="--lib $V/libavp_hip_$name.so"
    timeout 300 python scripts/variant_bench.py --no-profile $lib 2>/dev/null | tail -1 | cut -c1-420
  done
done 2>&1 | tee $O/ab2.log
for name in WORK "$@"; do
  lib=""; [ $name != WORK ] && lib="--lib $V/libavp_hip_$name.so"
  timeout 300 python scripts/variant_bench.py $lib 2>/dev/null | tail -1 > $O/prof_$name.json
  python - <<PY
import json
d=json.load(open("$O/prof_$name.json"))
p=d.get("phase_cyc_per_pop",{})
print("$name", d["c2_ms"], d["big_ms"], {k:p.get(k) for k in ("init","slow_resolve","(sweep)","res_push")}, d.get("cyc_per_pop"))
PY
done 2>&1 | tee -a $O/ab2.log
