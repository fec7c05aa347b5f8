#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2m
for v in s4 h3 s4h3 s4h4 s3h2; do
timeout -k 10 120 python scripts/variant_bench.py --lib automatedvaletparking_amd/variants/libavp_hip_$v.so --big 256 > gpurun_out/r2m/vb_$v.json 2> gpurun_out/r2m/vb_$v.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2m/vb_$v.json"))
print("$v", d["c2_ms"], d["digest"], d.get("cyc_per_pop"), {k:d["phase_cyc_per_pop"][k] for k in ("children||substeps","rs_words..replay","resolve||shot","shot_round0","shot_all","res_push","pop_ahead")})
PY
done
