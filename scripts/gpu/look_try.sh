#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/look
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export AVP_HIP_LIB=$PWD/automatedvaletparking_amd/variants/libavp_hip_$v.so; else unset AVP_HIP_LIB; fi
  timeout 300 python scripts/look_bench.py > gpurun_out/look/l_$v.json 2> gpurun_out/look/l_$v.err || tail -n 5 gpurun_out/look/l_$v.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/look/l_$v.json'))
    print('variant "$v":', d['without_lookahead']['ms_best'], '->', d['with_lookahead']['ms_best'], 'identical', d['identical_results'])
    print('   ', {k:v for k,v in d['with_lookahead'].items() if k not in ('ms_per_batch',)})
    r=d['record_pops_of_capped_problems']; print('    capped: record pops %.3f; wave0 timeline %s' % (r['record_pop_frac'], [round(r['cycles_since_pop_start_per_wave'][k][0]) for k in ('children_ready','resolution_done','end_of_pop')]))
except Exception as e: print('variant "$v" failed', e)
PY
done
unset AVP_HIP_LIB
if [ -n "$RUN_TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_lookahead.py tests/test_gpu_plan_wave.py $RUN_TESTS -x -q 2>&1 | tail -n 6; fi
