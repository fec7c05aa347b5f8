#!/bin/bash
# all-capped sets (scripts/look_capped.py) and batch sizes (scripts/look_scale.py) per variant; VARS="default ts4 ..."
mkdir -p gpurun_out/r06g
: > gpurun_out/r06g/capped.log; : > gpurun_out/r06g/scale.log
for rep in $(seq 1 ${REPS:-2}); do
  for v in ${VARS:-default}; do
    L=automatedvaletparking_amd/variants/libavp_hip_$v.so; [ $v = default ] && L=automatedvaletparking_amd/libavp_hip.so
    AVP_HIP_LIB=$L python scripts/look_capped.py ${KS:-64 100 128 160 200 256 400} 2>/dev/null >> gpurun_out/r06g/capped.log
    AVP_HIP_LIB=$L python scripts/look_scale.py ${NS:-256 512 768 1024 1536 2048} 2>/dev/null >> gpurun_out/r06g/scale.log
  done
done
python - <<'PY'
import json, collections
for f, key in (("capped", "capped_searches"), ("scale", "n")):
    rows=[json.loads(l) for l in open('gpurun_out/r06g/%s.log' % f) if l.startswith('{')]
    t=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows: t[r['lib']][r[key]].append((r['ms_on'], r.get('record_pop_frac')))
    libs=list(t)
    print(f, 'ms_on (record-pop frac) per lib:', libs)
    for n in sorted({n for l in t for n in t[l]}):
        off=[r['ms_off'] for r in rows if r[key]==n]
        print('  %5d off %6.2f | ' % (n, min(off)), ' | '.join(' '.join('%6.2f(%.2f)' % (a, b or 0) for a,b in t[l][n]) for l in libs))
PY
