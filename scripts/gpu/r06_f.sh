#!/bin/bash
# lookahead vs batch size, variants of the helpers : owners gate (scripts/look_scale.py); VARS="look16 ratio4 ..." NS="256 512 ..."
mkdir -p gpurun_out/r06f
: > gpurun_out/r06f/scale.log
for rep in $(seq 1 ${REPS:-2}); do
  for v in ${VARS:-look16 ratio2 ratio4 ratio8 ratio12}; do
    L=automatedvaletparking_amd/variants/libavp_hip_$v.so; [ $v = default ] && L=automatedvaletparking_amd/libavp_hip.so
    AVP_HIP_LIB=$L python scripts/look_scale.py ${NS:-256 512 768 1024 1536 2048 3000} 2>/dev/null >> gpurun_out/r06f/scale.log
  done
done
python - <<'PY'
import json, collections
rows=[json.loads(l) for l in open('gpurun_out/r06f/scale.log') if l.startswith('{')]
t=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows: t[r['lib']][r['n']].append((r['ms_off'], r['ms_on'], r.get('record_pop_frac'), r['identical']))
for lib in t:
    print(lib)
    for n in sorted(t[lib]):
        print('  n=%5d ' % n, '  '.join('off %6.2f on %6.2f rp %s %s' % (a, b, c, 'ok' if d else 'DIFF') for a, b, c, d in t[lib][n]))
PY
