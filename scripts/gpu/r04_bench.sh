#!/bin/bash
# round-4 working run: the default bench line only. Output under gpurun_out/r04b/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04b; mkdir -p $O
timeout -k 10 600 python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; tail -3 $O/bench_n1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04b/bench_n1.json'))
print('headline ms', d['ms_per_step'], 'value', d['value'], 'exp/s', d['expansions_per_s'])
print('without lookahead', d['without_lookahead']['ms_per_step'])
print('batch4096', d['batch4096']['ms_per_step'], d['batch4096'].get('forms_ms_per_step'))
print('sat', {k:round(v.get('ms_per_step'),1) for k,v in d['saturating_batch'].items() if isinstance(v,dict)})
print('c3', d['c3'].get('ms_per_step'), 'c5', d['c5'].get('ms_per_step'))
PY
