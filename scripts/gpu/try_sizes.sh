#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/look
for n in 256 512; do
timeout 600 python scripts/look_bench.py $n 1000 > gpurun_out/look/n$n.json 2> gpurun_out/look/n$n.err
python - <<PY
import json
d=json.load(open('gpurun_out/look/n$n.json'))
w=d['with_lookahead']
print($n, d['without_lookahead']['ms_best'], '->', w['ms_best'], d['identical_results'], w.get('record_pop_frac'), w.get('jobs_posted'), w.get('child_lookups'))
PY
done
timeout 600 python -m pytest tests/test_gpu_lookahead.py -x -q 2>&1 | tail -n 3
