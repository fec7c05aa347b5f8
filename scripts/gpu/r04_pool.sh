#!/bin/bash
# exploration for the next round (variants built outside the tree, nothing of it is in the product): the workgroup-wide
# Reeds-Shepp query pool with a batching window, wave form, 16 384 and 32 768 problems
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04ab; mkdir -p $O
V=automatedvaletparking_amd/variants
for cfg in "2 16384" "2 32768"; do set -- $cfg
  for lib in "" "--lib $V/libavp_hip_pool0.so" "--lib $V/libavp_hip_pool10000.so" "--lib $V/libavp_hip_pool30000.so"; do
    timeout -k 5 40 python scripts/variant_bench.py --no-profile --steps 1 $lib --big $2 --big-mode $1 2>/dev/null | tail -1 > $O/one.json
    python -c "
import json
try:
    d=json.load(open('$O/one.json')); print(d['lib'], 'mode', d['big_mode'], 'n', d['big_n'], 'sliced', d['time_sliced'], 'ms', d['big_ms'], 'exp/s', d['big_expansions_per_s'], d['big_digest'])
except Exception as e: print('FAILED / timed out', '$lib', '$cfg')"
  done
done 2>&1 | tee $O/pool.log
