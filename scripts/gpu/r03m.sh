#!/bin/bash
# round 3, GPU call M: time-sliced group forms -- parity tests, then the 4 096 / 16 384 batches with and without
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_plan_wave.py -m gpu -x -q -k "time_sliced or case1_batch" 2>&1 | tail -15 > $O/pytest.txt
tail -n 8 $O/pytest.txt
vb() { timeout 200 python scripts/variant_bench.py --no-profile "$@" 2>$O/err.txt | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d.get(k) for k in ('big_n','big_mode','time_sliced','slice_pops','big_ms','big_expansions_per_s','big_digest')})"; tail -n 2 $O/err.txt; }
vb --big 4096 --big-mode 4 --steps 2 --slice off | tee $O/q4096_off.txt
vb --big 4096 --big-mode 4 --steps 2 --slice on | tee $O/q4096_on64.txt
vb --big 4096 --big-mode 4 --steps 2 --slice on --slice-pops 32 | tee $O/q4096_on32.txt
vb --big 4096 --big-mode 4 --steps 2 --slice on --slice-pops 128 | tee $O/q4096_on128.txt
vb --big 16384 --big-mode 2 --steps 1 --slice off | tee $O/w16384_off.txt
vb --big 16384 --big-mode 2 --steps 1 --slice on | tee $O/w16384_on64.txt
vb --big 16384 --big-mode 4 --steps 1 --slice on | tee $O/q16384_on64.txt
