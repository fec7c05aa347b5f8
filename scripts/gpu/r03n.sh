#!/bin/bash
# round 3, GPU call N: time slicing where the long searches outnumber the groups (32 768 problems, wave form; 16 384, pair form)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03n; mkdir -p $O
vb() { timeout 300 python scripts/variant_bench.py --no-profile "$@" 2>$O/err.txt | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d.get(k) for k in ('big_n','big_mode','time_sliced','slice_pops','big_ms','big_expansions_per_s','big_digest')})"; tail -n 1 $O/err.txt; }
vb --big 32768 --big-mode 2 --steps 1 --slice off | tee $O/w32768_off.txt
vb --big 32768 --big-mode 2 --steps 1 --slice on | tee $O/w32768_on64.txt
vb --big 32768 --big-mode 2 --steps 1 --slice on --slice-pops 200 | tee $O/w32768_on200.txt
vb --big 16384 --big-mode 3 --steps 1 --slice off | tee $O/p16384_off.txt
vb --big 16384 --big-mode 3 --steps 1 --slice on | tee $O/p16384_on64.txt
