#!/bin/bash
# round 3, GPU call O: first-slice length of the time-sliced forms
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_plan_wave.py -m gpu -x -q -k "time_sliced" 2>&1 | tail -4 > $O/pytest.txt
tail -n 2 $O/pytest.txt
vb() { timeout 300 python scripts/variant_bench.py --no-profile "$@" 2>$O/err.txt | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d.get(k) for k in ('big_n','big_mode','time_sliced','slice_pops','slice_first','big_ms','big_expansions_per_s','big_digest')})"; tail -n 1 $O/err.txt | grep -v amdgpu.ids; }
for f in 4 16 32; do vb --big 4096 --big-mode 4 --steps 2 --slice on --slice-first $f | tee $O/q4096_f$f.txt; done
for f in 4 16 32; do vb --big 16384 --big-mode 2 --steps 1 --slice on --slice-first $f | tee $O/w16384_f$f.txt; done
vb --big 16384 --big-mode 3 --steps 1 --slice on --slice-first 16 | tee $O/p16384_f16.txt
vb --big 32768 --big-mode 2 --steps 1 --slice on --slice-first 16 | tee $O/w32768_f16.txt
vb --big 8192 --big-mode 4 --steps 1 --slice on --slice-first 16 | tee $O/q8192_f16.txt
vb --big 8192 --big-mode 3 --steps 1 --slice on --slice-first 16 | tee $O/p8192_f16.txt
vb --big 8192 --big-mode 2 --steps 1 --slice on --slice-first 16 | tee $O/w8192_f16.txt
