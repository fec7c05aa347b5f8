#!/bin/bash
# round 3, GPU call T: the sampler's bookkeeping by a whole wave in the group forms -- parity, timing
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plan_wave.py tests/test_gpu_staged.py tests/test_gpu_workloads.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest.txt
tail -n 4 $O/pytest.txt
vb() { timeout 300 python scripts/variant_bench.py --no-profile "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d.get(k) for k in ('lib','big_n','big_mode','time_sliced','big_ms','big_expansions_per_s','big_digest')})"; }
vb --big 4096 --big-mode 4 --steps 2 | tee -a $O/ab.txt
vb --big 16384 --big-mode 3 --steps 1 | tee -a $O/ab.txt
vb --big 16384 --big-mode 2 --steps 1 | tee -a $O/ab.txt
timeout 300 python scripts/wave_profile.py --n 4096 --mode 4 2>/dev/null | tee $O/wp_m4.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
