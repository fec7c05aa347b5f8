#!/bin/bash
# round 3, GPU call F: round-based shot checks with growing chunks, relax pre-check as the default
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plan_wave.py tests/test_gpu_staged.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest_forms.txt
tail -n 3 $O/pytest_forms.txt
for mode in 3 4 16; do timeout 300 python scripts/variant_bench.py --big 4096 --big-mode $mode --no-profile --steps 2 > $O/vb_4096_m$mode.json 2>/dev/null; done
for mode in 2 3; do timeout 300 python scripts/variant_bench.py --big 16384 --big-mode $mode --no-profile --steps 1 > $O/vb_16384_m$mode.json 2>/dev/null; done
timeout 300 python scripts/variant_bench.py --big 8192 --big-mode 3 --no-profile --steps 1 > $O/vb_8192_m3.json 2>/dev/null
for mode in 2 3 4; do timeout 300 python scripts/wave_profile.py --n 4096 --mode $mode > $O/wp_4096_m$mode.json 2>/dev/null; done
cat $O/vb_*.json $O/wp_*.json
