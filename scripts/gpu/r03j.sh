#!/bin/bash
# round 3, GPU call J: sweep loop with PL_SWEEP_U pairs per lane and trip -- h-field parity, form parity, timing
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hfield.py tests/test_gpu_plan.py tests/test_gpu_plan_wave.py tests/test_gpu_staged.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest_sweep.txt
tail -n 3 $O/pytest_sweep.txt
for mode in 1 4; do timeout 300 python scripts/variant_bench.py --big 4096 --big-mode $mode --no-profile --steps 2 > $O/vb_4096_m$mode.json 2>/dev/null; done
for mode in 2; do timeout 300 python scripts/variant_bench.py --big 16384 --big-mode $mode --no-profile --steps 1 > $O/vb_16384_m$mode.json 2>/dev/null; done
for mode in 2 4; do timeout 300 python scripts/wave_profile.py --n 4096 --mode $mode > $O/wp_4096_m$mode.json 2>/dev/null; done
timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_head.json 2>/dev/null
cat $O/vb_*.json $O/wp_*.json $O/bench_head.json
