#!/bin/bash
# round 3, GPU call Q: soak of the time-sliced forms' hand-over (short slices, 100 launches per form) + its pytest form
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 600 python scripts/slice_soak.py --launches 100 --slice-pops 4 > $O/slice_soak.json 2> $O/slice_soak.err; tail -n 2 $O/slice_soak.err; cat $O/slice_soak.json
timeout 600 python -m pytest tests/test_gpu_plan_wave.py -m gpu -x -q -k "soak or time_sliced" 2>&1 | tail -3
