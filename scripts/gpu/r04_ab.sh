#!/bin/bash
# A/B of the working build against variants/libavp_hip_base.so in ONE call (boxes differ by +-8 %): the group-form parity
# tests first, then the headline batch and the big batches in each form, alternating the two libraries.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04ab; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_plan_wave.py tests/test_gpu_plan.py -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -4 $O/pytest_sub.log
BASE=automatedvaletparking_amd/variants/libavp_hip_base.so
for rep in 1 2; do
  for mode in 4 3 2; do
    big=4096; [ $mode = 2 ] && big=16384
    echo "== rep $rep mode $mode big $big: base / new"
    timeout 300 python scripts/variant_bench.py --no-profile --lib $BASE --big $big --big-mode $mode 2>/dev/null | tail -2 | cut -c1-300
    timeout 300 python scripts/variant_bench.py --no-profile --big $big --big-mode $mode 2>/dev/null | tail -2 | cut -c1-300
  done
done 2>&1 | tee $O/ab.log
