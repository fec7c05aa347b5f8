#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2l
for cfg in "4096 1" "4096 2" "16384 1" "16384 2"; do
set -- $cfg
timeout -k 10 200 python scripts/variant_bench.py --big $1 --big-mode $2 --no-profile --steps 1 > gpurun_out/r2l/vb_$1_$2.json 2> gpurun_out/r2l/vb_$1_$2.err
cat gpurun_out/r2l/vb_$1_$2.json | cut -c150-600; tail -2 gpurun_out/r2l/vb_$1_$2.err
done
timeout -k 10 800 python -m pytest tests/test_gpu_plan_wave.py -q -m gpu > gpurun_out/r2l/pytest_wave.log 2>&1; tail -25 gpurun_out/r2l/pytest_wave.log
