#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2k
timeout -k 10 200 python scripts/variant_bench.py --big 2048 --big-mode 2 --no-profile > gpurun_out/r2k/vb_wave.json 2> gpurun_out/r2k/vb_wave.err
cat gpurun_out/r2k/vb_wave.json; tail -5 gpurun_out/r2k/vb_wave.err
timeout -k 10 600 python -m pytest tests/test_gpu_plan_wave.py -x -q -m gpu > gpurun_out/r2k/pytest_wave.log 2>&1; tail -25 gpurun_out/r2k/pytest_wave.log
