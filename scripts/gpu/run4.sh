#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2d
python scripts/variant_bench.py > gpurun_out/r2d/vb_default.json 2> gpurun_out/r2d/vb_default.err
cat gpurun_out/r2d/vb_*.json; tail -3 gpurun_out/r2d/vb_*.err
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r2d/pytest_gpu.log 2>&1; tail -12 gpurun_out/r2d/pytest_gpu.log
