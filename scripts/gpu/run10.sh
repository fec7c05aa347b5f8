#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2j
python scripts/variant_bench.py --big 2048 > gpurun_out/r2j/vb_default.json 2> gpurun_out/r2j/vb_default.err
cat gpurun_out/r2j/vb_*.json; tail -3 gpurun_out/r2j/vb_*.err
timeout 900 python -m pytest tests/test_gpu_plan.py -q -m gpu -x -k "batch_256 or golden_problems" > gpurun_out/r2j/pytest.log 2>&1; tail -4 gpurun_out/r2j/pytest.log
