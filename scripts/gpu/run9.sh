#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2i
python scripts/variant_bench.py --big 2048 > gpurun_out/r2i/vb_default.json 2> gpurun_out/r2i/vb_default.err
cat gpurun_out/r2i/vb_*.json; tail -3 gpurun_out/r2i/vb_*.err
timeout 900 python -m pytest tests/test_gpu_plan.py -q -m gpu -x > gpurun_out/r2i/pytest.log 2>&1; tail -6 gpurun_out/r2i/pytest.log
