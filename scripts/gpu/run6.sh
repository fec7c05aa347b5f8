#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2f
python scripts/variant_bench.py --big 256 > gpurun_out/r2f/vb_default.json 2> gpurun_out/r2f/vb_default.err
cat gpurun_out/r2f/vb_*.json; tail -3 gpurun_out/r2f/vb_*.err
timeout 600 python -m pytest tests/test_gpu_corridor.py tests/test_gpu_check.py -q -m gpu > gpurun_out/r2f/pytest.log 2>&1; tail -6 gpurun_out/r2f/pytest.log
python scripts/bench_check.py --iters 10 > gpurun_out/r2f/bench_check.jsonl 2>&1; cat gpurun_out/r2f/bench_check.jsonl
