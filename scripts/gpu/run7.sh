#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2g
scripts/microbench/latency2 | head -12
python scripts/variant_bench.py --big 2048 > gpurun_out/r2g/vb_default.json 2> gpurun_out/r2g/vb_default.err
cat gpurun_out/r2g/vb_*.json; tail -3 gpurun_out/r2g/vb_*.err
timeout 900 python -m pytest tests -q -m gpu -x -k "not configs" > gpurun_out/r2g/pytest.log 2>&1; tail -6 gpurun_out/r2g/pytest.log
