#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2c
for v in o2 o12; do
python scripts/variant_bench.py --lib automatedvaletparking_amd/variants/libavp_hip_$v.so > gpurun_out/r2c/vb_$v.json 2> gpurun_out/r2c/vb_$v.err
done
cat gpurun_out/r2c/vb_*.json; tail -3 gpurun_out/r2c/vb_*.err
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r2c/pytest_gpu.log 2>&1; tail -8 gpurun_out/r2c/pytest_gpu.log
