#!/bin/bash
# GPU session 2: phase breakdown with the finer timers, the 256-thread variant, the fixed tests
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2b
python scripts/variant_bench.py > gpurun_out/r2b/vb_default.json 2> gpurun_out/r2b/vb_default.err
python scripts/variant_bench.py --lib automatedvaletparking_amd/variants/libavp_hip_t256.so > gpurun_out/r2b/vb_t256.json 2> gpurun_out/r2b/vb_t256.err
cat gpurun_out/r2b/vb_*.json; tail -3 gpurun_out/r2b/vb_*.err
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_errors.py -q -m gpu > gpurun_out/r2b/pytest.log 2>&1; tail -5 gpurun_out/r2b/pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err; cat gpurun_out/r2b/bench.json; tail -5 gpurun_out/r2b/bench.err
AVP_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r2b/bench_dist.json 2> gpurun_out/r2b/bench_dist.err; cat gpurun_out/r2b/bench_dist.json; tail -5 gpurun_out/r2b/bench_dist.err
