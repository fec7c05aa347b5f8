#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/look
rm -f gpurun_out/look/ab.log
timeout 240 python scripts/look_bench.py 256 1000 2>&1 | grep -v "amdgpu" >> gpurun_out/look/ab.log
timeout 240 python scripts/look_bench.py profile 2>&1 | grep -v "amdgpu" >> gpurun_out/look/ab.log
cat gpurun_out/look/ab.log
timeout 900 python -m pytest tests/test_gpu_lookahead.py tests/test_gpu_plan_wave.py -x -q 2>&1 | tail -n 8 | tee gpurun_out/look/pytest.log
