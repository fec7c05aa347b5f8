#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05h; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_edge_inputs.py tests/test_gpu_limits.py tests/test_gpu_rs.py tests/test_gpu_errors.py -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -15 $O/pytest_sub.log
