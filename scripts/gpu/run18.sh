#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 120 python scripts/bench_check.py --iters 10 2>/dev/null | cut -c1-190
timeout -k 10 120 python scripts/bench_check.py --iters 10 --variants 0 --case 19 2>/dev/null | grep 'distance\|corridor' | cut -c1-190
timeout -k 10 600 python -m pytest tests/test_gpu_check.py tests/test_gpu_corridor.py tests/test_gpu_split.py -q -m gpu 2>&1 | tail -3
