#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 120 python scripts/bench_check.py --iters 10 --variants 0 2>/dev/null | grep rs_optimal | cut -c1-190
timeout -k 10 600 python -m pytest tests/test_gpu_rs.py -q -m gpu 2>&1 | tail -3
