#!/bin/bash
# round-4 working run: a subset of the GPU suite ($1 = pytest -k / file selection), the secondary-kernel rates, the bench summary.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04c; mkdir -p $O
timeout -k 10 900 python -m pytest $1 -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -4 $O/pytest_sub.log
timeout 300 python scripts/bench_check.py --variants 0 2>/dev/null | cut -c1-230
bash scripts/gpu/r04_bench.sh
