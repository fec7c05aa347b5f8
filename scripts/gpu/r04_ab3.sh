#!/bin/bash
# parity subset ($1 = pytest selection) first, then scripts/gpu/r04_ab2.sh with the remaining arguments (variant names)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04ab; mkdir -p $O
timeout -k 10 900 python -m pytest $1 -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -5 $O/pytest_sub.log
shift
bash scripts/gpu/r04_ab2.sh "$@"
