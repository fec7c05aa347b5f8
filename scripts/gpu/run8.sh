#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2h
python scripts/variant_bench.py --big 256 > gpurun_out/r2h/vb_default.json 2> gpurun_out/r2h/vb_default.err
cat gpurun_out/r2h/vb_*.json; tail -3 gpurun_out/r2h/vb_*.err
python scripts/bench_check.py --iters 10 > gpurun_out/r2h/bench_check.jsonl 2>&1; cat gpurun_out/r2h/bench_check.jsonl
