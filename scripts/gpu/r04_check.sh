#!/bin/bash
# round-4 working run: the whole GPU suite + the default bench line. Outputs under gpurun_out/r04a/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04a; rm -rf $O; mkdir -p $O
timeout -k 10 1200 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
grep -a "north-star batch" $O/pytest_gpu.log
timeout -k 10 600 python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; head -c 1500 $O/bench_n1.json; tail -3 $O/bench_n1.err
