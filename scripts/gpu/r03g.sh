#!/bin/bash
# round 3, GPU call G: trial of the rewritten bench.py (N = 1 and the N > 1 code path through RCCL with world size 1) + the whole GPU suite
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03g; mkdir -p $O
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; tail -n 5 $O/bench_n1.err; head -c 1500 $O/bench_n1.json
AVP_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 2 --warmup 1 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; tail -n 5 $O/bench_force_dist.err; head -c 1200 $O/bench_force_dist.json
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -n 8 $O/pytest_gpu.log
