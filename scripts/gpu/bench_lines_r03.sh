#!/bin/bash
# the two bench lines of the round-3 evidence alone (after host-side changes that leave the kernel sources untouched)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; head -c 300 $O/bench_n1.json; tail -3 $O/bench_n1.err
AVP_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --steps 3 --warmup 1 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; head -c 300 $O/bench_force_dist.json; tail -3 $O/bench_force_dist.err
