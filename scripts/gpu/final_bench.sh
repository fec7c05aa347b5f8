#!/bin/bash
# The bench line with the stamped PMC summary in place + rocprofv3 kernel stats of the headline-only form of the command.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r02b; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; python -c "import json; d=json.load(open('$O/bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_check']['frac'])"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/stats_headline --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --pmc-mode > $O/bench_headline_under_rocprof.json 2> $O/stats_headline.err)
AVP_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --steps 3 --warmup 1 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; wc -l $O/bench_force_dist.json
find $O -name "*kernel_stats.csv"
