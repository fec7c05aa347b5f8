#!/bin/bash
# round-5 third GPU run: the fused pop loop (PL_FUSE: three of a record pop's six barriers and its node load gone) against the same
# build without it and against round 4's kernels; the circle checker's persistent-tile forms; the host's CPU quota.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r05c; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
V=automatedvaletparking_amd/variants
timeout -k 10 900 python -m pytest tests/test_gpu_check.py tests/test_gpu_lookahead.py tests/test_gpu_plan.py tests/test_gpu_limits.py tests/test_gpu_staged.py -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -4 $O/pytest_sub.log
timeout 300 python scripts/bench_check.py --variants 0 2>/dev/null | grep circle | cut -c1-200 | tee $O/bench_check.jsonl
timeout 300 python scripts/bench_check.py --variants 0 --case 19 --iters 10 2>/dev/null | grep circle | cut -c1-200 | tee -a $O/bench_check.jsonl
for rep in 1 2 3; do
  for v in default nofuse r04base; do
    L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
    echo "== rep $rep $v"; timeout 300 python scripts/variant_bench.py --no-profile --no-big --steps 10 $L 2>/dev/null | tail -1 | cut -c1-200
  done
done 2>&1 | tee $O/ab_c2.log
for v in default nofuse; do
  L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
  echo "== 512 problems $v"; timeout 300 python scripts/variant_bench.py --no-profile --big 512 --big-mode 1 $L 2>/dev/null | tail -1 | cut -c200-520
done 2>&1 | tee $O/ab_512.log
timeout 300 python scripts/look_bench.py > $O/lookahead.json 2> $O/lookahead.err; head -c 2600 $O/lookahead.json | tail -c 1500
echo; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)  affinity: $(python -c 'import os; print(len(os.sched_getaffinity(0)))')  load: $(cat /proc/loadavg)" | tee $O/cpu_quota.txt
grep -c processor /proc/cpuinfo | tee -a $O/cpu_quota.txt; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8 | tee -a $O/cpu_quota.txt
