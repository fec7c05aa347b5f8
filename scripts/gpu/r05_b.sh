#!/bin/bash
# round-5 second GPU run: the circle checker with lane refill (parity + rate against the plain walk), which of the two round-5
# plan_kernel changes costs time (called resolution / whole-wave sample bookkeeping: four builds, alternating), the
# instruction-cache counters of the headline launch, the all-core CPU baseline's scaling with mallopt.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r05b; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
V=automatedvaletparking_amd/variants
timeout -k 10 600 python -m pytest tests/test_gpu_check.py tests/test_gpu_configs.py::test_c4_full tests/test_gpu_edge_inputs.py -q -m gpu -x > $O/pytest_sub.log 2>&1; tail -4 $O/pytest_sub.log
timeout 300 python scripts/bench_check.py --variants 0 2>/dev/null | cut -c1-260 | tee $O/bench_check.jsonl
timeout 300 python scripts/bench_check.py --variants 0 --case 19 --iters 10 2>/dev/null | grep circle | cut -c1-260 | tee -a $O/bench_check.jsonl
for rep in 1 2 3; do
  for v in default c1b0 c0b1 nocall r04base; do
    L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
    echo "== rep $rep $v"; timeout 300 python scripts/variant_bench.py --no-profile --no-big --steps 10 $L 2>/dev/null | tail -1 | cut -c1-200
  done
done 2>&1 | tee $O/ab_c2.log
for pass in "icache:SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "ifetch:SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM"; do
  name=${pass%%:*}; ctr=${pass#*:}
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc/$name --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --pmc-mode > $O/pmc_$name.log 2>&1)
  python scripts/pmc_counters_dump.py $O/pmc/$name "plan_kernel|check_distance" | tee $O/pmc_$name.json
done
python - <<'PY' 2>&1 | tee gpurun_out/r05b/cpu_scaling.txt
import sys, os; sys.path.insert(0, '.')
import numpy as np
from automatedvaletparking_amd import costmap, config, workloads
from oracle import oracle
cfg = config.default_config(); veh = costmap.Vehicle()
m = workloads.case_map(1, cfg)
o0 = oracle.Oracle(m, veh, cfg)
st, go = workloads.sample_pairs(m, lambda p: np.asarray(o0.check_batch(p, kind=0)).astype(bool), 256, np.random.default_rng(20260927))
o = oracle.Oracle(m, veh, cfg, max_pops=1000)
base = None
for t in (1, 8, 32, 64, 128, 256):
    r = o.plan_batch(st, go, threads=t, min_seconds=4.0)
    e = r['pops'] / r['seconds']
    base = base or e
    print(t, 'threads', round(e), 'expansions/s', round(e / base, 1), 'x', r['plans'], 'plans')
PY
rm -rf $O/pmc/*/*/*.db 2>/dev/null; du -sh $O
