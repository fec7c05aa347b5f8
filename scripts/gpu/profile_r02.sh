#!/bin/bash
# Round-2 evidence run: the whole GPU suite, the bench line, rocprofv3 kernel stats of the same command, the PMC passes
# (each its own run, --kernel-trace --pmc only) and the secondary kernels. Outputs under gpurun_out/r02/.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout -k 10 900 python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; head -c 600 $O/bench_n1.json; tail -3 $O/bench_n1.err
AVP_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --steps 3 --warmup 1 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; head -c 400 $O/bench_force_dist.json
(cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err)
for pass in "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "lds:SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc/$name --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --pmc-mode > $O/pmc_$name.log 2>&1)
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc/$name --output-format csv -- python $R/scripts/bench_check.py --iters 3 > $O/pmc_check_$name.log 2>&1)
done
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_check --output-format csv -- python $R/scripts/bench_check.py --iters 10 > $O/bench_check.jsonl 2> $O/stats_check.err)
python scripts/pmc_r02_summary.py $O/pmc > $O/pmc_summary.json; head -c 400 $O/pmc_summary.json
python scripts/variant_bench.py --big 2048 > $O/phase_profile.json 2> $O/phase_profile.err
timeout -k 10 300 python scripts/look_bench.py > $O/lookahead.json 2> $O/lookahead.err; head -c 300 $O/lookahead.json
find $O -name "*kernel_stats.csv" | head; du -sh $O
