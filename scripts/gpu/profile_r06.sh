#!/bin/bash
# Round-6 evidence run (one GPU box, ~20 min): the whole GPU suite, the PMC passes (each its own run, --kernel-trace --pmc
# only) on the bench's headline form and on the secondary kernels, the bench line with the stamped PMC summary in place, the
# N > 1 code paths through RCCL with world size 1, rocprofv3 kernel stats of the headline-only form of the command and of
# the secondary kernels, the SQ counters of the saturating batch per kernel form, the per-phase cycles of every form, the
# lookahead A/B and three soak builds, large maps, time slicing A/B. Outputs under gpurun_out/r06/;
# scripts/gpu/collect_r06.sh copies the summaries into profiles/.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r06; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
V=automatedvaletparking_amd/variants
timeout -k 10 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for pass in "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "lds:SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU" "lane:SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc/$name --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --pmc-mode > $O/pmc_$name.log 2>&1)
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc/$name --output-format csv -- python $R/scripts/bench_check.py --iters 3 > $O/pmc_check_$name.log 2>&1)
done
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $O/pmc_icache --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --pmc-mode > $O/pmc_icache.log 2>&1)
python scripts/pmc_counters_dump.py $O/pmc_icache "plan_kernel|check_distance" > $O/pmc_icache.json
python scripts/pmc_summary.py $O/pmc > $O/pmc_summary.json; head -c 300 $O/pmc_summary.json
cp $O/pmc_summary.json profiles/r06_pmc_summary.json       # (on the box only: the bench line below reads the stamped file)
timeout -k 10 900 python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; head -c 600 $O/bench_n1.json; tail -3 $O/bench_n1.err
AVP_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --steps 3 --warmup 1 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; head -c 400 $O/bench_force_dist.json; tail -3 $O/bench_force_dist.err
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/stats_headline --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --pmc-mode > $O/bench_headline_under_rocprof.json 2> $O/stats_headline.err)
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_check --output-format csv -- python $R/scripts/bench_check.py --iters 10 > $O/bench_check.jsonl 2> $O/stats_check.err)
for mode in 1 2 3 4; do
  for pass in "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "lane:SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
    name=${pass%%:*}; ctr=${pass#*:}
    (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_sat/mode$mode/$name --output-format csv -- python $R/scripts/variant_bench.py --big 16384 --big-mode $mode --no-profile --steps 1 > $O/pmc_sat_m${mode}_$name.log 2>&1)
  done
done
python scripts/pmc_sat_summary.py $O/pmc_sat > $O/pmc_saturating_batch.json 2> $O/pmc_sat_summary.err
python scripts/variant_bench.py --big 2048 > $O/phase_profile.json 2> $O/phase_profile.err
for mode in 2 3 4; do timeout -k 10 300 python scripts/wave_profile.py --n 4096 --mode $mode > $O/wave_profile_m$mode.json 2> $O/wave_profile_m$mode.err; done
timeout -k 10 300 python scripts/look_bench.py > $O/lookahead.json 2> $O/lookahead.err; head -c 300 $O/lookahead.json
for v in default look_atomics look_fault5; do
  L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
  [ $v = default -o -f $V/libavp_hip_$v.so ] && timeout -k 10 300 python scripts/look_soak.py $L --launches 300 > $O/soak_$v.json 2> $O/soak_$v.err
done
# the PRODUCT library with a record store of 1 024 entries (avp_plan_set_look_entries): tags collide, entries change hands all the time
timeout -k 10 300 python scripts/look_soak.py --entries-log2 10 --launches 300 > $O/soak_look_small.json 2> $O/soak_look_small.err
# a batch of six problems per CU: helpers only in its tail, owners posting fewer nodes while they are scarce (300 launches, digest of all 1 536 problems)
timeout -k 10 400 python scripts/look_soak.py --n 1536 --launches 300 > $O/soak_look_n1536.json 2> $O/soak_look_n1536.err
timeout -k 10 300 python scripts/look_scale.py 256 384 512 640 768 1024 1280 1536 2048 2560 3000 > $O/lookahead_batch_sizes.jsonl 2> $O/lookahead_batch_sizes.err
timeout -k 10 300 python scripts/look_capped.py 32 64 100 128 160 200 256 400 2>/dev/null | grep '^{' > $O/lookahead_capped_sets.jsonl
timeout -k 10 500 python scripts/large_map_bench.py > $O/large_maps.json 2> $O/large_maps.err
for sl in off on; do for cfgs in "16384 3" "32768 2"; do set -- $cfgs
  timeout -k 10 300 python scripts/variant_bench.py --no-profile --big $1 --big-mode $2 --steps 1 --slice $sl > $O/slice_$1_m$2_$sl.json 2>/dev/null
done; done
timeout -k 10 400 python scripts/slice_soak.py --launches 60 --slice-pops 4 > $O/slice_soak.json 2> $O/slice_soak.err
timeout -k 10 900 python tests/config_sweep.py > $O/config_sweep.jsonl 2> $O/config_sweep.err; cut -c1-300 $O/config_sweep.jsonl
timeout -k 10 400 python scripts/cap_growth.py 300 1000 2000 3000 > $O/cap_growth.jsonl 2> $O/cap_growth.err; cut -c1-300 $O/cap_growth.jsonl
rm -rf $O/pmc/*/*/*.db $O/pmc_sat/*/*/*/*.db $O/pmc_icache/*/*.db 2>/dev/null
find $O -name "*kernel_stats.csv" | head; du -sh $O
