#!/bin/bash
# round 3, GPU call D: per-kernel times of the staged call, SQ counters of the saturated wave form, lookahead soak variants
cd "$GRAFT_REPO_ROOT"
R=$PWD; O=$R/gpurun_out/r03d; mkdir -p $O
V=automatedvaletparking_amd/variants
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_staged4096 --output-format csv -- python $R/scripts/variant_bench.py --big 4096 --big-mode 16 --no-profile --steps 2 > $O/vb_staged4096_rocprof.json 2> $O/stats_staged4096.err)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_staged16384 --output-format csv -- python $R/scripts/variant_bench.py --big 16384 --big-mode 16 --no-profile --steps 1 > $O/vb_staged16384_rocprof.json 2> $O/stats_staged16384.err)
for mode in 2 4; do
  for pass in "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "lane:SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
    name=${pass%%:*}; ctr=${pass#*:}
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_sat/mode$mode/$name --output-format csv -- python $R/scripts/variant_bench.py --big 16384 --big-mode $mode --no-profile --steps 1 > $O/pmc_sat_m${mode}_$name.log 2>&1)
  done
done
python scripts/pmc_sat_summary.py $O/pmc_sat > $O/pmc_saturating_batch.json 2> $O/pmc_sat_summary.err
for v in default look_atomics look_sleep1 look_sleep127 look_wait0 look_wait50k look_fault5; do
  L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
  timeout 300 python scripts/look_soak.py $L --launches 300 > $O/soak_$v.json 2> $O/soak_$v.err
done
timeout 300 python scripts/look_bench.py > $O/lookahead.json 2> $O/lookahead.err
find $O -name "*kernel_stats.csv" | while read f; do echo $f; head -8 $f | cut -c1-200; done
cat $O/soak_*.json; head -c 1500 $O/pmc_saturating_batch.json
