#!/bin/bash
# round 3, GPU call B: the pair form (two waves per problem) -- parity, timing of mode 2 / 3 at 8 / 12 / 16 waves per CU,
# per-phase cycles, SQ counters of the saturated wave form at 16 waves per CU
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03b; mkdir -p $O
V=automatedvaletparking_amd/variants
timeout 900 python -m pytest tests/test_gpu_plan_wave.py tests/test_gpu_hfield.py tests/test_gpu_plan.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest_default.txt
for v in default w12 w16; do
  L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
  for mode in 2 3; do
    timeout 300 python scripts/variant_bench.py $L --big 16384 --big-mode $mode --no-profile --steps 2 > $O/vb_${v}_16384_m$mode.json 2> $O/vb_${v}_16384_m$mode.err
    timeout 300 python scripts/variant_bench.py $L --big 4096 --big-mode $mode --no-profile --steps 2 > $O/vb_${v}_4096_m$mode.json 2> $O/vb_${v}_4096_m$mode.err
  done
  timeout 300 python scripts/wave_profile.py $L --n 4096 --mode 3 > $O/wp_${v}_4096_m3.json 2> $O/wp_${v}_4096_m3.err
done
timeout 300 python scripts/wave_profile.py --lib $V/libavp_hip_w16.so --n 4096 --mode 2 > $O/wp_w16_4096_m2.json 2> $O/wp_w16_4096_m2.err
timeout 300 python scripts/variant_bench.py --lib $V/libavp_hip_w16.so --big 1024 --big-mode 3 --no-profile --steps 2 > $O/vb_w16_1024_m3.json 2>/dev/null
timeout 300 python scripts/variant_bench.py --lib $V/libavp_hip_w16.so --big 1024 --big-mode 1 --no-profile --steps 2 > $O/vb_w16_1024_m1.json 2>/dev/null
AVP_HIP_LIB=$V/libavp_hip_w16.so timeout 900 python -m pytest tests/test_gpu_plan_wave.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest_w16.txt
AVP_HIP_LIB=$V/libavp_hip_w12.so timeout 600 python -m pytest tests/test_gpu_plan_wave.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest_w12.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/$O/counters.txt 2>&1
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $pass | cut -d' ' -f1,8 | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $pass -d $R/$O/pmc_$tag -o pmc -- python $R/scripts/variant_bench.py --lib $R/$V/libavp_hip_w16.so --big 16384 --big-mode 2 --no-profile --steps 1 > $R/$O/pmc_$tag.log 2>&1
done
cd $R
tail -n 3 $O/pytest_default.txt $O/pytest_w16.txt $O/pytest_w12.txt; cat $O/vb_*.json $O/wp_*.json; ls $O
