#!/bin/bash
# SQ counters of plan_wave_kernel on the saturating batch (16 384 problems, one wave per problem)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r02c; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $O/pmc_wave --output-format csv -- python $R/scripts/variant_bench.py --big 16384 --big-mode 2 --no-profile --steps 1 > $O/pmc_wave.log 2>&1)
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $O/pmc_wg --output-format csv -- python $R/scripts/variant_bench.py --big 16384 --big-mode 1 --no-profile --steps 1 > $O/pmc_wg.log 2>&1)
tail -2 $O/pmc_wave.log | cut -c1-300; ls $O
