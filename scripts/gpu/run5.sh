#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2e
python scripts/variant_bench.py > gpurun_out/r2e/vb_default.json 2> gpurun_out/r2e/vb_default.err
cat gpurun_out/r2e/vb_*.json; tail -3 gpurun_out/r2e/vb_*.err
timeout 1500 python -m pytest tests -q -m gpu -k "plan or configs or hfield" > gpurun_out/r2e/pytest_gpu.log 2>&1; tail -6 gpurun_out/r2e/pytest_gpu.log
export TMPDIR=/tmp; R=$PWD
for pass in "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "lds:SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d "$R/gpurun_out/r2e/pmc/$name" --output-format csv -- python "$R/bench.py" --steps 3 --warmup 1 --pmc-mode > "$R/gpurun_out/r2e/pmc_$name.log" 2>&1)
  tail -2 gpurun_out/r2e/pmc_$name.log | cut -c1-300
done
python scripts/pmc_r02_summary.py gpurun_out/r2e/pmc > gpurun_out/r2e/pmc_summary.json; head -c 1500 gpurun_out/r2e/pmc_summary.json
