#!/bin/bash
# GPU session 1 (round 2): A/B of kernel variants, the whole GPU suite, counter list + I-cache / SQ PMC passes.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
python scripts/variant_bench.py > gpurun_out/r2a/vb_default.json 2> gpurun_out/r2a/vb_default.err
python scripts/variant_bench.py --lib automatedvaletparking_amd/variants/libavp_hip_calls.so > gpurun_out/r2a/vb_calls.json 2> gpurun_out/r2a/vb_calls.err
cat gpurun_out/r2a/vb_*.json; tail -3 gpurun_out/r2a/vb_*.err
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r2a/pytest_gpu.log 2>&1; tail -15 gpurun_out/r2a/pytest_gpu.log
(cd /tmp && rocprofv3 -L > "$R/gpurun_out/r2a/counters.txt" 2>&1)
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  tag=$(echo $c | cut -d' ' -f1)
  for v in default calls; do
    lib=""; [ $v = calls ] && lib="--lib $R/automatedvaletparking_amd/variants/libavp_hip_calls.so"
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$R/gpurun_out/r2a/pmc_${tag}_$v" --output-format csv -- python "$R/scripts/variant_bench.py" $lib --steps 2 --no-profile --big 256 > "$R/gpurun_out/r2a/pmc_${tag}_$v.log" 2>&1)
  done
done
ls gpurun_out/r2a; du -sh gpurun_out/r2a
