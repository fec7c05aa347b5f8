#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2o
for v in w8 w16; do
timeout -k 10 300 python scripts/variant_bench.py --lib automatedvaletparking_amd/variants/libavp_hip_$v.so --big 16384 --big-mode 2 --no-profile --steps 1 > gpurun_out/r2o/vb_$v.json 2> gpurun_out/r2o/vb_$v.err; echo $v; cut -c130-520 gpurun_out/r2o/vb_$v.json; tail -2 gpurun_out/r2o/vb_$v.err
done
