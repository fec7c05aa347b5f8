#!/bin/bash
# round 3, GPU call K: PL_SWEEP_U / pre-check variants of the sweep loop (digests must equal the product library's)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03k; mkdir -p $O
for v in u2 u8 u8np u4np; do
  timeout 200 python scripts/variant_bench.py --lib automatedvaletparking_amd/variants/libavp_hip_$v.so --big 16384 --big-mode 2 --no-profile --steps 1 > $O/$v.16384_m2.json 2>/dev/null
  timeout 200 python scripts/variant_bench.py --lib automatedvaletparking_amd/variants/libavp_hip_$v.so --big 4096 --big-mode 4 --no-profile --steps 2 > $O/$v.4096_m4.json 2>/dev/null
done
cat $O/*.json
