#!/bin/bash
# round 6: corridor kernel variants (queue size / waves per workgroup), bit-exactness first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06c; rm -rf $O; mkdir -p $O
V=automatedvaletparking_amd/variants
timeout -k 10 300 python -m pytest tests/test_gpu_corridor.py -q -m gpu > $O/pytest_cor.log 2>&1; tail -2 $O/pytest_cor.log
for rep in 1 2; do for v in default $VARS; do
  L=""; [ $v != default ] && L="$PWD/$V/libavp_hip_$v.so"
  echo "== $rep $v $(AVP_HIP_LIB=$L timeout 200 python scripts/bench_check.py --iters 20 2>/dev/null | grep corridor | cut -c1-120)"
  [ $v != default ] && [ $rep = 1 ] && (AVP_HIP_LIB=$L timeout 300 python -m pytest tests/test_gpu_corridor.py -q -m gpu 2>&1 | tail -1)
done; done 2>&1 | tee $O/sweep.log
