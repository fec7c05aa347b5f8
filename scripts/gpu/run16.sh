#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2p
for v in "" chk4 chk16; do
  if [ -z "$v" ]; then unset AVP_HIP_LIB; else export AVP_HIP_LIB=$PWD/automatedvaletparking_amd/variants/libavp_hip_$v.so; fi
  echo "== ${v:-default8}"; timeout -k 10 120 python scripts/bench_check.py --iters 10 --variants 0 2>/dev/null | grep '"distance"\|circle' | cut -c1-200
  timeout -k 10 120 python scripts/bench_check.py --iters 10 --variants 0 --case 19 2>/dev/null | grep '"distance"' | cut -c1-200
done
unset AVP_HIP_LIB
timeout -k 10 600 python -m pytest tests/test_gpu_check.py -q -m gpu 2>&1 | tail -3
