#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "" w5q4k w6q2k w8q1k w8q512; do
  if [ -z "$v" ]; then unset AVP_HIP_LIB; else export AVP_HIP_LIB=$PWD/automatedvaletparking_amd/variants/libavp_hip_$v.so; fi
  echo "== ${v:-default}"; timeout -k 10 120 python scripts/bench_check.py --iters 10 --variants 0 2>/dev/null | grep '"distance"' | cut -c1-170
  timeout -k 10 120 python scripts/bench_check.py --iters 10 --variants 0 --case 19 2>/dev/null | grep '"distance"' | cut -c1-170
  timeout -k 10 300 python -m pytest tests/test_gpu_check.py -q -m gpu -k "random_poses or golden_collision or edge_inputs or c4" 2>&1 | tail -1
done
