#!/bin/bash
# round-5 lookahead parameter sweep (compile-time variants), one box: config[1] ms per batch, digest must not change.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05e; rm -rf $O; mkdir -p $O
V=automatedvaletparking_amd/variants
for rep in 1 2; do
  for v in default $VARS; do
    L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
    echo "== rep $rep $v"; timeout 300 python scripts/variant_bench.py --no-profile --no-big --steps 8 $L 2>/dev/null | tail -1 | cut -c1-200
  done
done 2>&1 | tee $O/sweep.log
