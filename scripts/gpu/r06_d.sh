#!/bin/bash
# round 6: group forms before / after the lane-per-pose form became a called function (same box, alternating)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06d; rm -rf $O; mkdir -p $O
V=automatedvaletparking_amd/variants
for rep in 1 2; do for v in default head; do
  L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
  for cfgs in "4096 4" "16384 3" "16384 2"; do set -- $cfgs
    echo "== $rep $v n=$1 mode=$2 $(timeout 300 python scripts/variant_bench.py $L --no-profile --big $1 --big-mode $2 --steps 2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d.get("c2_ms"), d.get("big_ms"), d.get("big_digest"))')"
  done
done; done 2>&1 | tee $O/sweep.log
