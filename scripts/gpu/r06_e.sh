#!/bin/bash
# round 6: group forms -- parity first, then working tree vs the committed HEAD build (same box, alternating)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06e; rm -rf $O; mkdir -p $O
V=automatedvaletparking_amd/variants
timeout -k 10 900 python -m pytest tests/test_gpu_plan_wave.py tests/test_gpu_staged.py -q -m gpu -x > $O/pytest_wave.log 2>&1; tail -2 $O/pytest_wave.log
for rep in 1 2; do for v in default head; do
  L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
  for cfgs in "4096 4" "4096 3" "16384 3"; do set -- $cfgs
    echo "== $rep $v n=$1 mode=$2 $(timeout 300 python scripts/variant_bench.py $L --no-profile --big $1 --big-mode $2 --steps 2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d.get("big_ms"), d.get("big_digest"))')"
  done
done; done 2>&1 | tee $O/sweep.log
for mode in 3 4; do timeout -k 10 300 python scripts/wave_profile.py --n 4096 --mode $mode 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["mode"], d["capped_cycles_per_pop"], {k: round(v) for k, v in d["capped_per_pop"].items()})'; done
