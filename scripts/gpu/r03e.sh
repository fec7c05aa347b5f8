#!/bin/bash
# round 3, GPU call E: LDS-only group barriers + list-scheduled RS rounds (default build), relax pre-check variant
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03e; mkdir -p $O
V=automatedvaletparking_amd/variants
timeout 900 python -m pytest tests/test_gpu_plan_wave.py tests/test_gpu_staged.py tests/test_gpu_hfield.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest_forms.txt
tail -n 3 $O/pytest_forms.txt
for v in default precheck; do
  L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
  for mode in 4 16; do
    timeout 300 python scripts/variant_bench.py $L --big 4096 --big-mode $mode --no-profile --steps 2 > $O/vb_${v}_4096_m$mode.json 2>/dev/null
  done
  for mode in 2 3; do
    timeout 300 python scripts/variant_bench.py $L --big 16384 --big-mode $mode --no-profile --steps 1 > $O/vb_${v}_16384_m$mode.json 2>/dev/null
  done
  timeout 300 python scripts/variant_bench.py $L --big 8192 --big-mode 3 --no-profile --steps 1 > $O/vb_${v}_8192_m3.json 2>/dev/null
  timeout 300 python scripts/wave_profile.py $L --n 4096 --mode 4 > $O/wp_${v}_4096_m4.json 2>/dev/null
done
timeout 300 python scripts/variant_bench.py --big 8192 --big-mode 2 --no-profile --steps 1 > $O/vb_default_8192_m2.json 2>/dev/null
timeout 300 python scripts/variant_bench.py --big 8192 --big-mode 4 --no-profile --steps 1 > $O/vb_default_8192_m4.json 2>/dev/null
timeout 300 python scripts/variant_bench.py --big 2048 --big-mode 4 --no-profile --steps 2 > $O/vb_default_2048_m4.json 2>/dev/null
timeout 300 python scripts/variant_bench.py --big 2048 --big-mode 1 --no-profile --steps 2 > $O/vb_default_2048_m1.json 2>/dev/null
cat $O/vb_*.json $O/wp_*.json
