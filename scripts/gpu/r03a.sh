#!/bin/bash
# round 3, GPU call A: the restructured wave form (called phases) -- parity on the default build, timing of the 8 / 12 / 16
# waves-per-CU variants, per-phase cycles
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03a; mkdir -p $O
V=automatedvaletparking_amd/variants
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_default.txt
for v in default w12 w16; do
  L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
  timeout 300 python scripts/variant_bench.py $L --big 16384 --big-mode 2 --no-profile --steps 3 > $O/vb_${v}_16384_m2.json 2> $O/vb_${v}_16384_m2.err
  timeout 300 python scripts/variant_bench.py $L --big 4096 --big-mode 2 --no-profile --steps 3 > $O/vb_${v}_4096_m2.json 2> $O/vb_${v}_4096_m2.err
  timeout 300 python scripts/wave_profile.py $L --n 4096 > $O/wp_${v}_4096.json 2> $O/wp_${v}_4096.err
done
timeout 300 python scripts/variant_bench.py --big 4096 --big-mode 1 --steps 3 > $O/vb_default_4096_m1.json 2> $O/vb_default_4096_m1.err
timeout 300 python scripts/wave_profile.py --n 16384 > $O/wp_default_16384.json 2> $O/wp_default_16384.err
AVP_HIP_LIB=$V/libavp_hip_w16.so timeout 600 python -m pytest tests/test_gpu_plan_wave.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest_w16.txt
AVP_HIP_LIB=$V/libavp_hip_w12.so timeout 600 python -m pytest tests/test_gpu_plan_wave.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest_w12.txt
tail -3 $O/pytest_default.txt $O/pytest_w16.txt $O/pytest_w12.txt; cat $O/vb_*.json $O/wp_*.json
