#!/bin/bash
# round 3, GPU call L: wave-parallel heap pop + rotated wave roles -- parity (plan / group forms / staged / lookahead), A/B against
# the builds without each (digests must agree)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_plan_wave.py tests/test_gpu_staged.py tests/test_gpu_lookahead.py -m gpu -x -q --deselect tests/test_gpu_lookahead.py::test_lookahead_soak_300_launches 2>&1 | tail -6 > $O/pytest.txt
tail -n 3 $O/pytest.txt
for v in "" norot serialpop; do
  L=""; [ -n "$v" ] && L="--lib automatedvaletparking_amd/variants/libavp_hip_$v.so"
  timeout 200 python scripts/variant_bench.py $L --big 4096 --big-mode 4 --no-profile --steps 2 > $O/v_${v:-default}.4096_m4.json 2>/dev/null
  timeout 200 python scripts/variant_bench.py $L --big 16384 --big-mode 2 --no-profile --steps 1 > $O/v_${v:-default}.16384_m2.json 2>/dev/null
done
timeout 200 python scripts/variant_bench.py --big 4096 --big-mode 1 --no-profile --steps 2 > $O/v_default.4096_m1.json 2>/dev/null
for mode in 4; do timeout 300 python scripts/wave_profile.py --n 4096 --mode $mode > $O/wp_4096_m$mode.json 2>/dev/null; done
cat $O/v_*.json $O/wp_*.json
