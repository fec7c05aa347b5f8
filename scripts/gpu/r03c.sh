#!/bin/bash
# round 3, GPU call C: 16 waves per CU as the default build; wave / pair / quad forms and the staged call -- parity,
# determinism of the digests over repeated launches, timing on the 4 096 and 16 384 batches
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c; mkdir -p $O
V=automatedvaletparking_amd/variants
timeout 1200 python -m pytest tests/test_gpu_plan_wave.py tests/test_gpu_staged.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest_forms.txt
tail -n 3 $O/pytest_forms.txt
for mode in 1 2 3 4 16; do
  timeout 300 python scripts/variant_bench.py --big 4096 --big-mode $mode --no-profile --steps 2 > $O/vb_4096_m$mode.json 2> $O/vb_4096_m$mode.err
done
for mode in 2 3 4 16; do
  timeout 300 python scripts/variant_bench.py --big 16384 --big-mode $mode --no-profile --steps 2 > $O/vb_16384_m$mode.json 2> $O/vb_16384_m$mode.err
done
for rep in 1 2; do for mode in 3 4; do
  timeout 300 python scripts/variant_bench.py --big 16384 --big-mode $mode --no-profile --steps 1 > $O/vb_16384_m${mode}_rep$rep.json 2>/dev/null
done; done
timeout 300 python scripts/variant_bench.py --lib $V/libavp_hip_w8.so --big 4096 --big-mode 16 --no-profile --steps 2 > $O/vb_w8_4096_m16.json 2>/dev/null
timeout 300 python scripts/variant_bench.py --lib $V/libavp_hip_w8.so --big 4096 --big-mode 4 --no-profile --steps 2 > $O/vb_w8_4096_m4.json 2>/dev/null
timeout 300 python scripts/wave_profile.py --n 4096 --mode 4 > $O/wp_4096_m4.json 2> $O/wp_4096_m4.err
timeout 300 python scripts/wave_profile.py --n 4096 --mode 3 > $O/wp_4096_m3.json 2> $O/wp_4096_m3.err
cat $O/vb_*.json $O/wp_*.json
