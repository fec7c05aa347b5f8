#!/bin/bash
# round 3, GPU call H: plan_kernel with its cold / multi-site parts called (0 VGPR spills): parity + headline timing
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03h; mkdir -p $O
timeout 600 python scripts/variant_bench.py --big 4096 --big-mode 1 --steps 5 > $O/vb_4096_m1.json 2> $O/vb.err; cat $O/vb_4096_m1.json
timeout 300 python scripts/look_bench.py > $O/lookahead.json 2> $O/lookahead.err; head -c 700 $O/lookahead.json
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -n 4 $O/pytest_gpu.log
