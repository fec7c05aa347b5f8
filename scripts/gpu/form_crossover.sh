#!/bin/bash
# which kernel form at which batch size (the table in plan_pick_mode, csrc/avp_capi_plan.inc): 8 ... 128 problems per CU, auto time slicing
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/form_crossover; mkdir -p $O
vb() { timeout 300 python scripts/variant_bench.py --no-profile "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d.get(k) for k in ('big_n','big_mode','time_sliced','big_ms','big_expansions_per_s','big_digest')})"; }
for cfgs in "2048 1" "2048 4" "3072 1" "3072 4" "6144 4" "6144 3" "8192 4" "8192 3" "12288 3" "12288 2" "24576 3" "24576 2" "32768 3" "32768 2"; do set -- $cfgs
  vb --big $1 --big-mode $2 --steps 1 --slice auto | tee -a $O/forms.txt
done
