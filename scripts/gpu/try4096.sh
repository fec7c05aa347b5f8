#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/look
timeout 900 python scripts/order_bench.py > gpurun_out/look/order.json 2> gpurun_out/look/order.err; python -c "
import json
for k,v in json.load(open('gpurun_out/look/order.json')).items(): print(k, v)"
tail -n 3 gpurun_out/look/order.err
