#!/bin/bash
# round-5 first GPU run: the whole GPU suite on the reworked kernels (collision pass: any footprint width, 13-bit cells; circle
# checker on the distance kernel's skeleton; plan_kernel's phases as called functions + the whole-wave sample bookkeeping; the
# world-2 real-planner test), then A/Bs in ONE box: round-4 kernels vs today's vs today's without the called resolution / book,
# the circle kernel's two variants, the partial LDS heap at caps where the whole open list fits it, one bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05a; rm -rf $O; mkdir -p $O
V=automatedvaletparking_amd/variants
timeout -k 10 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 300 python scripts/bench_check.py --variants 0 2>/dev/null | cut -c1-260 | tee $O/bench_check.jsonl
for rep in 1 2; do
  for v in r04base default nocall; do
    L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
    echo "== rep $rep $v"; timeout 300 python scripts/variant_bench.py --no-profile --no-big $L 2>/dev/null | tail -1 | cut -c1-300
  done
done 2>&1 | tee $O/ab_c2.log
for cap in 300 150; do for v in default heaplds; do
  L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
  echo "== cap $cap $v"; timeout 300 python scripts/variant_bench.py --no-profile --no-big --cap $cap $L 2>/dev/null | tail -1 | cut -c1-300
done; done 2>&1 | tee $O/ab_heaplds.log
for v in r04base default; do
  L=""; [ $v != default ] && L="--lib $V/libavp_hip_$v.so"
  echo "== big 4096 quad $v"; timeout 300 python scripts/variant_bench.py --no-profile --big 4096 --big-mode 4 $L 2>/dev/null | tail -1 | cut -c1-420
  echo "== big 16384 wave $v"; timeout 300 python scripts/variant_bench.py --no-profile --big 16384 --big-mode 2 $L 2>/dev/null | tail -1 | cut -c1-420
done 2>&1 | tee $O/ab_big.log
timeout 300 python scripts/variant_bench.py --no-big > $O/phase_profile.json 2> $O/phase_profile.err; cut -c1-1500 $O/phase_profile.json
timeout 300 python scripts/look_bench.py > $O/lookahead.json 2> $O/lookahead.err; head -c 1500 $O/lookahead.json
timeout -k 10 900 python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; head -c 700 $O/bench_n1.json; tail -3 $O/bench_n1.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05a/bench_n1.json"))
print(json.dumps({k: d.get(k) for k in ("value", "ms_per_step", "expansions_per_s", "cpu_baseline", "cpu_baseline_all_cores")})[:1800])
print(json.dumps(d.get("roofline_check", {}))[:600])
PY
