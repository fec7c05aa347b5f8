"""config[2] (20 BenchmarkCases x 128 pairs, one launch per map on 20 streams) in each kernel form: the launches SHARE the device, so the
form that suits 128 problems alone (one workgroup per problem) is not the one that suits 2 560 problems at once. Prints one JSON line per form."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import torch
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="0,4,3,2")
    a = ap.parse_args()
    sys.argv = sys.argv[:1]
    args = argparse.Namespace(gpus=1, steps=1, warmup=0, workload="auto", no_cpu_baseline=True, no_extras=True, pmc_mode=False)
    b = bench.Bench(args)
    lab, xcfg, xcap, xsets = b.build("c3")
    ref = None
    for mode in [int(v) for v in a.modes.split(",")]:
        xg = [b.group(m, xcfg, st, go, xcap, mode=mode) for (m, st, go) in xsets]
        bench.plan_groups(xg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            xo = bench.plan_groups(xg)
        torch.cuda.synchronize()
        xe = (time.perf_counter() - t0) / 3
        recs = [bench.records(o[0], g.n) for o, g in zip(xo, xg)]
        paths = [o[1].cpu().numpy() for o in xo]
        x = bench.summarize(recs, [g.slots for g in xg], xe, time_sliced=any(g.bp.last_time_sliced for g in xg))
        same = True
        if ref is None:
            ref = (recs, paths)
        else:
            same = all(bench.same_results(ra, pa, rb, pb) for ra, pa, rb, pb in zip(ref[0], ref[1], recs, paths))
        print(json.dumps({"mode": mode, "kernel_form": bench.FORM_NAMES.get(xg[0].mode), "ms": xe * 1e3, "plans_per_s": x["plans_per_s"], "expansions_per_s": x["expansions_per_s"], "identical_to_first": same}), flush=True)
        del xg


if __name__ == "__main__":
    main()
