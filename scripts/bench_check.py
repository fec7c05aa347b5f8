"""Micro-benchmark of the footprint-collision kernels (checks/s and algorithmic GB/s)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", type=int, default=1)
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--variants", default="0,1")
    a = ap.parse_args()
    import torch
    from automatedvaletparking_amd import costmap, config, _native
    cfg = config.default_config()
    veh = costmap.Vehicle()
    m = costmap.Map(file=os.path.join(ROOT, "data", "BenchmarkCases", f"Case{a.case}.csv"), discrete_size=cfg["map_discrete_size"])
    dm = _native.DeviceMap(m, veh, cfg)
    rng = np.random.default_rng(0)
    b = m.boundary
    poses = np.stack([rng.uniform(b[0] + 6, b[1] - 6, a.n), rng.uniform(b[2] + 6, b[3] - 6, a.n), rng.uniform(-np.pi, np.pi, a.n)], 0)
    t = dm.dev_tensor(poses)
    out = dm.empty(a.n, torch.uint8)
    bytes_per = 16 * dm.P + 25
    for kind, variant in [(0, int(v)) for v in a.variants.split(",")] + [(1, 0)]:
        iters = a.iters if not (kind == 0 and variant == 1) else max(2, a.iters // 10)
        dm.check_batch_dev(t[0], t[1], t[2], out=out, kind=kind, variant=variant)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            dm.check_batch_dev(t[0], t[1], t[2], out=out, kind=kind, variant=variant)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        rate = a.n / (ms * 1e-3)
        print(json.dumps({"kernel": ["distance", "circle"][kind], "variant": variant, "case": a.case, "P": dm.P, "n": a.n,
                          "ms": ms, "checks_per_s": rate, "algorithmic_GBps": rate * bytes_per / 1e9,
                          "frac_of_8TBps": rate * bytes_per / 8e12, "collide_frac": float(out.float().mean().item())}))


if __name__ == "__main__":
    main()
