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


    # Reeds-Shepp batch kernel and corridor kernel
    n = 1 << 18
    q0 = np.stack([rng.uniform(b[0] + 6, b[1] - 6, n), rng.uniform(b[2] + 6, b[3] - 6, n), rng.uniform(-np.pi, np.pi, n)], 1)
    q1 = np.stack([rng.uniform(b[0] + 6, b[1] - 6, n), rng.uniform(b[2] + 6, b[3] - 6, n), rng.uniform(-np.pi, np.pi, n)], 1)
    import ctypes as C
    t0, t1 = dm.dev_tensor(q0), dm.dev_tensor(q1)
    st = dm.empty(n, torch.int32); L = dm.empty(n, torch.float64); ty = dm.empty((n, 5), torch.int8); le = dm.empty((n, 5), torch.float64)
    npts = dm.empty(n, torch.int32)
    def rs():
        _native.chk(_native.lib().avp_rs_optimal_batch(dm.h, C.c_void_p(t0.data_ptr()), C.c_void_p(t1.data_ptr()), C.c_double(dm.params.maxc),
                                                       C.c_int64(n), C.c_int32(0), C.c_void_p(st.data_ptr()), C.c_void_p(L.data_ptr()),
                                                       C.c_void_p(ty.data_ptr()), C.c_void_p(le.data_ptr()), C.c_void_p(npts.data_ptr()), None, None))
    rs(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [rs() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({"kernel": "rs_optimal (lengths only)", "n": n, "ms": ms, "solves_per_s": n / (ms * 1e-3)}))
    out4 = dm.empty((a.n, 4), torch.float64)
    def cor():
        _native.chk(_native.lib().avp_corridor_batch(dm.h, C.c_double(0.8), C.c_void_p(t[0].data_ptr()), C.c_void_p(t[1].data_ptr()),
                                                     C.c_void_p(t[2].data_ptr()), C.c_int64(a.n), C.c_void_p(out4.data_ptr())))
    cor(); torch.cuda.synchronize()
    e0.record(); [cor() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({"kernel": "corridor", "n": a.n, "ms": ms, "waypoints_per_s": a.n / (ms * 1e-3), "algorithmic_GBps": a.n * (16 * dm.P + 56) / (ms * 1e-3) / 1e9}))


if __name__ == "__main__":
    main()
