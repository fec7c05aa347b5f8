"""Reeds-Shepp interface of the reference (`path_plan/rs_curve.py`): `PATH`, `pi_2_pi`,
`calc_optimal_path`. The solver itself runs on the GPU (`avp_rs_optimal_batch`)."""
from __future__ import annotations

import math
from typing import List

import numpy as np

STEP_SIZE = 0.5
MAX_LENGTH = 1000.0
PI = math.pi
_TYPE = ("S", "L", "R")


class PATH:
    """Same fields as the reference's PATH (`rs_curve.py:87-96`)."""

    def __init__(self, lengths, ctypes, L, x, y, yaw, directions):
        self.lengths = lengths
        self.ctypes = ctypes
        self.L = L
        self.x = x
        self.y = y
        self.yaw = yaw
        self.directions = directions


def pi_2_pi(theta):
    """Wrap to [-pi, pi] by repeated +-2pi (`rs_curve.py:649-656`); used by the downstream stages."""
    while theta > PI:
        theta -= 2.0 * PI
    while theta < -PI:
        theta += 2.0 * PI
    return theta


def path_from_arrays(types, lens, L, pts, dirs) -> PATH:
    n = int((np.asarray(types) >= 0).sum())
    return PATH(lengths=[float(v) for v in lens[:n]], ctypes=[_TYPE[int(t)] for t in types[:n]], L=float(L),
                x=[float(v) for v in pts[:, 0]], y=[float(v) for v in pts[:, 1]], yaw=[float(v) for v in pts[:, 2]],
                directions=[int(d) for d in dirs])


def calc_optimal_path(sx, sy, syaw, gx, gy, gyaw, maxc, step_size=STEP_SIZE) -> PATH:
    """The reference's scalar entry (`rs_curve.py:99-134`), same signature: the optimal Reeds-Shepp PATH from
    (sx, sy, syaw) to (gx, gy, gyaw) with curvature `maxc`, sampled every `step_size` metres. One query through
    the batched kernel (no map handle needed: avp_rs_optimal_batch accepts map = NULL)."""
    if step_size != STEP_SIZE:
        raise ValueError("the kernel samples at the reference's STEP_SIZE = 0.5 (rs_curve.py:27)")
    from . import _native
    r = _native.rs_optimal_batch([[sx, sy, syaw]], [[gx, gy, gyaw]], float(maxc), maxpts=1024)
    return _paths_from_result(r)[0]


def calc_optimal_path_batch(device_map, q0, q1, maxc=None, maxpts: int = 256) -> List[PATH]:
    return _paths_from_result(device_map.rs_optimal_batch(q0, q1, maxc=maxc, maxpts=maxpts))


def _paths_from_result(r) -> List[PATH]:
    out = []
    for i in range(len(r["L"])):
        st = int(r["status"][i])
        if st == 2:
            raise AssertionError("path.L >= 0.01")          # rs_curve.py:153
        if st == 1:
            raise IndexError("list index out of range")     # paths[0] on an empty candidate list, rs_curve.py:103
        if st == 7:
            raise ValueError("Reeds-Shepp query refused (AVP_PLAN_BAD_POSE): a coordinate or yaw that is not finite, or |yaw| > 1e6 rad (the reference's pi_2_pi loop needs more than 1.6e5 iterations there and never returns beyond ~1e16)")
        if st:
            raise RuntimeError(f"Reeds-Shepp capacity (status {st}); raise maxpts")
        k = int(r["npts"][i])
        out.append(path_from_arrays(r["types"][i], r["lens"][i], r["L"][i], r["pts"][i, :k], r["dirs"][i, :k]))
    return out
