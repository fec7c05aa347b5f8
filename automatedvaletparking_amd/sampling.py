"""Synthetic start/goal pose samplers and synthetic-map writers for the benchmark configs.

SURVEY.md §8(d) configs C2-C5: poses are drawn uniformly inside the map bounds (6 m margin) and
rejected when the vehicle footprint collides with the rasterised obstacle edges or when a footprint
corner / the reference point lies inside an obstacle polygon (obstacles are hollow in the costmap,
reference `map/costmap.py:197-261`).

Pure host code (numpy); no reference code is involved.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def point_in_polygon(px: float, py: float, poly: np.ndarray) -> bool:
    """Crossing-number test of (px, py) against polygon vertices `poly` (nv, 2), any orientation."""
    inside = False
    n = len(poly)
    j = n - 1
    for i in range(n):
        xi, yi = poly[i]
        xj, yj = poly[j]
        if (yi > py) != (yj > py):
            xint = (xj - xi) * (py - yi) / (yj - yi) + xi
            if px < xint:
                inside = not inside
        j = i
    return inside


def footprint_points(x: float, y: float, th: float, lw=2.8, lf=0.96, lr=0.929, lb=1.942,
                     fr=0.1, side=0.1) -> List[Tuple[float, float]]:
    """Reference point + 4 inflated corners (same geometry as `map/costmap.py:85-121`)."""
    c, s = math.cos(th), math.sin(th)
    loc = [(0.0, 0.0), (-lr - fr, -lb / 2 - side), (lw + lf + fr, -lb / 2 - side),
           (lw + lf + fr, lb / 2 + side), (-lr - fr, lb / 2 + side), ((lw + lf - lr) / 2, 0.0)]
    return [(x + c * lx - s * ly, y + s * lx + c * ly) for lx, ly in loc]


def pose_is_free(x, y, th, obstacles: Sequence[np.ndarray], check: Optional[Callable] = None) -> bool:
    if check is not None and bool(check(x, y, th)):
        return False
    for (qx, qy) in footprint_points(x, y, th):
        for poly in obstacles:
            if point_in_polygon(qx, qy, poly):
                return False
    return True


def sample_free_poses(boundary, obstacles, n: int, rng: np.random.Generator, margin: float = 6.0,
                      check: Optional[Callable] = None, reject: bool = True) -> np.ndarray:
    """n poses (x, y, theta) with x~U[b0+m, b1-m], y~U[b2+m, b3-m], theta~U[-pi, pi)."""
    b0, b1, b2, b3 = [float(v) for v in boundary]
    out = np.empty((n, 3), dtype=np.float64)
    k = 0
    guard = 0
    while k < n:
        x = rng.uniform(b0 + margin, b1 - margin)
        y = rng.uniform(b2 + margin, b3 - margin)
        th = rng.uniform(-math.pi, math.pi)
        guard += 1
        if guard > 1000 * (n + 10):
            raise RuntimeError("sampler could not find free poses")
        if reject and not pose_is_free(x, y, th, obstacles, check):
            continue
        out[k] = (x, y, th)
        k += 1
    return out


def write_tpcap_csv(path: str, start, goal, obstacles: Sequence[np.ndarray]) -> None:
    """Write one scenario in the TPCAP one-row CSV layout read by `Case.read` (`map/costmap.py:134-156`):
    x0,y0,th0,xf,yf,thf,n_obs,nv_1..nv_n,then the vertices of every obstacle as x,y pairs."""
    vals: List[float] = [*start, *goal, float(len(obstacles))]
    vals += [float(len(o)) for o in obstacles]
    for o in obstacles:
        for vx, vy in np.asarray(o, dtype=np.float64):
            vals += [float(vx), float(vy)]
    with open(path, "w") as f:
        f.write(",".join(repr(float(v)) for v in vals))


def synthetic_polygon_map(seed: int = 4, size: float = 24.0, n_obs: int = 32) -> List[np.ndarray]:
    """Config C4: regular n-gons, n~U{3..8}, circum-radius~U[0.6,1.6] m, centres~U[3, size-3]^2 with
    2.5 m minimum centre separation beyond the radii."""
    rng = np.random.default_rng(seed)
    cent: List[Tuple[float, float, float]] = []
    polys: List[np.ndarray] = []
    tries = 0
    while len(polys) < n_obs and tries < 100000:
        tries += 1
        n = int(rng.integers(3, 9))
        r = float(rng.uniform(0.6, 1.6))
        cx, cy = rng.uniform(3.0, size - 3.0, size=2)
        if any(math.hypot(cx - ox, cy - oy) < 2.5 + 0.0 * (r + orr) for ox, oy, orr in cent):
            continue
        ph = float(rng.uniform(0, 2 * math.pi))
        ang = ph + 2 * math.pi * np.arange(n) / n
        polys.append(np.stack([cx + r * np.cos(ang), cy + r * np.sin(ang)], axis=1))
        cent.append((cx, cy, r))
    return polys


def parking_lot_map(n_per_row: int = 60, pitch: float = 2.5, car_l: float = 4.7, car_w: float = 1.9,
                    aisle: float = 5.5, empty_bay: int = 30):
    """Config C5: two rows of parked rectangles (perpendicular bays) either side of an aisle, one
    empty bay in the lower row = goal. Returns (obstacles, goal_pose, aisle_box)."""
    obs: List[np.ndarray] = []
    y_low0, y_low1 = 0.0, car_l
    y_up0, y_up1 = car_l + aisle, 2 * car_l + aisle
    goal = None
    for row, (ya, yb) in enumerate(((y_low0, y_low1), (y_up0, y_up1))):
        for k in range(n_per_row):
            xc = (k + 0.5) * pitch
            if row == 0 and k == empty_bay:
                # nose-in bay: rear axle near the bay mouth, heading -y
                goal = (xc, ya + car_l - 0.96 - 0.3, -math.pi / 2)
                continue
            x0, x1 = xc - car_w / 2, xc + car_w / 2
            obs.append(np.array([[x0, ya], [x1, ya], [x1, yb], [x0, yb]], dtype=np.float64))
    aisle_box = (0.0, n_per_row * pitch, car_l, car_l + aisle)
    return obs, goal, aisle_box
