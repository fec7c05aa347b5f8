"""Collision checkers with the reference's class API, executed by the HIP footprint kernels.

Mirror of `collision_check/collision_check.py` (reference): `collision_checker`,
`two_circle_checker`, `distance_checker`, each with `check(node_x, node_y, theta) -> bool` and
`get_near_obstacles(node_x, node_y, theta) -> ([xs, ys], vehicle_boundary)`; additive batched
entry `check_batch(poses[N,3]) -> uint8[N]` with identical per-element semantics. `check` and
`check_batch` run on the GPU through the C-ABI (`avp_check_batch`); there is no CPU path.
"""
from __future__ import annotations

from abc import abstractmethod
from typing import Tuple

import numpy as np

from . import _native
from .costmap import Map, Vehicle


class collision_checker:
    KIND = 0

    def __init__(self, map: Map, vehicle: Vehicle = None, config: dict = None) -> None:
        self.map = map
        self.config = config
        self.vehicle = vehicle
        self._dm = None

    def _device_map(self):
        if self._dm is None:
            self._dm = _native.device_map(self.map, self.vehicle, self.config)
        return self._dm

    def get_near_obstacles(self, node_x, node_y, theta) -> Tuple[list, np.ndarray]:
        """Obstacle-edge points inside the inflated footprint's AABB (inclusive), in costmap
        row-major order, and the (5,2,1) footprint (`collision_check.py:29-73`). Host-side view
        of the broad phase the kernel performs with the column bitmaps."""
        corners = self.vehicle.create_anticlockpoint(x=node_x, y=node_y, theta=theta, config=self.config)
        x_hi, x_lo = max(corners[:, 0]), min(corners[:, 0])
        y_hi, y_lo = max(corners[:, 1]), min(corners[:, 1])
        pk = self.map.pack()
        px, py = pk["obs_x"], pk["obs_y"]
        keep = (px >= x_lo) & (px <= x_hi) & (py >= y_lo) & (py <= y_hi)
        return [px[keep], py[keep]], corners

    def check_batch(self, poses, variant: int = 0) -> np.ndarray:
        return self._device_map().check_batch(poses, kind=self.KIND, variant=variant)

    def check(self, node_x, node_y, theta) -> bool:
        return bool(self.check_batch(np.array([[node_x, node_y, theta]], dtype=np.float64))[0])


class two_circle_checker(collision_checker):
    """Two discs covering the car body (`collision_check.py:76-137`)."""
    KIND = 1


class distance_checker(collision_checker):
    """Inflated rectangle vs. rasterised obstacle edges (`collision_check.py:140-240`)."""
    KIND = 0
