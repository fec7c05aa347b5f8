"""Corridor bounds of the path-smoothing stage: the collision part of the reference's `path_opti`
(`optimization/path_optimazition.py`). Only `compute_collision_H` is provided (SURVEY.md section 8f,
rank 2): the QP itself (cvxopt) and the curvature rows stay with the reference. The per-way-point scan runs
on the GPU (`avp_corridor_batch`); the matrix assembly below mirrors `:586-597`.

    po = path_opti(park_map, vehicle, config)
    po.original_path = path_i                      # list of [x, y, theta], as formate_matrix sets it (:40)
    H_collision, slack_H_collision = po.compute_collision_H()
"""
from __future__ import annotations

import numpy as np

from . import _native
from .costmap import Map, Vehicle


class path_opti:
    def __init__(self, park_map: Map, vehicle: Vehicle, config: dict) -> None:
        self.original_path = None
        self.map = park_map
        self.vehicle = vehicle
        self.matrix_dict = dict()
        self.expand_dis = config['expand_dis']
        self.config = config

    def corridor_bounds(self, path=None) -> np.ndarray:
        """(n, 4) array [x_max + x, y_max + y, x - x_min, y - y_min] per way-point."""
        path = self.original_path if path is None else path
        poses = np.array([[p[0], p[1], p[2]] for p in path], dtype=np.float64).reshape(-1, 3)
        return _native.device_map(self.map, self.vehicle, self.config).corridor_batch(poses, self.expand_dis)

    def compute_collision_H(self):
        """[E; -E] X <= [H_max; -H_min] and its slack-augmented form, as the reference returns them."""
        b = self.corridor_bounds()
        n = len(b)
        H_max = b[:, 0:2].reshape(2 * n, 1)
        H_min = b[:, 2:4].reshape(2 * n, 1)
        H_collision = np.vstack((H_max, -H_min))
        slack_H_collision = np.vstack((H_max, 999 * np.ones((n - 2, 1)), -H_min, np.zeros((n - 2, 1))))
        return H_collision, slack_H_collision


class ocp_optimization:
    """The corridor scan of the reference's OCP stage (`optimization/ocp_optimization.py:36-480`): the same
    per-way-point computation as `path_opti.compute_collision_H` (verified identical on the golden poses),
    returned as four lists. The Pyomo/IPOPT model itself is outside this package."""

    def __init__(self, park_map: Map, vehicle: Vehicle, config: dict) -> None:
        self.config = config
        self.map = park_map
        self.vehicle = vehicle
        self.expand_dis = config['expand_dis']

    def compute_collision_H(self, path):
        poses = np.array([[p[0], p[1], p[2]] for p in path], dtype=np.float64).reshape(-1, 3)
        b = _native.device_map(self.map, self.vehicle, self.config).corridor_batch(poses, self.expand_dis)
        return ([float(v) for v in b[:, 0]], [float(v) for v in b[:, 1]], [float(v) for v in b[:, 2]], [float(v) for v in b[:, 3]])
