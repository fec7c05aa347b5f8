"""World model: vehicle constants, TPCAP scenario reader and the obstacle-edge costmap.

Host-side mirror of the reference's `map/costmap.py` interface (same class / attribute / method
names, same numeric results) so that `PathPlanner(config, map, vehicle)` accepts either. Built
once per map on the host with numpy; the hot path consumes the packed arrays (`Map.pack()`).

Reference semantics followed (file:line under the reference repo):
  Vehicle            map/costmap.py:51-121
  Case.read          map/costmap.py:134-156   (bounds = start/goal +- 12 m)
  Map.__init__       map/costmap.py:160-176   (boundary = floor(...))
  discrete_map       map/costmap.py:178-195   (pitch = linspace step, NOT discrete_size)
  detect_obstacle_edge  map/costmap.py:197-261 (edges only, centroid-angle vertex sort)
  convert_position_to_index  map/costmap.py:319-329 (row stride int((b1-b0)/dx) != nx)
"""
from __future__ import annotations

import csv
import math
from typing import List, Optional, Sequence

import numpy as np


class Vehicle:
    """Vehicle geometry and limits (`map/costmap.py:52-63`)."""

    def __init__(self):
        self.lw = 2.8      # wheelbase
        self.lf = 0.96     # front overhang
        self.lr = 0.929    # rear overhang
        self.lb = 1.942    # width
        self.max_steering_angle = 0.75
        self.max_angular_velocity = 0.5
        self.max_acc = 1
        self.max_v = 2.5
        self.min_v = -2.5
        self.min_radius_turn = self.lw / np.tan(self.max_steering_angle) + self.lb / 2

    def create_polygon(self, x, y, theta):
        """Un-inflated outline rr, rf, lf, lr, rr as (5, 2) (`map/costmap.py:65-83`)."""
        c, s = np.cos(theta), np.sin(theta)
        local = np.array([[-self.lr, -self.lb / 2, 1], [self.lf + self.lw, -self.lb / 2, 1],
                          [self.lf + self.lw, self.lb / 2, 1], [-self.lr, self.lb / 2, 1],
                          [-self.lr, -self.lb / 2, 1]])
        tf = np.array([[c, -s, x], [s, c, y], [0, 0, 1]])
        return local.dot(tf.transpose())[:, 0:2]

    def create_anticlockpoint(self, x, y, theta, config: dict = None):
        """Inflated footprint corners rr, rf, lf, lr, rr with shape (5, 2, 1) (`map/costmap.py:85-121`)."""
        side = config['safe_side_dis']
        fr = config['safe_fr_dis']
        # world = R(theta) . local + origin, with R taken as the transposed VIEW of R(-theta) so that
        # the BLAS call (and therefore the rounding order) is the one the reference executes
        rot = np.array([[np.cos(theta), np.sin(theta)], [-np.sin(theta), np.cos(theta)]]).transpose()
        origin = np.array([[x], [y]])
        local = (np.array([[-self.lr - fr], [-self.lb / 2 - side]]),
                 np.array([[self.lw + self.lf + fr], [-self.lb / 2 - side]]),
                 np.array([[self.lw + self.lf + fr], [self.lb / 2 + side]]),
                 np.array([[-self.lr - fr], [self.lb / 2 + side]]))
        world = [rot.dot(p) + origin for p in local]
        world.append(world[0])
        return np.array([[w[0], w[1]] for w in world])


class Case:
    """One TPCAP scenario (`map/costmap.py:124-156`)."""

    def __init__(self):
        self.x0, self.y0, self.theta0 = 0, 0, 0
        self.xf, self.yf, self.thetaf = 0, 0, 0
        self.xmin, self.xmax = 0, 0
        self.ymin, self.ymax = 0, 0
        self.obs_num = 0
        self.obs: List[np.ndarray] = []
        self.vehicle = Vehicle()

    @staticmethod
    def from_values(v: Sequence[float]) -> "Case":
        case = Case()
        case.x0, case.y0, case.theta0 = v[0:3]
        case.xf, case.yf, case.thetaf = v[3:6]
        case.xmin = min(case.x0, case.xf) - 12
        case.xmax = max(case.x0, case.xf) + 12
        case.ymin = min(case.y0, case.yf) - 12
        case.ymax = max(case.y0, case.yf) + 12
        case.obs_num = int(v[6])
        counts = [int(c) for c in v[7:7 + case.obs_num]]
        pos = 7 + case.obs_num
        case.obs = []
        for nv in counts:
            case.obs.append(np.array(v[pos:pos + 2 * nv], dtype=np.float64).reshape((nv, 2)))
            pos += 2 * nv
        return case

    @staticmethod
    def read(file) -> "Case":
        with open(file, 'r') as f:
            row = next(csv.reader(f))
        return Case.from_values([float(t) for t in row])


class Map:
    """Obstacle-edge costmap over floor(bounds) with pitch = linspace step."""

    def __init__(self, discrete_size: float = 0.1, file: Optional[str] = None, case: Optional[Case] = None,
                 device=None):
        """device=None: host rasteriser (numpy, the reference's own arithmetic); device="cuda[:k]": the
        per-sample stage runs in libavp_hip.so (avp_rasterize_edges), bit-identical cells, no CPU fallback."""
        self.discrete_size = discrete_size
        self.grid_index = None
        self.cost_map = np.array([], dtype=np.float64)
        self.map_position = np.array([], dtype=np.float64)
        self.case = case if case is not None else Case.read(file)
        self.boundary = np.array([math.floor(self.case.xmin), math.floor(self.case.xmax),
                                  math.floor(self.case.ymin), math.floor(self.case.ymax)], dtype=np.float64)
        self._discrete_x = 0
        self._discrete_y = 0
        self._packed = None
        if device is None:
            self.detect_obstacle_edge()
        else:
            self.detect_obstacle_edge_device(device)

    # -- grid -------------------------------------------------------------------------------
    def discrete_map(self):
        nx = int((self.boundary[1] - self.boundary[0]) / self.discrete_size)
        ny = int((self.boundary[3] - self.boundary[2]) / self.discrete_size)
        self.cost_map = np.zeros((nx, ny), dtype=np.float64)
        xs = np.linspace(self.boundary[0], self.boundary[1], nx)
        ys = np.linspace(self.boundary[2], self.boundary[3], ny)
        self._discrete_x = xs[1] - xs[0]
        self._discrete_y = ys[1] - ys[0]
        self.map_position = (xs, ys)
        self.grid_index_max = nx * ny

    @staticmethod
    def _node_below(axis: np.ndarray, p: float, pitch: float) -> int:
        """Index i of the unique node with axis[i] < p and axis[i] > p - pitch, or -1 if none."""
        hit = np.where((axis < p) & (axis > (p - pitch)))[0]
        if len(hit) == 0:
            return -1
        if len(hit) > 1:
            raise TypeError("only length-1 arrays can be converted to Python scalars")
        return int(hit[0])

    def detect_obstacle_edge(self):
        """Host rasteriser: edge table, then per edge the samples of map/costmap.py:236-261 in numpy."""
        self.discrete_map()
        xs, ys = self.map_position
        dx, dy = self._discrete_x, self._discrete_y
        for p1x, p1y, ca, sa, length, count in self.edge_table():
            count = int(count)
            rot = np.array([[ca, sa], [-sa, ca]])
            along = np.vstack((np.linspace(0, length, count), np.zeros(count)))
            pts = np.dot(rot.transpose(), along)
            for q in range(count):
                i = self._node_below(xs, pts[0][q] + p1x, dx)
                jj = self._node_below(ys, pts[1][q] + p1y, dy)
                if i >= 0 and jj >= 0:
                    self.cost_map[i][jj] = 255
        self._packed = None

    def edge_table(self) -> np.ndarray:
        """Host half of detect_obstacle_edge (map/costmap.py:203-236): np.unique, centroid-angle sort and, per
        polygon edge, [p1x, p1y, cos, sin, rotated length, count = floor(length / dx)] -> float64 [E, 6].
        Requires discrete_map() to have run."""
        dx = self._discrete_x
        rows = []
        for k in range(self.case.obs_num):
            poly = np.unique(self.case.obs[k], axis=0)
            nv = len(poly[:, 0])
            cx, cy = np.mean(poly[:, 0]), np.mean(poly[:, 1])
            ang = np.arctan2(poly[:, 1] - cy, poly[:, 0] - cx) + np.pi
            poly = poly[np.argsort(ang)]
            for j in range(nv):
                p1 = [poly[j, 0], poly[j, 1]]
                p2 = [poly[0, 0], poly[0, 1]] if j + 1 == nv else [poly[j + 1, 0], poly[j + 1, 1]]
                edge = [p2[0] - p1[0], p2[1] - p1[1]]
                a = np.arctan2(edge[1], edge[0])
                rot = np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
                length = np.dot(rot, np.array(edge).reshape([2, 1]))[0].tolist()[0]
                rows.append([p1[0], p1[1], rot[0, 0], rot[0, 1], length, float(math.floor(length / dx))])
        return np.array(rows, dtype=np.float64).reshape(-1, 6)

    def detect_obstacle_edge_device(self, device="cuda"):
        """detect_obstacle_edge with the per-sample stage (map/costmap.py:236-261) on the GPU."""
        from . import _native
        self.discrete_map()
        xs, ys = self.map_position
        occ, multi = _native.rasterize_edges(xs, ys, self.edge_table(), device=device)
        if multi:
            raise TypeError("only length-1 arrays can be converted to Python scalars")      # map/costmap.py:260
        self.cost_map = occ.cpu().numpy().astype(np.float64)
        self._packed = None

    @staticmethod
    def load_batch(files, discrete_size: float = 0.1, device="cuda", cases=None):
        """Batched TPCAP ingest (map/costmap.py:134-156 + 197-261 for a LIST of scenario files): every file parsed and its edge table
        built on the host (numpy: the reference's own arithmetic), then ALL maps rasterised by one launch of the device rasteriser
        with one upload and one read-back (include/avp.h: avp_rasterize_edges_batch) -- instead of one launch and one
        synchronisation per map. The maps are those of `Map(file=f, discrete_size=..., device=device)`, cell for cell."""
        from . import _native
        maps = []
        for k, f in enumerate(files):
            m = Map.__new__(Map)
            m.discrete_size = discrete_size
            m.grid_index = None
            m.cost_map = np.array([], dtype=np.float64)
            m.map_position = np.array([], dtype=np.float64)
            m.case = cases[k] if cases is not None else Case.read(f)
            m.boundary = np.array([math.floor(m.case.xmin), math.floor(m.case.xmax), math.floor(m.case.ymin), math.floor(m.case.ymax)], dtype=np.float64)
            m._discrete_x = 0
            m._discrete_y = 0
            m._packed = None
            m.discrete_map()
            maps.append(m)
        occs, multis = _native.rasterize_edges_batch([m.map_position for m in maps], [m.edge_table() for m in maps], device=device)
        for m, occ, multi in zip(maps, occs, multis):
            if multi:
                raise TypeError("only length-1 arrays can be converted to Python scalars")      # map/costmap.py:260
            m.cost_map = occ.astype(np.float64)
        return maps

    def convert_position_to_index(self, grid_x, grid_y):
        col = math.floor((grid_x - self.boundary[0]) / self._discrete_x)
        row = math.floor((self.boundary[3] - grid_y) / self._discrete_y)
        return col + row * int((self.boundary[1] - self.boundary[0]) / self._discrete_x)

    # -- packed view consumed by the C-ABI ----------------------------------------------------
    def pack(self) -> dict:
        """Arrays handed to `avp_map_create`: uint8 occupancy [ix*ny+iy], node coordinates,
        obstacle points in np.where (row-major) order, pitch, strides. Cached until the costmap
        is rebuilt."""
        if self._packed is None:
            occ = (self.cost_map == 255)
            ix, iy = np.where(occ)
            xs, ys = self.map_position
            b = self.boundary
            self._packed = dict(
                nx=int(self.cost_map.shape[0]), ny=int(self.cost_map.shape[1]),
                occ=np.ascontiguousarray(np.where(occ, 255, 0).astype(np.uint8)),
                xs=np.ascontiguousarray(xs, dtype=np.float64), ys=np.ascontiguousarray(ys, dtype=np.float64),
                boundary=np.ascontiguousarray(b, dtype=np.float64),
                dx=float(self._discrete_x), dy=float(self._discrete_y),
                S=int((b[1] - b[0]) / self._discrete_x), Sy=int((b[3] - b[2]) / self._discrete_y),
                obs_x=np.ascontiguousarray(xs[ix], dtype=np.float64),
                obs_y=np.ascontiguousarray(ys[iy], dtype=np.float64),
                obs_ix=ix.astype(np.int32), obs_iy=iy.astype(np.int32))
        return self._packed

    @staticmethod
    def from_cells(case: Case, boundary, nx: int, ny: int, cells: np.ndarray) -> "Map":
        """Rebuild a Map from stored occupancy cells (used with golden fixtures so that a test does
        not depend on re-rasterising with this host's numpy SIMD dispatch)."""
        m = Map.__new__(Map)
        m.discrete_size = None
        m.grid_index = None
        m.case = case
        m.boundary = np.asarray(boundary, dtype=np.float64)
        xs = np.linspace(m.boundary[0], m.boundary[1], nx)
        ys = np.linspace(m.boundary[2], m.boundary[3], ny)
        m._discrete_x = xs[1] - xs[0]
        m._discrete_y = ys[1] - ys[0]
        m.map_position = (xs, ys)
        m.cost_map = np.zeros((nx, ny), dtype=np.float64)
        if len(cells):
            m.cost_map[cells[:, 0], cells[:, 1]] = 255
        m.grid_index_max = nx * ny
        m._packed = None
        return m
