"""ctypes binding of libavp_hip.so (C-ABI in include/avp.h) + device buffer plumbing.

PyTorch-ROCm is used for exactly two things: owning device buffers (tensors whose data_ptr() is
handed to the C-ABI) and naming the stream the kernels are queued on. There is NO CPU fallback:
if the shared library or a GPU is missing, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVP_HIP_LIB") or os.path.join(_PKG, "libavp_hip.so")    # AVP_HIP_LIB: kernel-variant experiments (scripts/variant_bench.py)
HOSTMATH_PATH = os.path.join(_PKG, "libavp_hostmath.so")
AVP_MAX_STEER = 32
AVP_MAX_SUBS = 512

EXPORTS = [
    "avp_version", "avp_sizeof_params", "avp_last_error", "avp_map_create", "avp_map_destroy", "avp_map_set_stream", "avp_sync",
    "avp_check_batch", "avp_corridor_batch", "avp_corridor_batch_v", "avp_trig_batch", "avp_libm_batch", "avp_ieee_batch", "avp_rs_optimal_batch",
    "avp_plan_workspace_bytes", "avp_plan_default_slots", "avp_sizeof_plan_result", "avp_plan_batch", "avp_plan_batch_profile", "avp_plan_batch_mode", "avp_plan_batch_ex", "avp_plan_look_bytes", "avp_plan_pick_mode", "avp_plan_slots", "avp_plan_group", "avp_plan_batch_staged", "avp_plan_set_slice_pops", "avp_plan_set_look_entries", "avp_plan_last_launch",
    "avp_hfield_id_capacity", "avp_hfield_queries", "avp_rasterize_edges", "avp_rasterize_edges_batch",
]


class AvpParams(C.Structure):
    _fields_ = [
        ("lw", C.c_double), ("lf", C.c_double), ("lr", C.c_double), ("lb", C.c_double),
        ("max_v", C.c_double), ("max_steer", C.c_double), ("min_radius", C.c_double),
        ("fp_xr", C.c_double), ("fp_xf", C.c_double), ("fp_yr", C.c_double), ("fp_yl", C.c_double),
        ("circ_rd", C.c_double), ("circ_cf", C.c_double), ("circ_cr", C.c_double),
        ("n_steer", C.c_int32), ("n_sub", C.c_int32),
        ("steer", C.c_double * AVP_MAX_STEER), ("dth_dt", C.c_double * AVP_MAX_STEER),
        ("dth_ddt1", C.c_double * AVP_MAX_STEER),
        ("travel_dt", C.c_double), ("travel_ddt1", C.c_double),
        ("flag_radius", C.c_double),
        ("cost_gear", C.c_double), ("cost_heading", C.c_double), ("cost_scale", C.c_double),
        ("maxc", C.c_double),
        ("extended_num", C.c_int32), ("checker_kind", C.c_int32), ("max_pops", C.c_int64),
    ]


def make_params(config: dict, vehicle, max_pops: int = 0) -> AvpParams:
    """Pack config + Vehicle into avp_params. Every constant is evaluated here with the reference's
    own expression order and numpy/Python functions (np.tan, np.sqrt, float **), see the file:line
    notes in include/avp.h."""
    v = vehicle
    p = AvpParams()
    p.lw, p.lf, p.lr, p.lb = v.lw, v.lf, v.lr, v.lb
    p.max_v, p.max_steer, p.min_radius = v.max_v, v.max_steering_angle, float(v.min_radius_turn)
    side, fr = config['safe_side_dis'], config['safe_fr_dis']
    p.fp_xr = -v.lr - fr                      # map/costmap.py:97
    p.fp_xf = v.lw + v.lf + fr                # :99
    p.fp_yr = -v.lb / 2 - side                # :97
    p.fp_yl = v.lb / 2 + side                 # :100
    p.circ_rd = float(0.5 * np.sqrt(((v.lr + v.lw + v.lf) / 2) ** 2 + (v.lb ** 2)))   # collision_check.py:92
    p.circ_cf = 1 / 4 * (3 * v.lw + 3 * v.lf - v.lr)                                     # :94
    p.circ_cr = 1 / 4 * (v.lw + v.lf - 3 * v.lr)                                         # :96
    n = int(config['steering_angle_num'])
    if not (1 <= n <= AVP_MAX_STEER):
        raise ValueError("steering_angle_num must be in 1..%d (2 x 32 children: one lane of a wave each)" % AVP_MAX_STEER)
    steer = np.linspace(-v.max_steering_angle, v.max_steering_angle, n)                  # hybrid_a_star.py:81-83
    dt, ddt = config['dt'], config['trajectory_dt']
    n_sub = math.ceil(dt / ddt)                                                         # :185
    if not (1 <= n_sub and 2 * n * n_sub <= AVP_MAX_SUBS):
        raise ValueError("2 * steering_angle_num * ceil(dt / trajectory_dt) must be in 1..%d (trajectory_dt too small for this steering_angle_num)" % AVP_MAX_SUBS)
    p.n_steer, p.n_sub = n, n_sub
    for i in range(n):
        p.steer[i] = float(steer[i])
        p.dth_dt[i] = float((v.max_v * np.tan(steer[i])) / v.lw * dt)                    # :146-148
        # :189-191 `(max_v * tan) / lw * ddt * (i + 1)` is evaluated left to right: everything up to the last factor here, the
        # multiplication by the exact small integer (i + 1) on the device (IEEE, -ffp-contract=off): the same double
        p.dth_ddt1[i] = float((v.max_v * np.tan(steer[i])) / v.lw * ddt)
    p.travel_dt = v.max_v * dt                                                          # :145
    p.travel_ddt1 = v.max_v * ddt                                                       # :188 `speed * ddt * (i + 1)`, likewise
    p.flag_radius = float(config['flag_radius'])
    p.cost_gear, p.cost_heading, p.cost_scale = config['cost_gear'], config['cost_heading_change'], config['cost_scale']
    p.maxc = 1 / float(v.min_radius_turn)                                               # :285
    p.extended_num = int(config['extended_num'])
    p.checker_kind = 1 if config['collision_check'] == 'circle' else 0
    p.max_pops = int(max_pops)
    return p


_lib = None


def lib():
    """The loaded C-ABI library. Raises if the extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the hot path)")
        # Load order matters: PyTorch-ROCm ships its own libamdhip64; if libavp_hip.so is loaded first it pulls
        # the system copy and the process ends up with two HIP runtimes (the second one sees no device).
        # Importing torch first lets our DT_NEEDED entry resolve to the runtime torch already loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            getattr(L, name).restype = C.c_int32
        L.avp_plan_workspace_bytes.restype = C.c_int64
        L.avp_plan_look_bytes.restype = C.c_int64
        L.avp_hfield_id_capacity.restype = C.c_int64
        if L.avp_sizeof_params() != C.sizeof(AvpParams):
            raise RuntimeError("avp_params layout mismatch between include/avp.h and _native.AvpParams")
        _lib = L
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().avp_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def chk(status: int, what: str = ""):
    if status != 0:
        raise RuntimeError(f"{what or 'libavp_hip'}: status {status}: {last_error()}")


def rasterize_edges(xs: np.ndarray, ys: np.ndarray, edges: np.ndarray, device=None, stream=None):
    """Device rasteriser (include/avp.h: avp_rasterize_edges): returns the uint8 occupancy [nx, ny] as a
    CUDA tensor and the number of samples that matched more than one node. No CPU fallback."""
    torch = torch_cuda()
    dev = torch.device(device if device is not None else "cuda")
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    ys = np.ascontiguousarray(ys, dtype=np.float64)
    edges = np.ascontiguousarray(edges, dtype=np.float64).reshape(-1, 6)
    nx, ny = len(xs), len(ys)
    d_xs, d_ys = torch.as_tensor(xs, device=dev), torch.as_tensor(ys, device=dev)
    occ = torch.zeros((nx, ny), dtype=torch.uint8, device=dev)
    multi = torch.zeros(1, dtype=torch.int32, device=dev)
    st = stream if stream is not None else torch.cuda.current_stream(dev)
    # the C-ABI takes at most 65535 edges per call (grid.y)
    for lo in range(0, len(edges), 65535):
        part = edges[lo:lo + 65535]
        d_e = torch.as_tensor(part, device=dev)
        chk(lib().avp_rasterize_edges(C.c_int32(dev.index if dev.index is not None else torch.cuda.current_device()),
                                      C.c_void_p(st.cuda_stream), C.c_void_p(d_xs.data_ptr()), C.c_void_p(d_ys.data_ptr()),
                                      C.c_int32(nx), C.c_int32(ny), C.c_double(float(xs[0])), C.c_double(float(xs[1] - xs[0])),
                                      C.c_double(float(ys[0])), C.c_double(float(ys[1] - ys[0])),
                                      C.c_void_p(d_e.data_ptr()), C.c_int64(len(part)), C.c_int32(int(part[:, 5].max()) if len(part) else 0),
                                      C.c_void_p(occ.data_ptr()), C.c_void_p(multi.data_ptr())), "avp_rasterize_edges")
        st.synchronize()        # d_e must outlive the launch
    return occ, int(multi.item())


def rasterize_edges_batch(grids, edge_tables, device=None, stream=None):
    """Batched device rasteriser (include/avp.h: avp_rasterize_edges_batch): grids = [(xs, ys)], edge_tables = [float64 [E_k, 6]], one per
    map. ONE upload of the packed tables, ONE launch, ONE read-back: returns ([uint8 occupancy [nx_k, ny_k] numpy arrays], [multi_k])."""
    torch = torch_cuda()
    dev = torch.device(device if device is not None else "cuda")
    n = len(grids)
    if n == 0:
        return [], []
    nxs = np.array([len(g[0]) for g in grids], dtype=np.int32)
    nys = np.array([len(g[1]) for g in grids], dtype=np.int32)
    node_off = np.zeros(n, np.int64)
    node_off[1:] = np.cumsum(nxs[:-1].astype(np.int64) + nys[:-1])
    nodes = np.concatenate([np.concatenate([np.asarray(g[0], dtype=np.float64), np.asarray(g[1], dtype=np.float64)]) for g in grids])
    geo = np.array([[g[0][0], g[0][1] - g[0][0], g[1][0], g[1][1] - g[1][0]] for g in grids], dtype=np.float64)
    cells = nxs.astype(np.int64) * nys
    occ_off = np.zeros(n, np.int64)
    occ_off[1:] = np.cumsum(cells[:-1])
    tabs = [np.ascontiguousarray(t, dtype=np.float64).reshape(-1, 6) for t in edge_tables]
    edges = np.concatenate(tabs) if tabs else np.zeros((0, 6))
    emap = np.concatenate([np.full(len(t), k, np.int32) for k, t in enumerate(tabs)]) if tabs else np.zeros(0, np.int32)
    st = stream if stream is not None else torch.cuda.current_stream(dev)
    # every allocation, upload, zero-fill and read-back runs on the stream the kernel is launched on (torch's current stream is
    # not necessarily `stream`: a zero-fill or a read-back queued elsewhere would race the launch)
    with torch.cuda.stream(st):
        d_nodes, d_edges, d_emap = torch.as_tensor(nodes, device=dev), torch.as_tensor(edges, device=dev), torch.as_tensor(emap, device=dev)
        occ = torch.zeros(int(cells.sum()), dtype=torch.uint8, device=dev)
        multi = torch.zeros(n, dtype=torch.int32, device=dev)
        scratch = torch.empty(64 * n, dtype=torch.uint8, device=dev)
        chk(lib().avp_rasterize_edges_batch(C.c_int32(dev.index if dev.index is not None else torch.cuda.current_device()), C.c_void_p(st.cuda_stream),
                                            C.c_int32(n), C.c_void_p(d_nodes.data_ptr()), _vp(node_off), _vp(nxs), _vp(nys), _vp(np.ascontiguousarray(geo)), _vp(occ_off),
                                            C.c_void_p(d_edges.data_ptr()), C.c_void_p(d_emap.data_ptr()), C.c_int64(len(edges)),
                                            C.c_int32(int(edges[:, 5].max()) if len(edges) else 0), C.c_void_p(occ.data_ptr()), C.c_void_p(multi.data_ptr()),
                                            C.c_void_p(scratch.data_ptr()), C.c_int64(scratch.numel())), "avp_rasterize_edges_batch")
        occ_h = occ.cpu().numpy()                          # (the one synchronisation of the batch)
        mult = multi.cpu().numpy()
    return [occ_h[occ_off[k]:occ_off[k] + cells[k]].reshape(int(nxs[k]), int(nys[k])) for k in range(n)], [int(v) for v in mult]


def torch_cuda():
    """torch with a visible GPU, or a loud failure."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no ROCm GPU visible: the hybrid-A* hot path has no CPU fallback "
                           "(the CPU restatement under oracle/ is test infrastructure only)")
    return torch


def _vp(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class DeviceMap:
    """One costmap resident in HBM on one device (avp_map handle)."""

    def __init__(self, park_map, vehicle, config: dict, device: Optional[int] = None, max_pops: int = 0):
        torch = torch_cuda()
        self.torch = torch
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.params = make_params(config, vehicle, max_pops)
        pk = park_map.pack()
        self.pack = pk
        self.P = len(pk["obs_ix"])
        h = C.c_void_p()
        bnd = np.ascontiguousarray(pk["boundary"], dtype=np.float64)
        chk(lib().avp_map_create(C.byref(self.params), _vp(pk["occ"]), C.c_int32(pk["nx"]), C.c_int32(pk["ny"]),
                                 _vp(pk["xs"]), _vp(pk["ys"]), _vp(bnd), _vp(pk["obs_ix"]), _vp(pk["obs_iy"]),
                                 C.c_int32(self.P), C.c_int32(self.device), C.byref(h)), "avp_map_create")
        self.h = h
        self.use_current_stream()

    def use_current_stream(self):
        """Queue kernels on torch's current stream of this device. Called by every launch wrapper, so that the launch
        is ordered against the tensors the caller has just produced on that stream (also under torch.cuda.stream(s))
        and torch.cuda.Event timing sees it."""
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        chk(lib().avp_map_set_stream(self.h, C.c_void_p(s)), "avp_map_set_stream")

    def planner_launch_begin(self):
        """The planner's ticket counters (and a BatchPlanner's workspace) are one per handle: a planner launch issued on
        another stream than the previous one waits for it (launches on one stream are ordered anyway)."""
        cur = self.torch.cuda.current_stream(self.device)
        ev = getattr(self, "_plan_event", None)
        if ev is not None and self._plan_stream != cur.cuda_stream:
            cur.wait_event(ev)

    def planner_launch_end(self):
        cur = self.torch.cuda.current_stream(self.device)
        ev = getattr(self, "_plan_event", None)
        if ev is None:
            ev = self._plan_event = self.torch.cuda.Event()
        ev.record(cur)
        self._plan_stream = cur.cuda_stream

    def sync(self):
        chk(lib().avp_sync(self.h), "avp_sync")

    def dev_tensor(self, arr, dtype=None):
        t = self.torch.as_tensor(np.ascontiguousarray(arr), device=f"cuda:{self.device}")
        return t if dtype is None else t.to(dtype)

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=f"cuda:{self.device}")

    def zeros(self, shape, dtype):
        return self.torch.zeros(shape, dtype=dtype, device=f"cuda:{self.device}")

    # ---- collision -----------------------------------------------------------------------------
    def check_batch_dev(self, x, y, th, out=None, kind: int = 0, variant: int = 0):
        """x, y, th: float64 CUDA tensors (SoA). Returns a uint8 CUDA tensor (asynchronous)."""
        torch = self.torch
        self.use_current_stream()
        n = x.numel()
        if out is None:
            out = torch.empty(n, dtype=torch.uint8, device=x.device)
        chk(lib().avp_check_batch(self.h, C.c_int32(kind), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                  C.c_void_p(th.data_ptr()), C.c_int64(n), C.c_void_p(out.data_ptr()), C.c_int32(variant)),
            "avp_check_batch")
        return out

    def check_batch(self, poses, kind: int = 0, variant: int = 0) -> np.ndarray:
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
        if len(poses) == 0:
            return np.zeros(0, np.uint8)
        soa = self.dev_tensor(poses.T.copy())
        out = self.check_batch_dev(soa[0], soa[1], soa[2], kind=kind, variant=variant)
        return out.cpu().numpy()

    # ---- corridor bounds ---------------------------------------------------------------------------
    def corridor_batch(self, poses, expand_dis: float, variant: int = 0) -> np.ndarray:
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
        n = len(poses)
        if n == 0:
            return np.zeros((0, 4))
        self.use_current_stream()
        soa = self.dev_tensor(poses.T.copy())
        out = self.empty((n, 4), self.torch.float64)
        chk(lib().avp_corridor_batch_v(self.h, C.c_double(float(expand_dis)), C.c_void_p(soa[0].data_ptr()), C.c_void_p(soa[1].data_ptr()),
                                       C.c_void_p(soa[2].data_ptr()), C.c_int64(n), C.c_void_p(out.data_ptr()), C.c_int32(variant)), "avp_corridor_batch_v")
        return out.cpu().numpy()

    # ---- Reeds-Shepp ---------------------------------------------------------------------------
    def rs_optimal_batch(self, q0, q1, maxc=None, maxpts: int = 128) -> dict:
        return rs_optimal_batch(q0, q1, float(self.params.maxc if maxc is None else maxc), maxpts, dm=self)

    # ---- heuristic field (diagnostic entry) ----------------------------------------------------------
    def hfield_queries(self, goal_xy, queries, force=None) -> dict:
        torch = self.torch
        self.use_current_stream()
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, 2)
        nq = len(q)
        f = np.zeros(nq, np.int32) if force is None else np.ascontiguousarray(force, dtype=np.int32)
        cap = int(lib().avp_hfield_id_capacity(self.h))
        nbytes = int(lib().avp_plan_workspace_bytes(self.h, C.c_int32(1), C.c_int32(16)))
        ws = self.empty(nbytes, torch.uint8)
        tq, tf = self.dev_tensor(q), self.dev_tensor(f)
        od, om = self.empty(max(nq, 1), torch.int32), self.empty(max(nq, 1), torch.int32)
        dist, flags, info = self.empty(cap, torch.int32), self.empty(cap, torch.uint8), self.empty(8, torch.int64)
        g = (C.c_double * 2)(float(goal_xy[0]), float(goal_xy[1]))
        chk(lib().avp_hfield_queries(self.h, g, C.c_void_p(tq.data_ptr()), C.c_void_p(tf.data_ptr()), C.c_int32(nq),
                                     C.c_void_p(ws.data_ptr()), C.c_int64(nbytes), C.c_void_p(od.data_ptr()), C.c_void_p(om.data_ptr()),
                                     C.c_void_p(dist.data_ptr()), C.c_void_p(flags.data_ptr()), C.c_void_p(info.data_ptr())),
            "avp_hfield_queries")
        return dict(d=od.cpu().numpy()[:nq], miss=om.cpu().numpy()[:nq], dist=dist.cpu().numpy().view(np.uint32),
                    flags=flags.cpu().numpy(), info=info.cpu().numpy())

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().avp_map_destroy(self.h)
                self.h = None
        except Exception:
            pass


def rs_optimal_batch(q0, q1, maxc: float, maxpts: int = 128, dm: Optional[DeviceMap] = None) -> dict:
    """avp_rs_optimal_batch. With a DeviceMap: its device and stream; without (the reference's module-level
    `rs_curve.calc_optimal_path` needs no map): torch's current device, the NULL stream."""
    torch = dm.torch if dm is not None else torch_cuda()
    dev = f"cuda:{dm.device}" if dm is not None else f"cuda:{torch.cuda.current_device()}"
    if dm is not None:
        dm.use_current_stream()
    q0 = np.ascontiguousarray(q0, dtype=np.float64).reshape(-1, 3)
    q1 = np.ascontiguousarray(q1, dtype=np.float64).reshape(-1, 3)
    n = len(q0)
    t0, t1 = torch.as_tensor(q0, device=dev), torch.as_tensor(q1, device=dev)
    st = torch.empty(n, dtype=torch.int32, device=dev)
    L = torch.empty(n, dtype=torch.float64, device=dev)
    ty = torch.empty((n, 5), dtype=torch.int8, device=dev)
    le = torch.empty((n, 5), dtype=torch.float64, device=dev)
    npts = torch.empty(n, dtype=torch.int32, device=dev)
    pts = torch.zeros((n, max(maxpts, 1), 3), dtype=torch.float64, device=dev)
    dr = torch.zeros((n, max(maxpts, 1)), dtype=torch.int8, device=dev)
    if dm is None:
        torch.cuda.current_stream().synchronize()      # the inputs were produced on torch's stream, the launch goes to the NULL stream
    chk(lib().avp_rs_optimal_batch(dm.h if dm is not None else None, C.c_void_p(t0.data_ptr()), C.c_void_p(t1.data_ptr()),
                                   C.c_double(float(maxc)), C.c_int64(n), C.c_int32(maxpts), C.c_void_p(st.data_ptr()),
                                   C.c_void_p(L.data_ptr()), C.c_void_p(ty.data_ptr()), C.c_void_p(le.data_ptr()),
                                   C.c_void_p(npts.data_ptr()), C.c_void_p(pts.data_ptr()) if maxpts > 0 else None,
                                   C.c_void_p(dr.data_ptr()) if maxpts > 0 else None), "avp_rs_optimal_batch")
    if dm is None:
        torch.cuda.synchronize()
    return dict(status=st.cpu().numpy(), L=L.cpu().numpy(), types=ty.cpu().numpy(), lens=le.cpu().numpy(),
                npts=npts.cpu().numpy(), pts=pts.cpu().numpy(), dirs=dr.cpu().numpy())


def device_map(park_map, vehicle, config, device: Optional[int] = None) -> DeviceMap:
    """DeviceMap cached on the Map object per (device, config identity)."""
    torch = torch_cuda()
    dev = torch.cuda.current_device() if device is None else int(device)
    cache = park_map.__dict__.setdefault("_avp_device_maps", {})
    key = (dev, id(config), tuple(sorted((k, repr(v)) for k, v in config.items())))
    dm = cache.get(key)
    if dm is None:
        dm = DeviceMap(park_map, vehicle, config, dev)
        cache[key] = dm
    return dm
