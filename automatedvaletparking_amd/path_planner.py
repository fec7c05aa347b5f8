"""`PathPlanner` with the reference's interface (`path_plan/path_planner.py`), executed by the
batched HIP planner.

    planner = PathPlanner(config=config, map=park_map, vehicle=ego_vehicle)     # main.py:37-39
    original_path, path_info, split_path = planner.path_planning()              # main.py:66

plus the additive batched surface `plan_batch(starts[N,3], goals[N,3]) -> list[PlanResult]`.
`a_star_plan`/`path_planning` raise what the reference raises (AttributeError when no path exists,
IndexError when the path has no gear change); conditions under which the reference hangs or has no
error of its own are reported as RuntimeError with the per-problem status.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _native
from . import collision_check
from .costmap import Map, Vehicle
from .rs_curve import PATH, path_from_arrays

STATUS_NAMES = {0: "OK", 1: "NO_PATH", 2: "H_UNREACHABLE", 3: "RS_ERROR", 4: "ITER_LIMIT", 5: "CAPACITY", 6: "LATTICE", 7: "BAD_POSE",
                100: "DEFERRED", -1: "UNFINISHED"}       # (DEFERRED: only from the first stage of a staged call run on its own, BatchPlanner.plan_dev(first_stage_only=True))
STAGED = 16                           # BatchPlanner mode: avp_plan_batch_staged


class AvpPlanResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_pops", C.c_int32), ("n_astar", C.c_int32), ("n_rs_pts", C.c_int32),
                ("n_final", C.c_int32), ("rs_n", C.c_int32), ("in_radius_last", C.c_int32), ("rs_collision", C.c_int32),
                ("n_checks", C.c_int64), ("n_rs", C.c_int64), ("n_closed", C.c_int64), ("n_open", C.c_int64),
                ("h_cells", C.c_int64), ("h_misses", C.c_int64), ("global_index", C.c_int64), ("n_nodes", C.c_int64),
                ("rs_types", C.c_int8 * 8), ("rs_lengths", C.c_double * 5), ("rs_L", C.c_double),
                ("rs_start", C.c_double * 3), ("rs_dir0", C.c_int32), ("slot", C.c_int32), ("phase_cycles", C.c_int64 * 64)]


RESULT_DTYPE = np.dtype([("status", "<i4"), ("n_pops", "<i4"), ("n_astar", "<i4"), ("n_rs_pts", "<i4"), ("n_final", "<i4"),
                         ("rs_n", "<i4"), ("in_radius_last", "<i4"), ("rs_collision", "<i4"),
                         ("n_checks", "<i8"), ("n_rs", "<i8"), ("n_closed", "<i8"), ("n_open", "<i8"), ("h_cells", "<i8"),
                         ("h_misses", "<i8"), ("global_index", "<i8"), ("n_nodes", "<i8"),
                         ("rs_types", "i1", (8,)), ("rs_lengths", "<f8", (5,)), ("rs_L", "<f8"),
                         ("rs_start", "<f8", (3,)), ("rs_dir0", "<i4"), ("slot", "<i4"), ("phase_cycles", "<i8", (64,))])
assert RESULT_DTYPE.itemsize == C.sizeof(AvpPlanResult)


@dataclass
class PlanResult:
    status: int
    n_pops: int
    final_path: np.ndarray            # (n_final, 3)  astar way-points + RS samples[1:]
    astar_path: np.ndarray            # (n_astar, 3)
    rs_types: List[str]
    rs_lengths: List[float]
    rs_L: float
    n_rs_pts: int
    rs_xyyaw: np.ndarray = None       # (n_rs_pts, 3) samples of the final RS shot (sample 0 = last popped node)
    rs_dirs: np.ndarray = None        # (n_rs_pts,) +1 forward / -1 reverse
    counters: Dict[str, int] = field(default_factory=dict)
    trace: Optional[np.ndarray] = None
    # filled by plan_batch(..., split=True): what path_planning() adds on top of a_star_plan() (path_planner.py:45-56)
    segments: Optional[List[List[List[float]]]] = None   # split_path_list: gear segments incl. the extension points
    change_gear: Optional[int] = None
    split_error: Optional[str] = None                    # "IndexError": the path has no gear change (path_planner.py:181)

    @property
    def ok(self) -> bool:
        return self.status == 0

    @property
    def status_name(self) -> str:
        return STATUS_NAMES.get(self.status, str(self.status))


def long_search_mode(dm, n: int) -> int:
    """Kernel form for n searches that are all known to be LONG (the second stage of the two-stage deal,
    automatedvaletparking_amd.distributed): the form with the shortest pop whose problem slots hold them all at once --
    one workgroup per search up to the number of compute units, then four waves, a pair of waves, one wave per search."""
    L = _native.lib()
    for mode in (1, 4, 3):
        if n <= int(L.avp_plan_slots(dm.h, C.c_int32(mode))):
            return mode
    return 2


class BatchPlanner:
    """Device-side batched planner bound to one DeviceMap. Owns the scratch workspace (torch tensor)."""

    def __init__(self, device_map: _native.DeviceMap, max_nodes: int = 65536, n_slots: Optional[int] = None,
                 max_path: int = 512, mode: int = 0, lookahead: Optional[bool] = None,
                 longest_first: bool = False, stage_pops: int = 16, time_slice: Optional[bool] = None,
                 slice_pops: Optional[int] = None, look_entries_log2: int = 0):
        """mode: 0 = the library's choice by batch size, 1 = one workgroup per problem, 2 = one wave, 3 = a pair of waves,
        4 = four waves per problem (include/avp.h: avp_plan_batch_mode), STAGED = avp_plan_batch_staged: every problem in
        the wave form for stage_pops pops, the searches still running then planned again in the form that suits their
        number (same results, the shortest time for batches of a few to a few dozen problems per CU).
        n_slots: problem slots (default: what the chosen form can keep busy).
        lookahead: let the compute units without a problem of their own pre-compute node expansions for the running
        searches (avp_plan_batch_ex; results are identical either way). None = when the library wants it for the batch
        and its record store fits LOOK_BYTES_MAX, True = whenever the library supports it, False = never.
        time_slice: group forms (modes 2 - 4) with more problems than groups: give every problem its own workspace slot, so
        that the kernel parks a search that is still running after slice_pops pops (default: the library's, 64) while
        others wait and the long searches advance side by side (include/avp.h: avp_plan_set_slice_pops; same results).
        None = from SLICE_MIN_RATIO problems per group on, when the slots fit SLICE_FREE_FRAC of the free device memory;
        True = whenever there are more problems than groups; False = never.
        longest_first: start a batch with more problems than slots by decreasing start-goal distance (the `order` argument of
        avp_plan_batch_ex; results keep the caller's order). Off by default: on the bench's random pairs the distance does not
        predict the length of the search (4 096 problems: 119 vs 116 ms in index order, scripts/order_bench.py)."""
        self.dm = device_map
        self.max_nodes = int(max_nodes)
        L = _native.lib()
        self.mode = int(mode)
        self.n_slots = int(n_slots) if n_slots else 0
        self.max_path = int(max_path)
        if L.avp_sizeof_plan_result() != C.sizeof(AvpPlanResult):
            raise RuntimeError("avp_plan_result layout mismatch")
        self._ws = None
        self._ws_slots = 0
        self.lookahead = lookahead
        self.longest_first = bool(longest_first)
        self.stage_pops = int(stage_pops)
        self.time_slice = time_slice
        self.slice_pops = slice_pops
        self.last_time_sliced = False
        self._look = None
        self.last_lookahead = False
        self.look_entries_log2 = int(look_entries_log2)      # diagnostics: 2^this records in the lookahead's store (0 = the library's 2^18)

    LOOK_BYTES_MAX = 1 << 30           # (the record store has a fixed size since round 6 -- 262 144 records, 0.2 GB -- whatever the batch and the node arena)
    LOOK_FREE_FRAC = 0.25              # ... and at most this share of the device memory that is free right now
    SLICE_FREE_FRAC = 0.5              # time slicing: one workspace slot per problem, at most this share of the free memory
    SLICE_MIN_RATIO = 6                # ... and by default only from this many problems per group on (measured on the Case1 sets:
                                       # 4 per group: 0 to -4 %; 8 per group: +14 % wave form, +19 % pair form)

    def _slice_slots(self, n, wg, mode):
        """Slot count for a time-sliced launch of n problems in group form `mode` (a slot per problem), or 0."""
        if self.time_slice is False or self.n_slots or mode < 2:
            return 0
        L = _native.lib()
        groups = int(L.avp_plan_slots(self.dm.h, C.c_int32(mode)))
        if n <= groups or (self.time_slice is None and n < self.SLICE_MIN_RATIO * groups):
            return 0                                        # every problem has a group of its own from the start / too few to pay
        want = wg * ((n + wg - 1) // wg)
        if self.time_slice is None and not (self._ws is not None and self._ws_slots >= want):
            nbytes = int(L.avp_plan_workspace_bytes(self.dm.h, C.c_int32(want), C.c_int32(self.max_nodes)))
            free, _ = self.dm.torch.cuda.mem_get_info(self.dm.device)
            if nbytes <= 0 or nbytes > self.SLICE_FREE_FRAC * free:
                return 0
        return want

    def _look_workspace(self, n):
        """The lookahead's record store, or None. lookahead=None (the default) is conservative: never for a single problem
        (a lone a_star_plan() leaves the other compute units to whoever else uses the device: the helpers would park a
        spinning workgroup on each), never when the store (a fixed 0.2 GB: avp_plan_look_bytes) would take more than LOOK_FREE_FRAC
        of the free device memory, and an allocation failure falls back to planning without it. lookahead=True asks for it
        whenever the library supports it for the batch (and raises if the store cannot be allocated)."""
        if self.lookahead is False:
            return None
        _native.chk(_native.lib().avp_plan_set_look_entries(self.dm.h, C.c_int32(self.look_entries_log2)), "avp_plan_set_look_entries")      # (state of the handle: set for every launch)
        nbytes = int(_native.lib().avp_plan_look_bytes(self.dm.h, C.c_int64(n), C.c_int32(self.max_nodes)))
        if nbytes <= 0:
            return None
        have = self._look is not None and self._look.numel() >= nbytes
        if self.lookahead is None:
            if n < 2 or nbytes > self.LOOK_BYTES_MAX:
                return None
            if not have:
                free, _ = self.dm.torch.cuda.mem_get_info(self.dm.device)
                if nbytes > self.LOOK_FREE_FRAC * free:
                    return None
        if not have:
            self._look = None
            try:
                self._look = self.dm.empty(nbytes, self.dm.torch.uint8)
            except self.dm.torch.cuda.OutOfMemoryError:
                if self.lookahead is True:
                    raise
                return None
        return self._look

    def _workspace(self, slots):
        if self._ws is None or self._ws_slots < slots:
            L = _native.lib()
            nbytes = int(L.avp_plan_workspace_bytes(self.dm.h, C.c_int32(slots), C.c_int32(self.max_nodes)))
            if nbytes <= 0:
                raise RuntimeError(_native.last_error())
            self._ws, self._ws_slots = None, 0              # (release the old buffer first: the two need not coexist)
            self._ws = self.dm.empty(nbytes, self.dm.torch.uint8)
            self._ws_slots = slots
        return self._ws

    def plan_dev(self, starts_t, goals_t, want_paths=True, max_trace: int = 0, profile: bool = False, first_stage_only: bool = False):
        """starts_t/goals_t: (n,3) float64 CUDA tensors. Asynchronous; returns device tensors
        (results as uint8 (n, sizeof result), paths (n, max_path, 3) or None, trace or None).
        profile=True runs the instrumented kernel (phase_cycles filled). first_stage_only (STAGED mode): stop after the
        first stage; unfinished searches carry status 100 (DEFERRED)."""
        torch = self.dm.torch
        self.dm.use_current_stream()
        self.dm.planner_launch_begin()
        try:
            return self._plan_dev(starts_t, goals_t, want_paths, max_trace, profile, first_stage_only)
        finally:
            self.dm.planner_launch_end()                    # also when a launch raised: the next one is ordered after what was enqueued

    def _plan_dev(self, starts_t, goals_t, want_paths, max_trace, profile, first_stage_only):
        torch = self.dm.torch
        n = starts_t.shape[0]
        L = _native.lib()
        if self.mode == STAGED and not profile:
            grp = int(L.avp_plan_group(C.c_int32(2)))
            cap = self.n_slots if self.n_slots else int(L.avp_plan_slots(self.dm.h, C.c_int32(2)))
            slots = max(grp, min(cap, grp * ((n + grp - 1) // grp)))
            ws = self._workspace(slots)
            res = self.dm.empty((max(n, 1), C.sizeof(AvpPlanResult)), torch.uint8)
            paths = self.dm.empty((max(n, 1), self.max_path, 4), torch.float64) if want_paths else None
            trace = self.dm.zeros((max(n, 1), max_trace, 11), torch.float64) if max_trace > 0 else None
            self.last_lookahead = False
            self.last_time_sliced = False
            # (the staged stages never park a search; its non-staged fall-back inside the library -- more children than the
            #  group forms hold -- must not inherit the slice length the handle's previous user left)
            _native.chk(L.avp_plan_set_slice_pops(self.dm.h, C.c_int32(0)), "avp_plan_set_slice_pops")
            _native.chk(L.avp_plan_batch_staged(self.dm.h, C.c_void_p(starts_t.data_ptr()), C.c_void_p(goals_t.data_ptr()), C.c_int64(n), C.c_int32(slots),
                                                C.c_int32(self.max_nodes), C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()), C.c_void_p(res.data_ptr()),
                                                C.c_void_p(paths.data_ptr()) if paths is not None else None, C.c_int32(self.max_path),
                                                C.c_void_p(trace.data_ptr()) if trace is not None else None, C.c_int32(max_trace),
                                                C.c_int32(self.stage_pops), C.c_int32(1 if first_stage_only else 0), None), "avp_plan_batch_staged")
            return res, paths, trace
        mode = (self.mode if self.mode in (2, 3, 4) else 1) if profile else int(L.avp_plan_pick_mode(self.dm.h, C.c_int64(n), C.c_int32(self.mode if self.mode != STAGED else 0)))
        cap = self.n_slots if self.n_slots else int(L.avp_plan_slots(self.dm.h, C.c_int32(mode)))
        wg = int(L.avp_plan_group(C.c_int32(mode)))         # problems per workgroup of the kernel form
        slots = max(1, min(cap, wg * ((n + wg - 1) // wg)))
        if mode >= 2 and slots < wg:
            mode = 1                                        # fewer than one workgroup of slots: the workgroup form
        sliced = 0 if profile else self._slice_slots(n, wg, mode)
        try:
            ws = self._workspace(sliced if sliced else slots)
            if sliced:
                slots = sliced
        except torch.cuda.OutOfMemoryError:
            if not sliced or self.time_slice is True:
                raise
            sliced = 0                                      # default time slicing is opportunistic: plan unsliced instead
            ws = self._workspace(slots)
        # The library parks searches whenever the workspace has a slot per problem, there are more problems than groups and
        # the handle's slice length is non-zero (avp_plan_batch_mode). The slice length is state of the handle, so it is set
        # for EVERY group-form launch -- 0 when this planner does not want slicing -- and last_time_sliced is the library's
        # own condition, not a guess (a caller's explicit n_slots >= n used to be sliced silently with whatever length the
        # handle's previous user left behind).
        pops = 0
        if mode >= 2 and not profile:
            if sliced or (self.n_slots and self.time_slice is True):
                pops = -1 if self.slice_pops is None else int(self.slice_pops)
            _native.chk(L.avp_plan_set_slice_pops(self.dm.h, C.c_int32(pops)), "avp_plan_set_slice_pops")
        res = self.dm.empty((max(n, 1), C.sizeof(AvpPlanResult)), torch.uint8)
        paths = self.dm.empty((max(n, 1), self.max_path, 4), torch.float64) if want_paths else None
        trace = self.dm.zeros((max(n, 1), max_trace, 11), torch.float64) if max_trace > 0 else None
        args = (self.dm.h, C.c_void_p(starts_t.data_ptr()), C.c_void_p(goals_t.data_ptr()), C.c_int64(n), C.c_int32(slots),
                C.c_int32(self.max_nodes), C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()), C.c_void_p(res.data_ptr()),
                C.c_void_p(paths.data_ptr()) if paths is not None else None, C.c_int32(self.max_path),
                C.c_void_p(trace.data_ptr()) if trace is not None else None, C.c_int32(max_trace))
        look = self._look_workspace(n) if (mode == 1 and (not profile or self.lookahead)) else None
        self.last_lookahead = look is not None
        order = None
        if self.longest_first and n > slots:
            d = ((starts_t[:, :2] - goals_t[:, :2]) ** 2).sum(1)
            order = torch.argsort(d, descending=True, stable=True).to(torch.int32)
        if profile and look is None and mode == 1:
            _native.chk(L.avp_plan_batch_profile(*args), "avp_plan_batch_profile")
        else:
            _native.chk(L.avp_plan_batch_ex(*args, C.c_int32(mode | (0x100 if profile else 0)),
                                            C.c_void_p(look.data_ptr()) if look is not None else None, C.c_int64(look.numel() if look is not None else 0),
                                            C.c_void_p(order.data_ptr()) if order is not None else None), "avp_plan_batch_ex")
        self._keep = (look, order)                          # (alive until the next call: the launch is asynchronous)
        self.last_time_sliced = bool(int(L.avp_plan_last_launch(self.dm.h)) & 1)      # the library's own report, not a re-derivation
        return res, paths, trace

    def plan(self, starts, goals, max_trace: int = 0) -> List[PlanResult]:
        starts = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
        goals = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 3)
        if len(starts) != len(goals):
            raise ValueError("starts and goals must have the same length")
        n = len(starts)
        if n == 0:
            return []
        res, paths, trace = self.plan_dev(self.dm.dev_tensor(starts), self.dm.dev_tensor(goals), True, max_trace)
        rec = res.cpu().numpy().view(RESULT_DTYPE).reshape(-1)[:n]
        if (rec["status"] < 0).any():
            raise RuntimeError(f"{int((rec['status'] < 0).sum())} searches of a time-sliced launch were never finished (status UNFINISHED): internal error")
        paths = paths.cpu().numpy()
        trace = trace.cpu().numpy() if trace is not None else None
        out = []
        for i in range(n):
            r = rec[i]
            nf, na, k = int(r["n_final"]), int(r["n_astar"]), int(r["rs_n"])
            cnt = {name: int(r[name]) for name in ("n_checks", "n_rs", "n_closed", "n_open", "h_cells", "h_misses",
                                                   "global_index", "n_nodes", "in_radius_last", "rs_collision")}
            if k > 0 and nf > 0:
                rs_xy = np.concatenate([np.asarray(r["rs_start"], dtype=np.float64)[None, :], paths[i, na:nf, :3]], 0)
                rs_dirs = np.concatenate([[int(r["rs_dir0"])], paths[i, na:nf, 3].astype(np.int64)]).astype(np.int8)
            else:
                rs_xy, rs_dirs = np.zeros((0, 3)), np.zeros(0, np.int8)
            out.append(PlanResult(status=int(r["status"]), n_pops=int(r["n_pops"]), final_path=paths[i, :nf, :3].copy(),
                                  astar_path=paths[i, :na, :3].copy(), rs_types=[("S", "L", "R")[int(t)] for t in r["rs_types"][:k]],
                                  rs_lengths=[float(v) for v in r["rs_lengths"][:k]], rs_L=float(r["rs_L"]),
                                  n_rs_pts=int(r["n_rs_pts"]), rs_xyyaw=rs_xy, rs_dirs=rs_dirs, counters=cnt,
                                  trace=None if trace is None else trace[i, :min(int(r["n_pops"]), trace.shape[1])].copy()))
        return out


class _OpenListView:
    """Read-only stand-in for `hybrid_a_star.open_list` (a `queue.PriorityQueue`, reference hybrid_a_star.py:96).
    The open list lives in the planner's device workspace; what the host knows exactly is its SIZE at the end of the last
    `a_star_plan()` (`qsize()`, `empty()` -- len(open_list.queue) of the reference, checked against the goldens). The
    node objects themselves are not mirrored: the device heap is a scratch structure (the kernel resolves the children of
    the final pop speculatively), so `.queue`, `get()` and `put()` raise instead of returning something subtly different."""

    def __init__(self):
        self._n = 0

    def qsize(self) -> int:
        return self._n

    def empty(self) -> bool:
        return self._n == 0

    def __len__(self) -> int:
        return self._n

    @property
    def queue(self):
        raise NotImplementedError("the open list's nodes stay on the device; only its size is mirrored (qsize())")

    def get(self, *a, **k):
        raise NotImplementedError("PathPlanner.planner.open_list is a read-only size view")

    put = get


class _PlannerView:
    """What the reference exposes as `PathPlanner.planner` (a hybrid_a_star instance): the pieces
    callers read (`ddt`, `dt`, `collision_checker`, `steering_angle`, `open_list`)."""

    def __init__(self, config, vehicle, checker):
        self.config = config
        self.vehicle = vehicle
        self.dt = config['dt']
        self.ddt = config['trajectory_dt']
        self.collision_checker = checker
        self.steering_angle = np.linspace(-vehicle.max_steering_angle, vehicle.max_steering_angle, config['steering_angle_num'])
        self.open_list = _OpenListView()


class PathPlanner:
    def __init__(self, config: dict = None, map: Map = None, vehicle: Vehicle = None) -> None:
        self.config = config
        self.map = map
        self.vehicle = vehicle
        if config['collision_check'] == 'circle':
            self.collision_checker = collision_check.two_circle_checker(map=map, vehicle=vehicle, config=config)
        else:
            self.collision_checker = collision_check.distance_checker(map=map, vehicle=vehicle, config=config)
        self.planner = _PlannerView(config, vehicle, self.collision_checker)
        self._batch: Optional[BatchPlanner] = None

    # -- batched surface -----------------------------------------------------------------------------
    def batch_planner(self, **kw) -> BatchPlanner:
        if self._batch is None or kw:
            self._batch = BatchPlanner(_native.device_map(self.map, self.vehicle, self.config), **kw)
        return self._batch

    def plan_batch(self, starts, goals, max_trace: int = 0, split: bool = False) -> List[PlanResult]:
        """N independent (start, goal) problems on this map; element i equals what
        `a_star_plan()` gives for `map.case.{x0..thetaf}` = (starts[i], goals[i]).
        split=True also runs `split_path` on every solved problem -- all extension poses of the batch in ONE
        collision-check launch -- and fills `segments` / `change_gear` (= what `path_planning()` returns,
        path_planner.py:45-56) or `split_error` where the reference raises IndexError (:181)."""
        res = self.batch_planner().plan(starts, goals, max_trace=max_trace)
        if split:
            todo = [r for r in res if r.rs_types and r.status in (0, 1)]
            outs = split_path_batch([r.final_path for r in todo], self.config, self.vehicle, self.collision_checker)
            for r, o in zip(todo, outs):
                if isinstance(o, Exception):
                    r.split_error = type(o).__name__
                else:
                    r.segments, r.change_gear = o
        return res

    # -- reference API -----------------------------------------------------------------------------------
    def a_star_plan(self) -> Tuple[List[List], List[List], PATH]:
        c = self.map.case
        r = self.plan_batch([[c.x0, c.y0, c.theta0]], [[c.xf, c.yf, c.thetaf]])[0]
        # The reference has no capacity limits: when the node arena or the path buffer of the default
        # workspace is exhausted (status CAPACITY) the search is repeated with both doubled, up to ~4 M nodes.
        bp = self.batch_planner()
        max_nodes, max_path = bp.max_nodes, bp.max_path
        while r.status == 5 and max_nodes < (1 << 22):
            max_nodes, max_path = 2 * max_nodes, 2 * max_path
            big = BatchPlanner(bp.dm, max_nodes=max_nodes, n_slots=1, max_path=max_path, lookahead=False)
            r = big.plan([[c.x0, c.y0, c.theta0]], [[c.xf, c.yf, c.thetaf]])[0]
        self.planner.open_list._n = int(r.counters.get("n_open", 0))
        if r.status in (0, 1):
            if not r.rs_types:
                raise AttributeError("'NoneType' object has no attribute 'x'")       # path_planner.py:104
        elif r.status == 3:
            raise AssertionError("path.L >= 0.01")                                   # rs_curve.py:153
        elif r.status == 7:
            # AVP_PLAN_BAD_POSE is this library's refusal, not a planner failure: its own exception type, so that a caller that
            # treats "any status but 0 / 1" as "no plan found" does not mistake it for one. The reference itself would still
            # return on a finite heading up to ~1e16 rad (rs_curve.py:648-655 needs |theta| / 2 pi loop trips: > 1.6e5 beyond 1e6 rad).
            raise ValueError("start / goal pose refused (AVP_PLAN_BAD_POSE): a coordinate or heading that is not finite, or a heading "
                             "beyond 1e6 rad (the reference's pi_2_pi loop needs more than 1.6e5 iterations there and never returns beyond ~1e16)")
        else:
            raise RuntimeError(f"hybrid A* stopped with status {r.status_name}")
        final_path = [[float(p[0]), float(p[1]), float(p[2])] for p in r.final_path]
        astar_path = [[float(p[0]), float(p[1]), float(p[2])] for p in r.astar_path]
        rs_path = PATH(lengths=list(r.rs_lengths), ctypes=list(r.rs_types), L=r.rs_L,
                       x=[float(v) for v in r.rs_xyyaw[:, 0]], y=[float(v) for v in r.rs_xyyaw[:, 1]],
                       yaw=[float(v) for v in r.rs_xyyaw[:, 2]], directions=[int(d) for d in r.rs_dirs])
        return final_path, astar_path, rs_path

    def split_path(self, final_path: List[List]) -> Tuple[List[List[List]], int]:
        # (through the batch form: every extension pose of the path in ONE check launch instead of a synchronous
        # upload + launch + download per pose; same segments -- tests/test_gpu_split.py runs both against the goldens)
        out = split_path_batch([final_path], self.config, self.vehicle, self.collision_checker)[0]
        if isinstance(out, Exception):
            raise out
        return out

    def path_planning(self) -> Tuple[List[List], Dict, List[List[List]]]:
        final_path, astar_path, rs_path = self.a_star_plan()
        split_path_list, change_gear = self.split_path(final_path)
        path_info = {'astar_path': astar_path, 'rs_path': rs_path, 'change_gear': change_gear}
        return sum(split_path_list, []), path_info, split_path_list


def _cosine_distance(u, v):
    """scipy.spatial.distance.cosine (1 - u.v / sqrt(u.u * v.v), clipped to [0, 2]); np.dot keeps
    the BLAS rounding the reference gets."""
    import math
    u = np.asarray(u, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    uv, uu, vv = np.dot(u, v), np.dot(u, u), np.dot(v, v)
    with np.errstate(all="ignore"):
        dist = 1.0 - uv / math.sqrt(uu * vv) if uu * vv > 0 else np.float64(np.nan)
    return np.clip(dist, 0.0, 2.0)


def split_path(final_path, config, vehicle, checker):
    """Gear-change segmentation + collision-checked end extension (`path_planner.py:112-192`).
    A cut is placed where consecutive steps point in opposite directions (cosine < 0); each
    segment end is extended by up to `extended_num` collision-free points which are also prepended
    (last first) to the next segment. Raises IndexError when the path has no gear change, like
    the reference (`path_planner.py:181`)."""
    segments: List[List[List]] = []
    change_gear = 0
    seg_start = 0
    n_extend = config['extended_num']
    ddt = config['trajectory_dt']
    carried = 0                      # extension points of the previous segment still to be prepended

    def prepend_carried(seg):
        prev = segments[-1]
        for j in range(carried):
            q = prev[-(carried - j)]
            seg.insert(0, [q[0], q[1], q[2]])

    for i in range(len(final_path) - 2):
        a, b, c = final_path[i], final_path[i + 1], final_path[i + 2]
        cosine = 1 - _cosine_distance((b[0] - a[0], b[1] - a[1]), (c[0] - b[0], c[1] - b[1]))
        if not (cosine < 0):
            continue
        change_gear += 1
        seg = final_path[seg_start:i + 2]
        if change_gear > 1 and carried > 0:
            prepend_carried(seg)
            carried = 0
        for j in range(n_extend):
            heading = a[2]
            moving_pos_x = b[0] > a[0]
            moving_neg_x = b[0] < a[0]
            facing_pos_x = -np.pi / 2 < heading < np.pi / 2
            facing_neg_x = (np.pi / 2 < heading < np.pi) or (-np.pi < heading < -np.pi / 2)
            forward = (moving_pos_x and facing_pos_x) or (moving_neg_x and facing_neg_x)
            speed = vehicle.max_v if forward else -vehicle.max_v
            step = speed * ddt * (j + 1)
            th = b[2]
            ex = b[0] + step * np.cos(th)
            ey = b[1] + step * np.sin(th)
            if not checker.check(node_x=ex, node_y=ey, theta=th):
                seg.append([ex, ey, th])
                carried += 1
        segments.append(seg)
        seg_start = i + 1

    tail = final_path[seg_start:]
    prev = segments[-1]              # IndexError here when there was no gear change (as the reference)
    if carried > 0:
        prepend_carried(tail)
    segments.append(tail)
    return segments, int(change_gear)


def _gear_cuts(P: np.ndarray) -> np.ndarray:
    """Indices i with cosine(P[i+1]-P[i], P[i+2]-P[i+1]) < 0 (path_planner.py:126-134), evaluated per triple with the
    scalar routine above so that the BLAS rounding of the reference's scipy call is kept."""
    return np.array([i for i in range(len(P) - 2)
                     if (1 - _cosine_distance((P[i + 1, 0] - P[i, 0], P[i + 1, 1] - P[i, 1]),
                                              (P[i + 2, 0] - P[i + 1, 0], P[i + 2, 1] - P[i + 1, 1]))) < 0], dtype=np.int64)


def _extension_poses(a, b, n_extend, vmax, ddt):
    """The `extended_num` candidate poses behind a cut (path_planner.py:142-166): same expressions as split_path."""
    heading = a[2]
    moving_pos_x = b[0] > a[0]
    moving_neg_x = b[0] < a[0]
    facing_pos_x = -np.pi / 2 < heading < np.pi / 2
    facing_neg_x = (np.pi / 2 < heading < np.pi) or (-np.pi < heading < -np.pi / 2)
    forward = (moving_pos_x and facing_pos_x) or (moving_neg_x and facing_neg_x)
    speed = vmax if forward else -vmax
    th = b[2]
    return [[b[0] + speed * ddt * (j + 1) * np.cos(th), b[1] + speed * ddt * (j + 1) * np.sin(th), th] for j in range(n_extend)]


def split_path_batch(paths, config, vehicle, checker):
    """`split_path` for many paths with ONE collision-check launch for all their extension poses.
    The candidate extension poses of a cut depend only on the path (not on each other's check results), so they are
    generated for every cut of every path first, checked together (`check_batch`), and the segments are then
    assembled with the reference's bookkeeping (`path_planner.py:112-192`). Returns one entry per path:
    `(segments, change_gear)` or the `IndexError` the reference raises when the path has no gear change."""
    n_extend = config['extended_num']
    ddt = config['trajectory_dt']
    plist = [[[float(q[0]), float(q[1]), float(q[2])] for q in p] for p in paths]
    cuts, cand, owner = [], [], []
    for k, p in enumerate(plist):
        P = np.asarray(p, dtype=np.float64).reshape(-1, 3)
        c = _gear_cuts(P)
        cuts.append(c)
        for i in c:
            cand += _extension_poses(p[i], p[i + 1], n_extend, vehicle.max_v, ddt)
    hit = checker.check_batch(np.asarray(cand, dtype=np.float64).reshape(-1, 3)) if cand else np.zeros(0, np.uint8)
    out = []
    pos = 0
    for k, p in enumerate(plist):
        if len(cuts[k]) == 0:
            out.append(IndexError("list index out of range"))        # segments[-1] on an empty list, path_planner.py:181
            continue
        segments: List[List[List]] = []
        seg_start = 0
        carried = 0
        for ci, i in enumerate(cuts[k]):
            seg = p[seg_start:i + 2]
            if ci > 0 and carried > 0:
                prev = segments[-1]
                for j in range(carried):
                    q = prev[-(carried - j)]
                    seg.insert(0, [q[0], q[1], q[2]])
                carried = 0
            for j in range(n_extend):
                if not hit[pos]:
                    seg.append(list(cand[pos]))
                    carried += 1
                pos += 1
            segments.append(seg)
            seg_start = i + 1
        tail = p[seg_start:]
        if carried > 0:
            prev = segments[-1]
            for j in range(carried):
                q = prev[-(carried - j)]
                tail.insert(0, [q[0], q[1], q[2]])
        segments.append(tail)
        out.append((segments, int(len(cuts[k]))))
    return out
