"""Multi-GPU sharding of pose batches: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The hot path shards embarrassingly: problems (start, goal) are independent and the costmap is
read-only. The only exchanges are
  * one broadcast of the packed map blob from rank 0 (KB..MB, latency bound), and
  * one gather of fixed-stride result records + way-points to rank 0 per batch.
There is no all-reduce and no per-step collective inside the search.

Two ways to run a batch over N ranks (bench.py --gpus N):
  * `plan_weak`      every rank plans its own block of a world x per problem set, ONE gather to rank 0 at the end of the
                     step: no collective between the launches, per-GPU work fixed as N grows (weak scaling, throughput);
  * `two_stage_plan` one fixed problem set split over the ranks (strong scaling): a first stage on the index slice, the
                     records all-gathered (every rank must know which searches are still running), those dealt evenly,
                     planned, gathered; the way-points travel to rank 0 only.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np

from .costmap import Case, Map

_HDR = 16  # doubles in the blob header


def pack_map_blob(m: Map) -> np.ndarray:
    """float64 blob: header | boundary(4) | case poses(6) | cells (2*P) | obstacle vertex counts | vertices."""
    pk = m.pack()
    cells = np.stack([pk["obs_ix"], pk["obs_iy"]], 1).astype(np.float64).ravel()
    counts = np.array([len(o) for o in m.case.obs], dtype=np.float64)
    verts = np.concatenate([np.asarray(o, dtype=np.float64).ravel() for o in m.case.obs]) if len(m.case.obs) else np.zeros(0)
    c = m.case
    hdr = np.zeros(_HDR)
    hdr[0:6] = [pk["nx"], pk["ny"], len(pk["obs_ix"]), len(counts), len(verts), 1.0]
    return np.concatenate([hdr, np.asarray(m.boundary, dtype=np.float64),
                           np.array([c.x0, c.y0, c.theta0, c.xf, c.yf, c.thetaf], dtype=np.float64), cells, counts, verts])


def unpack_map_blob(blob: np.ndarray) -> Map:
    blob = np.asarray(blob, dtype=np.float64)
    nx, ny, P, nobs, nverts = [int(v) for v in blob[0:5]]
    o = _HDR
    boundary = blob[o:o + 4]; o += 4
    poses = blob[o:o + 6]; o += 6
    cells = blob[o:o + 2 * P].reshape(P, 2).astype(np.int64); o += 2 * P
    counts = blob[o:o + nobs].astype(np.int64); o += nobs
    verts = blob[o:o + nverts]
    case = Case()
    case.x0, case.y0, case.theta0, case.xf, case.yf, case.thetaf = [float(v) for v in poses]
    case.xmin = min(case.x0, case.xf) - 12
    case.xmax = max(case.x0, case.xf) + 12
    case.ymin = min(case.y0, case.yf) - 12
    case.ymax = max(case.y0, case.yf) + 12
    case.obs_num = nobs
    case.obs = []
    p = 0
    for k in counts:
        case.obs.append(verts[p:p + 2 * k].reshape(k, 2).copy())
        p += 2 * k
    return Map.from_cells(case, boundary, nx, ny, cells)


def broadcast_map(m: Optional[Map], src: int = 0, device=None) -> Map:
    """Rank `src` passes its Map; every rank returns an identical Map (one size + one payload broadcast)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return m
    rank = dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    if rank == src:
        blob = pack_map_blob(m)
        size = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    else:
        blob = None
        size = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.broadcast(size, src=src)
    t = torch.as_tensor(blob, device=dev) if rank == src else torch.empty(int(size.item()), dtype=torch.float64, device=dev)
    dist.broadcast(t, src=src)
    return m if rank == src else unpack_map_blob(t.cpu().numpy())


def shard_indices(starts: np.ndarray, goals: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Deal problems round-robin in order of decreasing start-goal distance (cost proxy: the
    heuristic sweep grows with that distance), ties by index: every rank gets a similar mix."""
    starts = np.asarray(starts, dtype=np.float64).reshape(-1, 3)
    goals = np.asarray(goals, dtype=np.float64).reshape(-1, 3)
    d = np.hypot(starts[:, 0] - goals[:, 0], starts[:, 1] - goals[:, 1])
    order = np.lexsort((np.arange(len(d)), -d))
    return order[rank::world]


def shard_problems(starts: np.ndarray, goals: np.ndarray, rank: int, world: int):
    """This rank's share of one problem set for the strong-scaling step (bench.py): `shard_indices` deals the
    problems, every rank gets the same number `per` = ceil(n / world) of rows so that the results can travel in ONE
    all_gather_into_tensor of fixed shape; short shards are padded with start == goal problems (they end at once with
    status RS_ERROR) whose index is -1. Returns (starts_local, goals_local, idx_padded, per)."""
    starts = np.asarray(starts, dtype=np.float64).reshape(-1, 3)
    goals = np.asarray(goals, dtype=np.float64).reshape(-1, 3)
    idx = shard_indices(starts, goals, rank, world)
    per = (len(starts) + world - 1) // world
    pad = per - len(idx)
    s_l = np.concatenate([starts[idx], np.tile(goals[:1], (pad, 1))]) if pad else starts[idx]
    g_l = np.concatenate([goals[idx], np.tile(goals[:1], (pad, 1))]) if pad else goals[idx]
    return np.ascontiguousarray(s_l), np.ascontiguousarray(g_l), np.concatenate([idx, -np.ones(pad, np.int64)]), per


def unshard_rows(idx_all: np.ndarray, *blocks):
    """Undo the deal: idx_all = the gathered padded index vectors (world * per entries, -1 = padding), blocks = gathered
    arrays whose first axis runs over the same world * per rows. Returns the blocks in original problem order."""
    idx_all = np.asarray(idx_all).reshape(-1)
    keep = idx_all >= 0
    order = np.argsort(idx_all[keep], kind="stable")
    return [np.asarray(b)[keep][order] for b in blocks]


def gather_records(local_idx: np.ndarray, local_rec: np.ndarray, n_total: int, dst: int = 0, device=None) -> Optional[np.ndarray]:
    """Gather fixed-stride float64 records (one row per problem) to rank `dst`, restoring the
    original problem order. local_rec: (len(local_idx), stride). Returns (n_total, stride) on dst."""
    import torch
    import torch.distributed as dist
    local_rec = np.ascontiguousarray(local_rec, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = np.zeros((n_total, local_rec.shape[1]))
        out[local_idx] = local_rec
        return out
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    per = (n_total + world - 1) // world
    stride = local_rec.shape[1]
    buf = torch.zeros((per, stride + 1), dtype=torch.float64, device=dev)
    buf[:, 0] = -1
    if len(local_idx):
        buf[:len(local_idx), 0] = torch.as_tensor(local_idx.astype(np.float64), device=dev)
        buf[:len(local_idx), 1:] = torch.as_tensor(local_rec, device=dev)
    allb = torch.empty((world, per, stride + 1), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(allb, buf) if dist.get_backend() == "nccl" else dist.all_gather(list(allb.unbind(0)), buf)
    if rank != dst:
        return None
    flat = allb.reshape(-1, stride + 1).cpu().numpy()
    flat = flat[flat[:, 0] >= 0]
    out = np.zeros((n_total, stride))
    out[flat[:, 0].astype(np.int64)] = flat[:, 1:]
    return out


def all_gather_rows(local, out=None):
    """All-gather equally shaped row blocks (any dtype) from every rank: returns (world, *local.shape).
    nccl/RCCL: one all_gather_into_tensor; gloo (CPU tests): list form. `out` may be a preallocated buffer."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local.unsqueeze(0)
    world = dist.get_world_size()
    local = local.contiguous()
    if out is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, local)
    else:
        dist.all_gather(list(out.unbind(0)), local)
    return out


# ---- the two-stage deal (bench.py --gpus N) ------------------------------------------------------------------
# The start-goal distance does not predict the length of a search (measured: DESIGN.md), and a fifth of the random pose
# pairs hold 95 % of a batch's expansions, so a static deal by distance leaves the slowest rank ~15 % above the mean at
# 8 ranks. The staged planner already separates the two kinds of searches; across GPUs its stages become:
#   stage 1  every rank runs the first stage (wave form, stage_pops pops) on the index slice [rank::world];
#            one all-gather of the fixed-stride records and paths -- four searches out of five are final here;
#   stage 2  the searches still running (status AVP_PLAN_DEFERRED), in index order, are dealt round-robin -- they are
#            the long ones, every rank gets the same number of them --, planned from scratch, and all-gathered.
# Row (w, j) of a gathered block is problem j * world + w of the dealt list, so the merge is a permute + one index_copy.
DEFERRED = 100


def deal_slice(n: int, rank: int, world: int):
    """Indices [rank::world] of a list of n, padded with -1 to per = ceil(n / world)."""
    per = (n + world - 1) // world
    idx = np.arange(rank, n, world, dtype=np.int64)
    return np.concatenate([idx, -np.ones(per - len(idx), np.int64)]), per


PAD_POSE = (0.0, 0.0, 0.0)         # padding of an EMPTY problem list only (nothing to borrow a pose from)


def take_padded(starts, goals, idx):
    """Problems idx (-1 = padding: a start == goal problem, which ends at its first pop with status RS_ERROR,
    rs_curve.py:153). The padding pose is the list's first goal -- a pose INSIDE the map: the heuristic sweep's lattice
    set-up walks from the goal to the map's borders before the start == goal shortcut is reached, so a pose far outside
    (the origin, for a map in UTM-like coordinates) would cost millions of iterations and end as LATTICE instead. Padding
    results are discarded either way."""
    idx = np.asarray(idx, dtype=np.int64)
    goals_a = np.asarray(goals, dtype=np.float64).reshape(-1, 3)
    pad = goals_a[0] if len(goals_a) else np.array(PAD_POSE, dtype=np.float64)
    s_l = np.tile(pad, (len(idx), 1))
    g_l = s_l.copy()
    keep = idx >= 0
    if keep.any():
        s_l[keep], g_l[keep] = np.asarray(starts, dtype=np.float64)[idx[keep]], np.asarray(goals, dtype=np.float64)[idx[keep]]
    return np.ascontiguousarray(s_l), np.ascontiguousarray(g_l)


def gather_rows(local, dst: int = 0):
    """Gather equally shaped row blocks (any dtype) to rank dst: (world, *local.shape) there, None elsewhere. One
    point-to-point-per-rank collective (RCCL gather); no rank but dst receives anything."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local.unsqueeze(0)
    world, rank = dist.get_world_size(), dist.get_rank()
    local = local.contiguous()
    if rank == dst:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, list(out.unbind(0)), dst=dst)
        return out
    dist.gather(local, None, dst=dst)
    return None


def plan_weak(plan_block, starts, goals, rank: int, world: int, dst: int = 0):
    """Weak-scaling step: the GLOBAL set holds world x per problems, rank r plans the contiguous block
    [r * per, (r + 1) * per) with plan_block(starts (per,3), goals (per,3)) -> (records (per, stride) uint8 tensor,
    paths (per, max_path, 4) float64 tensor) and the blocks are gathered to rank dst. No other collective.
    Returns (records (n, stride), paths (n, max_path, 4)) in problem order on dst, (None, None) elsewhere."""
    starts = np.asarray(starts, dtype=np.float64).reshape(-1, 3)
    goals = np.asarray(goals, dtype=np.float64).reshape(-1, 3)
    n = len(starts)
    assert n % world == 0, "the weak-scaling set holds the same number of problems for every rank"
    per = n // world
    r, p = plan_block(starts[rank * per:(rank + 1) * per], goals[rank * per:(rank + 1) * per])
    gr, gp = gather_rows(r, dst), gather_rows(p, dst)
    if gr is None:
        return None, None
    return gr.reshape((n,) + tuple(r.shape[1:])), gp.reshape((n,) + tuple(p.shape[1:]))


def gathered_in_list_order(g, count: int):
    """(world, per, ...) gathered block -> the first `count` rows in dealt-list order (row (w, j) = item j * world + w)."""
    return g.transpose(0, 1).reshape((-1,) + tuple(g.shape[2:]))[:count]


def record_status(rec_t):
    """int32 status column of a (k, stride) uint8 record tensor (avp_plan_result.status is the first field)."""
    import torch
    return rec_t[:, :4].contiguous().view(torch.int32).reshape(-1)


def two_stage_plan(stage1, stage2, starts, goals, rank: int, world: int, paths_to: Optional[int] = None):
    """One step of the two-stage deal. stage1 / stage2: callables (starts (k,3), goals (k,3)) -> (records (k, stride)
    uint8 tensor, paths (k, max_path, 4) float64 tensor) on the communication device; stage1 leaves the searches it does
    not finish with status DEFERRED (AVP_PLAN_DEFERRED; the library uses the same value for a search the wave form
    cannot hold at all, which is dealt like a long one). The records are all-gathered after each stage -- every rank
    needs the statuses to deal the second stage --; the way-points (max_path x 4 doubles per problem, 98 % of the
    payload) are all-gathered too when paths_to is None, and GATHERED to that rank alone otherwise. Returns
    (records (n, stride), paths (n, max_path, 4) or None on a rank that did not receive them, deferred indices (numpy)),
    in problem order. An empty problem list returns at once, without a collective (every rank sees the same n)."""
    import torch
    starts = np.asarray(starts, dtype=np.float64).reshape(-1, 3)
    goals = np.asarray(goals, dtype=np.float64).reshape(-1, 3)
    n = len(starts)
    if n == 0:
        r0, p0 = stage1(starts, goals)
        return r0, p0, np.zeros(0, np.int64)
    collect = (lambda t: all_gather_rows(t)) if paths_to is None else (lambda t: gather_rows(t, paths_to))
    idx1, _ = deal_slice(n, rank, world)
    r1, p1 = stage1(*take_padded(starts, goals, idx1))
    rec = gathered_in_list_order(all_gather_rows(r1), n).contiguous()
    gp = collect(p1)
    paths = gathered_in_list_order(gp, n).contiguous() if gp is not None else None
    deferred_t = (record_status(rec) == DEFERRED).nonzero().reshape(-1)
    deferred = deferred_t.cpu().numpy()                       # (the one host synchronisation of the step: stage 2's shape)
    nd = len(deferred)
    if nd:
        idx2, _ = deal_slice(nd, rank, world)
        share = np.where(idx2 >= 0, deferred[np.where(idx2 >= 0, idx2, 0)], -1)
        r2, p2 = stage2(*take_padded(starts, goals, share))
        rec.index_copy_(0, deferred_t, gathered_in_list_order(all_gather_rows(r2), nd))
        gp2 = collect(p2)
        if gp2 is not None:
            paths.index_copy_(0, deferred_t, gathered_in_list_order(gp2, nd))
    return rec, paths, deferred


def simulate_deals(n_pops, status, starts, goals, stage_pops: int, worlds=(2, 4, 8)):
    """Predicted load of the busiest rank relative to the mean (max / mean - 1) from the per-problem pop counts of a
    finished run, for the deal by decreasing start-goal distance (`shard_indices`) and for the two-stage deal."""
    n_pops = np.asarray(n_pops, dtype=np.float64)
    out = {}
    n = len(n_pops)
    for w in worlds:
        by_dist = [n_pops[shard_indices(starts, goals, r, w)].sum() for r in range(w)]
        deferred = np.where(n_pops > stage_pops)[0]
        s1 = [np.minimum(n_pops[np.arange(r, n, w)], stage_pops).sum() for r in range(w)]
        s2 = [n_pops[deferred[r::w]].sum() for r in range(w)]
        two = [a + b for a, b in zip(s1, s2)]
        out[str(w)] = {"by_distance_imbalance": float(max(by_dist) / np.mean(by_dist) - 1.0),
                       "two_stage_imbalance": float(max(two) / np.mean(two) - 1.0),
                       "two_stage_extra_pops_frac": float((sum(two) - n_pops.sum()) / n_pops.sum())}
    return out


def plan_sharded(plan_fn: Callable[[np.ndarray, np.ndarray], np.ndarray], starts, goals, dst: int = 0, device=None):
    """Run `plan_fn(starts_shard, goals_shard) -> (k, stride) float64 records` on this rank's shard
    and gather to rank dst. The result is identical for every world size (shard invariance)."""
    import torch.distributed as dist
    starts = np.asarray(starts, dtype=np.float64).reshape(-1, 3)
    goals = np.asarray(goals, dtype=np.float64).reshape(-1, 3)
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    idx = shard_indices(starts, goals, rank, world)
    rec = plan_fn(starts[idx], goals[idx]) if len(idx) else np.zeros((0, 1))
    if len(idx) == 0:
        # stride must agree across ranks: ask plan_fn for an empty result shape
        rec = plan_fn(starts[:0], goals[:0])
    return gather_records(idx, rec, len(starts), dst=dst, device=device)


def result_records(results, max_pts: int = 128) -> np.ndarray:
    """Fixed-stride float64 record per PlanResult: [status, n_pops, n_final, rs_L, path(max_pts*3)]."""
    out = np.zeros((len(results), 4 + 3 * max_pts))
    for i, r in enumerate(results):
        k = min(len(r.final_path), max_pts)
        out[i, 0:4] = [r.status, r.n_pops, len(r.final_path), r.rs_L]
        out[i, 4:4 + 3 * k] = r.final_path[:k].ravel()
    return out
