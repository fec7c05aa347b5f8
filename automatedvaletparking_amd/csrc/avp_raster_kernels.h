// avp_raster_kernels.h -- obstacle-edge rasteriser (map/costmap.py:197-261).
//
// Split of the reference's detect_obstacle_edge:
//   host (numpy, kept there because np.arctan2/np.cos/np.sin are SIMD-dispatch sensitive): np.unique, centroid
//     angle sort, per edge the rotation angle, its cos/sin and the rotated edge length  -> the EDGE TABLE,
//     one row per polygon edge: [p1x, p1y, cos, sin, length, count = floor(length / dx)];
//   device (this file): per edge sample q of `count`: numpy.linspace(0, length, count)[q], rotation back
//     (np.dot(rot.T, [t; 0]) = (round(cos * t), round(sin * t)): the second product is an exact zero), + p1,
//     and the strict node search  X[i] < px  and  X[i] > px - dx  (:253-257); a point marks its cell when exactly
//     one node matches on each axis, is skipped when an axis has none (:259) and is counted in `multi` when an
//     axis has more than one (the reference raises there, :260).
// One thread per (edge, sample); writes are idempotent byte stores (255), so no atomics are needed.
#pragma once
#include "avp_device.h"

struct RasterGrid {
    const double* X;      // nx node abscissae (np.linspace(b0, b1, nx))
    const double* Y;      // ny node ordinates
    int32_t nx, ny;
    double b0, b2;        // X[0], Y[0]
    double dx, dy;        // X[1] - X[0], Y[1] - Y[0]
};

// (avp_linspace0 -- numpy.linspace(0.0, stop, num)[q] -- lives in avp_math.h, where the host tests reach it)

// cell of one sample, or -1 (no node on an axis) / -2 (more than one)
AVP_HD int64_t avp_raster_cell(const RasterGrid& g, const double* e, int q)
{
    const double p1x = e[0], p1y = e[1], ca = e[2], sa = e[3], length = e[4];
    const int count = (int)e[5];
    const double t = avp_linspace0(length, count, q);
    const double lx = ca * t, ly = sa * t;
    const double px = lx + p1x, py = ly + p1y;
    const int ilo = avp_first_gt(g.X, g.nx, g.b0, g.dx, px - g.dx), ihi = avp_last_lt(g.X, g.nx, g.b0, g.dx, px);
    const int jlo = avp_first_gt(g.Y, g.ny, g.b2, g.dy, py - g.dy), jhi = avp_last_lt(g.Y, g.ny, g.b2, g.dy, py);
    const int ni = ihi - ilo + 1, nj = jhi - jlo + 1;
    if (ni <= 0 || nj <= 0) return -1;          // len(points_x_index[0]) > 0 and len(points_y_index[0]) > 0
    if (ni > 1 || nj > 1) return -2;            // int() of a 2-element array raises in the reference
    return (int64_t)ilo * g.ny + jlo;
}

#if defined(__HIPCC__)
// grid: (ceil(max_count / 64), n_edges); occ is nx*ny bytes, zeroed by the caller
__global__ void rasterize_kernel(RasterGrid g, const double* __restrict__ edges, int64_t n_edges,
                                 uint8_t* __restrict__ occ, int32_t* __restrict__ multi)
{
    const int64_t ei = blockIdx.y;
    if (ei >= n_edges) return;
    const double* e = edges + ei * 6;
    const int count = (int)e[5];
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < count; q += gridDim.x * blockDim.x) {
        const int64_t cell = avp_raster_cell(g, e, q);
        if (cell >= 0) occ[cell] = 255;
        else if (cell == -2) atomicAdd(multi, 1);
    }
}

// Many maps in ONE launch (batched TPCAP ingest, map/costmap.py:134-156 + 197-261 for a list of scenario files): the edge tables of
// all maps concatenated, edge e belongs to map emap[e]; a map is a grid descriptor (its node tables inside one packed buffer) and
// the offset of its occupancy bytes inside one packed, zeroed buffer; multi[k] counts map k's multi-match samples.
struct RasterGridB { RasterGrid g; int64_t occ_off; };
__global__ void rasterize_batch_kernel(const RasterGridB* __restrict__ grids, int32_t n_maps, const double* __restrict__ edges, const int32_t* __restrict__ emap,
                                       int64_t n_edges, uint8_t* __restrict__ occ, int32_t* __restrict__ multi)
{
    const int64_t ei = (int64_t)blockIdx.y + (int64_t)blockIdx.z * 65535;
    if (ei >= n_edges) return;
    const int32_t k = emap[ei];
    if (k < 0 || k >= n_maps) return;        // (edge_map lives in device memory: the host cannot validate it; an edge of no map is skipped, never an out-of-bounds access)
    const RasterGridB gb = grids[k];
    const double* e = edges + ei * 6;
    const int count = (int)e[5];
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < count; q += gridDim.x * blockDim.x) {
        const int64_t cell = avp_raster_cell(gb.g, e, q);
        if (cell >= 0) occ[gb.occ_off + cell] = 255;
        else if (cell == -2) atomicAdd(multi + k, 1);
    }
}
#endif
