// avp_capi.hip -- the C-ABI of libavp_hip.so (see include/avp.h). Single translation unit:
// every kernel header is included here and compiled for gfx950 with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/avp.h"
#include "avp_device.h"
#include "avp_check_kernels.h"
#include "avp_rs_kernels.h"
#include "avp_plan_kernels.h"
#include "avp_planw_kernels.h"
#include "avp_raster_kernels.h"

static thread_local char g_err[512] = "";
static int32_t set_err(int32_t code, const char* fmt, const char* a = "", const char* b = "")
{
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}
#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return set_err(AVP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Every entry point that touches the device runs under this guard: the calling thread's current HIP device is
// restored on every exit path (a process that drives several GPUs keeps its own current device).
struct DeviceGuard {
    int prev = -1;
    bool armed = false;
    hipError_t enter(int device)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev == device) return hipSuccess;
        e = hipSetDevice(device);
        armed = e == hipSuccess;
        return e;
    }
    ~DeviceGuard() { if (armed) (void)hipSetDevice(prev); }
};
#define AVP_ON_DEVICE(dev) DeviceGuard guard_; HIPCHK(guard_.enter(dev))

struct avp_map {
    avp_params params;
    DevMap dev;
    int32_t device;
    hipStream_t stream;
    void* blob;          // one device allocation holding every table
    size_t blob_bytes;
    int32_t n_cu;
    // planner scratch owned by the handle (problem queue counter etc.)
    void* counters;
    int32_t slice_pops;  // time slice of the group forms in pops (0 = never park a search); avp_plan_set_slice_pops
    int32_t last_sliced, last_mode;   // what the last planner call on this handle did: time-sliced? which kernel form (avp_plan_last_launch)
    int32_t look_ent_log2;            // record store of the expansion lookahead: 2^this entries (avp_plan_set_look_entries)
};

extern "C" {
#define AVP_EXPORT __attribute__((visibility("default")))

AVP_EXPORT int32_t avp_version(void) { return AVP_VERSION; }
AVP_EXPORT int32_t avp_sizeof_params(void) { return (int32_t)sizeof(avp_params); }

AVP_EXPORT int32_t avp_last_error(char* buf, int32_t n)
{
    if (!buf || n <= 0) return AVP_ERR_ARG;
    strncpy(buf, g_err, (size_t)n - 1);
    buf[n - 1] = 0;
    return AVP_OK;
}

AVP_EXPORT int32_t avp_map_create(const avp_params* params, const uint8_t* occ, int32_t nx, int32_t ny, const double* xs,
                       const double* ys, const double boundary[4], const int32_t* obs_ix, const int32_t* obs_iy,
                       int32_t P, int32_t device, avp_map** out)
{
    // (round 6: up to AVP_MAX_NODES_PER_AXIS nodes per axis and 2^30 cells; rounds 1 - 5: 8 191 -- the cell indices the compacting kernels pack
    //  in 13 bits. Those kernels still serve maps up to 8 191 nodes per axis; larger ones take the lane-per-pose forms: same results)
    if (!params || !occ || !xs || !ys || !boundary || !out || nx < 2 || ny < 2 || nx > AVP_MAX_NODES_PER_AXIS || ny > AVP_MAX_NODES_PER_AXIS ||
        (int64_t)nx * ny > ((int64_t)1 << 30) || P < 0 || (P > 0 && (!obs_ix || !obs_iy)))
        return set_err(AVP_ERR_ARG, "avp_map_create: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(AVP_ERR_NOGPU, "no HIP device visible");
    if (device < 0 || device >= ndev) return set_err(AVP_ERR_ARG, "avp_map_create: device ordinal out of range");
    AVP_ON_DEVICE(device);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));

    avp_map* m = new avp_map();
    memset(m, 0, sizeof(*m));
    m->params = *params;
    m->device = device;
    m->stream = nullptr;
    m->n_cu = prop.multiProcessorCount;
    DevMap& d = m->dev;
    d.nx = nx; d.ny = ny; d.P = P;
    d.b0 = boundary[0]; d.b1 = boundary[1]; d.b2 = boundary[2]; d.b3 = boundary[3];
    d.dx = xs[1] - xs[0];                       // map/costmap.py:190-191
    d.dy = ys[1] - ys[0];
    d.S = (int32_t)((d.b1 - d.b0) / d.dx);      // map/costmap.py:328
    d.Sy = (int32_t)((d.b3 - d.b2) / d.dy);     // compute_h.py:245-246
    d.wpc = (ny + 63) / 64;

    // host-side tables
    std::vector<double> ox((size_t)P), oy((size_t)P);
    std::vector<int32_t> colStart((size_t)nx + 1, 0);
    std::vector<uint64_t> bits((size_t)nx * d.wpc, 0);
    for (int32_t q = 0; q < P; q++) {
        const int32_t ix = obs_ix[q], iy = obs_iy[q];
        if (ix < 0 || ix >= nx || iy < 0 || iy >= ny || occ[(size_t)ix * ny + iy] != 255 ||
            (q > 0 && (ix < obs_ix[q - 1] || (ix == obs_ix[q - 1] && iy <= obs_iy[q - 1])))) {
            delete m;
            return set_err(AVP_ERR_ARG, "avp_map_create: obstacle cells must be the np.where(cost_map==255) list");
        }
        ox[q] = xs[ix]; oy[q] = ys[iy];
        colStart[(size_t)ix + 1]++;
        bits[(size_t)ix * d.wpc + (iy >> 6)] |= 1ull << (iy & 63);
    }
    for (int32_t i = 0; i < nx; i++) colStart[(size_t)i + 1] += colStart[i];
    {
        // the list must be complete too: the kernels read obstacle cells through the bitmaps built from it
        int64_t n255 = 0;
        for (size_t c = 0; c < (size_t)nx * ny; c++) n255 += occ[c] == 255;
        if (n255 != P) { delete m; return set_err(AVP_ERR_ARG, "avp_map_create: obstacle cell list does not cover every 255 cell of the costmap"); }
    }

    // one blob: [X][Y][ox][oy][bits][colStart][occ]
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    const size_t oX = off; off = al(off + (size_t)nx * 8);
    const size_t oY = off; off = al(off + (size_t)ny * 8);
    const size_t oOx = off; off = al(off + (size_t)(P > 0 ? P : 1) * 8);
    const size_t oOy = off; off = al(off + (size_t)(P > 0 ? P : 1) * 8);
    const size_t oB = off; off = al(off + bits.size() * 8);
    const size_t oC = off; off = al(off + colStart.size() * 4);
    const size_t oOcc = off; off = al(off + (size_t)nx * ny);
    const size_t oCnt = off; off = al(off + 256);
    m->blob_bytes = off;
    if (hipMalloc(&m->blob, off) != hipSuccess) { delete m; return set_err(AVP_ERR_HIP, "hipMalloc of the map blob failed"); }
    char* base = (char*)m->blob;
    std::vector<char> host(off, 0);
    memcpy(host.data() + oX, xs, (size_t)nx * 8);
    memcpy(host.data() + oY, ys, (size_t)ny * 8);
    if (P > 0) { memcpy(host.data() + oOx, ox.data(), (size_t)P * 8); memcpy(host.data() + oOy, oy.data(), (size_t)P * 8); }
    memcpy(host.data() + oB, bits.data(), bits.size() * 8);
    memcpy(host.data() + oC, colStart.data(), colStart.size() * 4);
    memcpy(host.data() + oOcc, occ, (size_t)nx * ny);
    if (hipMemcpy(base, host.data(), off, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(m->blob); delete m;
        return set_err(AVP_ERR_HIP, "hipMemcpy of the map blob failed");
    }
    d.X = (const double*)(base + oX); d.Y = (const double*)(base + oY);
    d.ox = (const double*)(base + oOx); d.oy = (const double*)(base + oOy);
    d.colBits = (const uint64_t*)(base + oB); d.colStart = (const int32_t*)(base + oC);
    d.occ = (const uint8_t*)(base + oOcc);
    m->counters = base + oCnt;
    m->slice_pops = PW_SLICE_POPS;
    m->last_sliced = 0; m->last_mode = 0; m->look_ent_log2 = PL_LOOK_ENT_LOG2;
    *out = m;
    return AVP_OK;
}

AVP_EXPORT int32_t avp_map_destroy(avp_map* map)
{
    if (!map) return AVP_OK;
    DeviceGuard g;
    (void)g.enter(map->device);
    if (map->blob) (void)hipFree(map->blob);
    delete map;
    return AVP_OK;
}

AVP_EXPORT int32_t avp_map_set_stream(avp_map* map, void* hip_stream)
{
    if (!map) return set_err(AVP_ERR_ARG, "null map");
    map->stream = (hipStream_t)hip_stream;
    return AVP_OK;
}

AVP_EXPORT int32_t avp_sync(avp_map* map)
{
    if (!map) return set_err(AVP_ERR_ARG, "null map");
    AVP_ON_DEVICE(map->device);
    HIPCHK(hipStreamSynchronize(map->stream));
    return AVP_OK;
}

AVP_EXPORT int32_t avp_check_batch(avp_map* map, int32_t kind, const double* x, const double* y, const double* th, int64_t n,
                        uint8_t* out, int32_t variant)
{
    if (!map || n < 0 || (n > 0 && (!x || !y || !th || !out))) return set_err(AVP_ERR_ARG, "avp_check_batch: bad argument");
    if (n == 0) return AVP_OK;
    AVP_ON_DEVICE(map->device);
    const DevMap& d = map->dev;
    if (kind == 1) {
        const int64_t blocks = (n + 255) / 256;
        hipLaunchKernelGGL(check_circle_kernel, dim3((unsigned)blocks), dim3(256), 0, map->stream, d, map->params, x, y, th, n, out);
    } else if (kind == 0) {
        // the production kernel assumes the footprint AABB spans at most two 64-row bitmap words
        const double diag = sqrt((map->params.fp_xf - map->params.fp_xr) * (map->params.fp_xf - map->params.fp_xr) +
                                 (map->params.fp_yl - map->params.fp_yr) * (map->params.fp_yl - map->params.fp_yr));
        const bool rows_ok = diag / d.dy + 3.0 < 64.0;
        if (variant == 1 || !rows_ok || d.nx > 8191 || d.ny > 8191) {        // (the production kernel packs cell indices in 13 bits)
            const int64_t blocks = (n + 255) / 256;
            hipLaunchKernelGGL(check_distance_naive_kernel, dim3((unsigned)blocks), dim3(256), 0, map->stream, d, map->params, x, y, th, n, out);
        } else {
            // as many waves per workgroup (= per CU) as fit the LDS next to the staged map tables, at least 4; maps whose
            // tables leave no room for 4 are read through L1/L2 instead, with the full 8 waves
            int waves = CHK_WAVES;
            while (waves > 4 && check_distance_lds_bytes(d, true, waves) + AVP_LDS_TABLE_BYTES > 160 * 1024) waves--;
            const bool stage = check_distance_lds_bytes(d, true, waves) + AVP_LDS_TABLE_BYTES <= 160 * 1024;     // static LDS: the trig tables
            if (!stage) waves = CHK_WAVES;
            const size_t lds = check_distance_lds_bytes(d, stage, waves);
            const int64_t tiles = (n + 63) / 64;
            int64_t blocks = (tiles + waves - 1) / waves;
            const int64_t cap = (int64_t)map->n_cu;
            if (blocks > cap) blocks = cap;
            if (stage) {
                HIPCHK(hipFuncSetAttribute((const void*)check_distance_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(check_distance_kernel<true>, dim3((unsigned)blocks), dim3(64 * waves), lds, map->stream, d, map->params, x, y, th, n, out);
            } else {
                HIPCHK(hipFuncSetAttribute((const void*)check_distance_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(check_distance_kernel<false>, dim3((unsigned)blocks), dim3(64 * waves), lds, map->stream, d, map->params, x, y, th, n, out);
            }
        }
    } else
        return set_err(AVP_ERR_ARG, "avp_check_batch: kind must be 0 (distance) or 1 (circle)");
    HIPCHK(hipGetLastError());
    return AVP_OK;
}

AVP_EXPORT int32_t avp_corridor_batch(avp_map* map, double expand_dis, const double* x, const double* y, const double* th,
                                      int64_t n, double* out)
{
    return avp_corridor_batch_v(map, expand_dis, x, y, th, n, out, 0);
}

AVP_EXPORT int32_t avp_corridor_batch_v(avp_map* map, double expand_dis, const double* x, const double* y, const double* th,
                                        int64_t n, double* out, int32_t variant)
{
    if (!map || n < 0 || !(expand_dis >= 0.0) || (n > 0 && (!x || !y || !th || !out))) return set_err(AVP_ERR_ARG, "avp_corridor_batch: bad argument");
    if (n == 0) return AVP_OK;
    AVP_ON_DEVICE(map->device);
    const DevMap& d = map->dev;
    // production kernel: cell indices packed in 13 bits, way-point lane in 6; otherwise (or variant 1, the on-device
    // cross-check) the lane-per-way-point kernel
    // the production kernel walks at most three 64-row bitmap words per column: grown AABB height <= 128 rows
    const double diag_c = sqrt((map->params.fp_xf - map->params.fp_xr) * (map->params.fp_xf - map->params.fp_xr) +
                               (map->params.fp_yl - map->params.fp_yr) * (map->params.fp_yl - map->params.fp_yr));
    const bool rows_fit = (diag_c + 2.0 * expand_dis) / d.dy + 3.0 < 128.0;
    if (d.nx > 8191 || d.ny > 8191 || variant == 1 || !rows_fit) {
        hipLaunchKernelGGL(corridor_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, map->stream, map->dev, map->params, expand_dis,
                           x, y, th, n, out);
    } else {
        int waves = COR_WAVES;
        while (waves > 3 && corridor_lds_bytes(d, true, waves) + AVP_LDS_TABLE_BYTES > 160 * 1024) waves--;
        const bool stage = corridor_lds_bytes(d, true, waves) + AVP_LDS_TABLE_BYTES <= 160 * 1024;
        if (!stage) waves = COR_WAVES;
        const size_t lds = corridor_lds_bytes(d, stage, waves);
        const int64_t tiles = (n + 63) / 64;
        int64_t blocks = (tiles + waves - 1) / waves;
        const int64_t cap = (int64_t)map->n_cu;
        if (blocks > cap) blocks = cap;
        if (stage) {
            HIPCHK(hipFuncSetAttribute((const void*)corridor_compact_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(corridor_compact_kernel<true>, dim3((unsigned)blocks), dim3(64 * waves), lds, map->stream, d, map->params, expand_dis, x, y, th, n, out);
        } else {
            HIPCHK(hipFuncSetAttribute((const void*)corridor_compact_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(corridor_compact_kernel<false>, dim3((unsigned)blocks), dim3(64 * waves), lds, map->stream, d, map->params, expand_dis, x, y, th, n, out);
        }
    }
    HIPCHK(hipGetLastError());
    return AVP_OK;
}

AVP_EXPORT int32_t avp_trig_batch(avp_map* map, const double* x, int64_t n, double* out_sin, double* out_cos)
{
    if (!map || n <= 0 || !x || !out_sin || !out_cos) return set_err(AVP_ERR_ARG, "avp_trig_batch: bad argument");
    AVP_ON_DEVICE(map->device);
    hipLaunchKernelGGL(trig_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, map->stream, x, n, out_sin, out_cos);
    HIPCHK(hipGetLastError());
    return AVP_OK;
}

AVP_EXPORT int32_t avp_libm_batch(avp_map* map, int32_t kind, const double* a, const double* b, int64_t n, double* out)
{
    if (!map || n <= 0 || !a || !out || kind < 0 || kind > 4 || (kind == 0 && !b)) return set_err(AVP_ERR_ARG, "avp_libm_batch: bad argument");
    AVP_ON_DEVICE(map->device);
    hipLaunchKernelGGL(libm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, map->stream, kind, a, b, n, out);
    HIPCHK(hipGetLastError());
    return AVP_OK;
}

AVP_EXPORT int32_t avp_ieee_batch(avp_map* map, const double* a, const double* b, int64_t n, double* q, double* r, double* h)
{
    if (!map || n <= 0 || !a || !b || !q || !r || !h) return set_err(AVP_ERR_ARG, "avp_ieee_batch: bad argument");
    AVP_ON_DEVICE(map->device);
    hipLaunchKernelGGL(ieee_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, map->stream, a, b, n, q, r, h);
    HIPCHK(hipGetLastError());
    return AVP_OK;
}

AVP_EXPORT int32_t avp_rasterize_edges(int32_t device, void* stream, const double* xs, const double* ys, int32_t nx, int32_t ny,
                                       double x0, double dx, double y0, double dy, const double* edges, int64_t n_edges,
                                       int32_t max_count, uint8_t* occ, int32_t* multi)
{
    if (!xs || !ys || nx < 2 || ny < 2 || !(dx > 0) || !(dy > 0) || n_edges < 0 || (n_edges > 0 && !edges) || !occ || !multi)
        return set_err(AVP_ERR_ARG, "avp_rasterize_edges: bad argument");
    if (n_edges == 0 || max_count <= 0) return AVP_OK;
    if (n_edges > 65535) return set_err(AVP_ERR_ARG, "avp_rasterize_edges: more than 65535 edges in one call");
    DeviceGuard guard_;
    if (device >= 0) HIPCHK(guard_.enter(device));
    RasterGrid g;
    g.X = xs; g.Y = ys; g.nx = nx; g.ny = ny; g.dx = dx; g.dy = dy; g.b0 = x0; g.b2 = y0;
    hipLaunchKernelGGL(rasterize_kernel, dim3((unsigned)((max_count + 63) / 64), (unsigned)n_edges), dim3(64), 0, (hipStream_t)stream,
                       g, edges, n_edges, occ, multi);
    HIPCHK(hipGetLastError());
    return AVP_OK;
}

// Batched form ("batched TPCAP CSV ingest"): the edge tables of n_maps maps in ONE launch. nodes: device buffer
// holding every map's X then Y table back to back (map k: X at node_off[k], Y at node_off[k] + nx[k]); geo[k] = {x0, dx, y0, dy};
// edges / edge_map: the concatenated tables and the map of every edge (device); occ: one zeroed device buffer, map k's nx*ny bytes at
// occ_off[k]; multi: n_maps zeroed counters (device). The descriptors are built on the host and uploaded here (one small copy).
AVP_EXPORT int32_t avp_rasterize_edges_batch(int32_t device, void* stream, int32_t n_maps, const double* nodes, const int64_t* node_off,
                                             const int32_t* nx, const int32_t* ny, const double* geo, const int64_t* occ_off,
                                             const double* edges, const int32_t* edge_map, int64_t n_edges, int32_t max_count,
                                             uint8_t* occ, int32_t* multi, void* grid_scratch, int64_t grid_scratch_bytes)
{
    if (n_maps < 0 || n_edges < 0 || (n_maps > 0 && (!nodes || !node_off || !nx || !ny || !geo || !occ_off || !occ || !multi || !grid_scratch)) ||
        (n_edges > 0 && (!edges || !edge_map)))
        return set_err(AVP_ERR_ARG, "avp_rasterize_edges_batch: bad argument");
    if (n_maps == 0 || n_edges == 0 || max_count <= 0) return AVP_OK;
    if (grid_scratch_bytes < (int64_t)(n_maps * sizeof(RasterGridB))) return set_err(AVP_ERR_CAPACITY, "avp_rasterize_edges_batch: grid scratch too small");
    if (n_edges > (int64_t)65535 * 65535) return set_err(AVP_ERR_ARG, "avp_rasterize_edges_batch: too many edges");
    DeviceGuard guard_;
    if (device >= 0) HIPCHK(guard_.enter(device));
    std::vector<RasterGridB> h((size_t)n_maps);
    for (int32_t k = 0; k < n_maps; k++) {
        if (nx[k] < 2 || ny[k] < 2 || !(geo[4 * k + 1] > 0) || !(geo[4 * k + 3] > 0)) return set_err(AVP_ERR_ARG, "avp_rasterize_edges_batch: bad map geometry");
        if (node_off[k] < 0 || occ_off[k] < 0) return set_err(AVP_ERR_ARG, "avp_rasterize_edges_batch: negative buffer offset");
        RasterGrid& g = h[k].g;
        g.X = nodes + node_off[k]; g.Y = g.X + nx[k]; g.nx = nx[k]; g.ny = ny[k];
        g.b0 = geo[4 * k]; g.dx = geo[4 * k + 1]; g.b2 = geo[4 * k + 2]; g.dy = geo[4 * k + 3];
        h[k].occ_off = occ_off[k];
    }
    HIPCHK(hipMemcpyAsync(grid_scratch, h.data(), h.size() * sizeof(RasterGridB), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));        // (the pageable host vector goes out of scope)
    const unsigned gy = (unsigned)(n_edges < 65535 ? n_edges : 65535), gz = (unsigned)((n_edges + 65534) / 65535);
    hipLaunchKernelGGL(rasterize_batch_kernel, dim3((unsigned)((max_count + 63) / 64), gy, gz), dim3(64), 0, (hipStream_t)stream,
                       (const RasterGridB*)grid_scratch, n_maps, edges, edge_map, n_edges, occ, multi);
    HIPCHK(hipGetLastError());
    return AVP_OK;
}

#include "avp_capi_rs.inc"
#include "avp_capi_plan.inc"

}  // extern "C"
