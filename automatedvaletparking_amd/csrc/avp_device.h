// avp_device.h -- device-side views of the map / parameters and the exact footprint test.
// Included by every kernel file of libavp_hip.so (single translation unit build).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/avp.h"
#include "avp_math.h"
#include "../../include/avp_libm.h"

#define AVP_LDS __attribute__((address_space(3)))     // pointers the called phases take: LDS addresses, 32 bits each

// static LDS taken by the trig tables in every kernel that evaluates trig
#define AVP_LDS_TABLE_BYTES (sizeof(AVP_SINCOS_TAB) + 1024)   /* + the Reeds-Shepp word tables (avp_rs_kernels.h) */

#if defined(__HIP_DEVICE_COMPILE__)
// Copy the sin/cos table into LDS; all threads of the workgroup, once, before any trig call. (WITH_ATAN is a leftover of
// the rounds that staged an atan table too: glibc's atan2 / asin / acos / tan / pow tables -- 46 KB, include/avp_glibc_tab.h --
// stay in global memory and are read through L1 / L2.)
template <bool WITH_ATAN>
__device__ __forceinline__ void avp_lds_tables_fill()
{
    const int n1 = (int)(sizeof(AVP_SINCOS_TAB) / sizeof(double));
    for (int i = threadIdx.x; i < n1; i += blockDim.x) (&AVP_SINCOS_LDS[0][0])[i] = (&AVP_SINCOS_TAB[0][0])[i];
    __syncthreads();
}
#else
template <bool WITH_ATAN> __device__ inline void avp_lds_tables_fill() {}       // host pass of hipcc: declaration only
#endif

// libm pow(v, 2) behind Python's ** / pow(v, 2) in two THRESHOLD tests of the reference:
//   two_circle_checker (collision_check.py:131-134)   np.sqrt(pow(dx, 2) + pow(dy, 2)) <= Rd
//   try_reach_goal (hybrid_a_star.py:308-310)         np.sqrt(dx ** 2 + dy ** 2) < flag_radius
// pow(v, 2) is NOT v*v (0.08 % of arguments differ in the last bit), but the two differ by <= 1 ulp, the sum by <= 2 ulp,
// its root by <= 2 ulp: outside a band of 8 ulp around the threshold the plain squares decide, inside it (one test in
// ~1e13 on real maps) the exact avp_pow2 -- the reference's booleans at the cost of one comparison. The exact form is a
// called function on the device: inlined, its table look-ups and 40 live doubles sat in the register budget of the
// planner's pop loop (29 spilled VGPRs in plan_kernel).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __noinline__ double avp_pow2_norm(double dx, double dy) { return sqrt(avp_pow2(dx) + avp_pow2(dy)); }
#else
static inline double avp_pow2_norm(double dx, double dy) { return sqrt(avp_pow2(dx) + avp_pow2(dy)); }
#endif
AVP_HD bool avp_circle_hit(double dx, double dy, double Rd)
{
    const double s = sqrt(dx * dx + dy * dy);
    if (fabs(s - Rd) > Rd * 0x1p-49) return s <= Rd;
    return avp_pow2_norm(dx, dy) <= Rd;
}
// ... and without the square root: d2 = dx*dx + dy*dy against Rd^2 (1 -+ 2^-40) settles all but one test in ~1e11 (s = sqrt(d2) is
// then further than Rd 2^-41 from Rd: avp_circle_hit would return the same through its own first branch); the rest take avp_circle_hit.
struct AvpRd2 { double lo, hi; };
AVP_HD AvpRd2 avp_circle_rd2(double Rd) { AvpRd2 r; r.lo = (Rd * Rd) * (1.0 - 0x1p-40); r.hi = (Rd * Rd) * (1.0 + 0x1p-40); return r; }
AVP_HD bool avp_circle_hit2(double dx, double dy, double Rd, const AvpRd2 r)
{
    const double d2 = dx * dx + dy * dy;
    if (d2 < r.lo) return true;
    if (d2 > r.hi) return false;
    return avp_circle_hit(dx, dy, Rd);          // (NaN lands here too: both comparisons false)
}
AVP_HD bool avp_within_radius(double dx, double dy, double radius)
{
    const double s = sqrt(dx * dx + dy * dy);
    if (fabs(s - radius) > radius * 0x1p-49) return s < radius;
    return avp_pow2_norm(dx, dy) < radius;
}

// Costmap resident in HBM. Column-major occupancy in two forms:
//  - obstacle points in np.where(cost_map == 255) order (sorted by ix, then iy): ox/oy + colStart
//  - per-column bitmaps: colBits[ix*wpc + (iy >> 6)] bit (iy & 63)
struct DevMap {
    int32_t nx, ny, S, Sy, P, wpc;
    double b0, b1, b2, b3, dx, dy;
    const double* X;          // nx node x coordinates (map_position[0])
    const double* Y;          // ny
    const uint8_t* occ;       // nx*ny, 255 = obstacle
    const double* ox;         // P
    const double* oy;         // P
    const int32_t* colStart;  // nx + 1
    const uint64_t* colBits;  // nx * wpc
};

// Inflated vehicle rectangle prepared for distance_checker.check (collision_check.py:144-195).
// 26 doubles = 208 bytes. rden[i] = 1 / den[i] serves the division-free first look of avp_footprint_point_hit.
struct Footprint {
    double cx[4], cy[4];      // rr, rf, lf, lr (map/costmap.py:106-113)
    double k[4], b[4], den[4];
    double wthr, lthr;        // v_lb - 0.01, v_length - 0.01
    double rden[4];
};
// The 22 doubles the point test reads on its fast path (the check kernel's LDS record: den is recomputed from k on the
// rare exact path, sqrt(1 + k*k) being the expression that produced it).
struct FootprintFast {
    double cx[4], cy[4];
    double k[4], b[4], rden[4];
    double wthr, lthr;
};

// Footprint corners: world = R(theta).local + (x, y), the 2x2 product rounded the way the
// reference's BLAS call rounds it: acc = a0*b0; acc = fma(a1, b1, acc)  (numerics).
// (the parameter block is any struct with the members fp_xr, fp_xf, fp_yr, fp_yl: avp_params or the planner's LDS copy)
// 1 / den for the first look of the point test; NaN (= "always take the exact path") for a subnormal slope, where the
// equality filter's relative bound does not hold (never seen: a slope is a quotient of metre-sized differences)
AVP_HD double avp_footprint_rden(double k, double den) { return (k != 0.0 && fabs(k) < 0x1p-1000) ? NAN : 1.0 / den; }
template <class P>
AVP_HD void avp_footprint_setup_cs(const P& p, double x, double y, double cs, double sn, Footprint& f)
{
    const double lx[4] = { p.fp_xr, p.fp_xf, p.fp_xf, p.fp_xr };
    const double ly[4] = { p.fp_yr, p.fp_yr, p.fp_yl, p.fp_yl };
#pragma unroll
    for (int i = 0; i < 4; i++) {
        f.cx[i] = AVP_FMA(-sn, ly[i], cs * lx[i]) + x;
        f.cy[i] = AVP_FMA(cs, ly[i], sn * lx[i]) + y;
    }
    double t0 = f.cx[0] - f.cx[3], t1 = f.cy[0] - f.cy[3];
    f.wthr = sqrt(t0 * t0 + t1 * t1) - 0.01;
    t0 = f.cx[3] - f.cx[2]; t1 = f.cy[3] - f.cy[2];
    f.lthr = sqrt(t0 * t0 + t1 * t1) - 0.01;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int j = (i + 1) & 3;
        f.k[i] = (f.cy[j] - f.cy[i]) / (f.cx[j] - f.cx[i]);   // +-inf / NaN when axis aligned, as numpy
        f.b[i] = f.cy[i] - f.k[i] * f.cx[i];
        f.den[i] = sqrt(1 + f.k[i] * f.k[i]);
        f.rden[i] = avp_footprint_rden(f.k[i], f.den[i]);
    }
}
AVP_HD void avp_footprint_setup(const avp_params& p, double x, double y, double th, Footprint& f)
{
    double cs, sn;
    avp_sincos(th, sn, cs);
    avp_footprint_setup_cs(p, x, y, cs, sn, f);
}

// AABB of the 4 corners (collision_check.py:49-52)
AVP_HD void avp_footprint_aabb(const Footprint& f, double& xmin, double& xmax, double& ymin, double& ymax)
{
    xmin = xmax = f.cx[0]; ymin = ymax = f.cy[0];
#pragma unroll
    for (int i = 1; i < 4; i++) {
        if (f.cx[i] > xmax) xmax = f.cx[i];
        if (f.cx[i] < xmin) xmin = f.cx[i];
        if (f.cy[i] > ymax) ymax = f.cy[i];
        if (f.cy[i] < ymin) ymin = f.cy[i];
    }
}

// One obstacle point against the rectangle (collision_check.py:197-238): inside test by point-line distances, exact
// corner test, exact edge-slope test -- the reference's booleans, eight IEEE divisions per point.
// den[i] of a record: stored (Footprint), or recomputed by the expression that produced it (a record without den)
template <class F> AVP_HD auto avp_footprint_den_(const F& f, int i, int) -> decltype(f.den[0] + 0.0) { return f.den[i]; }
template <class F> AVP_HD double avp_footprint_den_(const F& f, int i, long) { return sqrt(1 + f.k[i] * f.k[i]); }
template <class F>
AVP_HD bool avp_footprint_point_hit_exact(const F& f, double px, double py)
{
    double d[4];
#pragma unroll
    for (int i = 0; i < 4; i++) d[i] = fabs(f.k[i] * px + f.b[i] - py) / avp_footprint_den_(f, i, 0);
    const bool c1 = fabs(d[0] - d[2]) < f.wthr;
    const bool c2 = fabs(d[1] - d[3]) < f.lthr;
    if (c1 && c2) return true;
    bool on_x = false, on_y = false;
#pragma unroll
    for (int i = 0; i < 4; i++) { on_x |= (px == f.cx[i]); on_y |= (py == f.cy[i]); }
    if (on_x && on_y) return true;
    bool edge = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double k1 = (f.cy[i] - py) / (f.cx[i] - px);
        edge |= (k1 == f.k[i]);
    }
    return edge;
}
// The same booleans without a division in all but ~1 point in 1e12. The divisions only feed COMPARISONS, so a first look
// with bounded error settles them unless an operand sits within that bound of its threshold:
//  * distances: n_i * rden_i instead of n_i / den_i differs from the reference's quotient by <= 3.01 u d_i (u = 2^-53:
//    rden and the product round once each, the quotient once), so |d0 - d2| moves by <= 5.1 u (d0 + d2); the test
//    against wthr is settled when it misses wthr by more than 32 u (d0 + d2 + wthr) -- likewise d1, d3, lthr;
//  * slopes: fl(a / c) == k needs |k c - a| <= u |k c| (1 + u); fma(k, c, -a) rounds once, so |fma| > 8 u |fl(k c)| rules
//    equality out (k = 0: fma = -a exactly; k subnormal: rden is NaN by avp_footprint_rden; inf / NaN make every
//    comparison below false, which means "not settled").
// Anything not settled -- including every point of an axis-aligned pose, whose slopes are +-inf -- takes the exact form.
#ifndef AVP_POINT_FAST
#define AVP_POINT_FAST 1          // 0: every point takes the exact form (A/B builds: make variant DEFS=-DAVP_POINT_FAST=0)
#endif
template <class F>
AVP_HD bool avp_footprint_point_hit(const F& f, double px, double py)
{
    if (!AVP_POINT_FAST) return avp_footprint_point_hit_exact(f, px, py);
    double q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = fabs(f.k[i] * px + f.b[i] - py) * f.rden[i];
    const double D02 = fabs(q[0] - q[2]), D13 = fabs(q[1] - q[3]);
    bool settled = fabs(D02 - f.wthr) > (q[0] + q[2] + f.wthr) * 0x1p-48 && fabs(D13 - f.lthr) > (q[1] + q[3] + f.lthr) * 0x1p-48;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double c = f.cx[i] - px;
        settled = settled && fabs(AVP_FMA(f.k[i], c, -(f.cy[i] - py))) > fabs(f.k[i] * c) * 0x1p-50;
    }
    if (!settled) return avp_footprint_point_hit_exact(f, px, py);
    if (D02 < f.wthr && D13 < f.lthr) return true;
    bool on_x = false, on_y = false;
#pragma unroll
    for (int i = 0; i < 4; i++) { on_x |= (px == f.cx[i]); on_y |= (py == f.cy[i]); }
    return on_x && on_y;
}

// (the index searches avp_first_ge / avp_last_le / avp_first_gt / avp_last_lt / avp_node_search live in avp_math.h)

// map/costmap.py:319-329
AVP_HD int64_t avp_pos_to_index(const DevMap& m, double gx, double gy)
{
    const int64_t c = (int64_t)floor((gx - m.b0) / m.dx);
    const int64_t r = (int64_t)floor((m.b3 - gy) / m.dy);
    return c + r * (int64_t)m.S;
}
