// avp_math.h -- scalar fp64 maths shared by every kernel of the hybrid-A* hot path.
//
// The reference evaluates the bicycle-model expansion and the footprint corners with numpy /
// CPython floats, i.e. with glibc 2.35 libm on x86-64 (FMA ifunc variant). Closed/open-list
// membership in the reference is exact float equality (path_plan/hybrid_a_star.py:156,170), so the
// expansion trig must agree with that libm to the last bit. avp_sin/avp_cos below restate glibc's
// published algorithm for |x| < 105414350 (IBM Accurate Mathematical Library lineage: 1/128-step
// double-double table + short polynomials, 3-part pi/2 reduction) with the FMA contractions of the
// x86-64 FMA build written out explicitly; the table is regenerated from first principles by
// gen_sincos_table.py. tests/test_math_host.py checks bit equality against this host's libm.
//
// THIRD-PARTY NOTICE. avp_sin / avp_cos / avp_sincos and their helpers (the "glibc 2.35 s_sin.c restated" section:
// branch thresholds, polynomial coefficients, the 3-part pi/2 reduction constants and the table layout) follow
// the algorithm of the GNU C Library's sysdeps/ieee754/dbl-64/s_sin.c, dosincos.c and branred-free paths:
//   IBM Accurate Mathematical Library, written by International Business Machines Corp.
//   Copyright (C) 2001-2022 Free Software Foundation, Inc.
//   The GNU C Library is free software; you can redistribute it and/or modify it under the terms of the GNU
//   Lesser General Public License as published by the Free Software Foundation; either version 2.1 of the
//   License, or (at your option) any later version.
// The code here is an independent restatement of that published algorithm (no glibc source text is included); the
// 440-entry table is regenerated from first principles by gen_sincos_table.py. avp_hypot restates CPython's
// Modules/mathmodule.c vector_norm (Python Software Foundation License v2).
//
// Everything here compiles for the device (hipcc, gfx950) and for the host (gcc, used only by
// the CPU-side unit tests of this header). Build with -ffp-contract=off: every fused operation
// is spelled AVP_FMA.
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define AVP_HD __host__ __device__ __forceinline__
#define AVP_D __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#define AVP_TAB_QUAL static __device__ const
#else
#define AVP_TAB_QUAL static const
#endif
#else
#define AVP_HD static inline
#define AVP_D static inline
#define AVP_TAB_QUAL static const
#endif

#include "avp_sincos_tab.h"

// Device code reads the trig tables from LDS: a lookup sits on the dependent chain of every sin/cos/atan
// (dozens per expanded node), and an LDS read costs ~1/10 of a global one whose L1 line the kernel's scratch
// traffic keeps evicting. Every kernel that evaluates trig calls avp_lds_tables_fill() first (avp_device.h).
#if defined(__HIP_DEVICE_COMPILE__)
__shared__ double AVP_SINCOS_LDS[sizeof(AVP_SINCOS_TAB) / sizeof(AVP_SINCOS_TAB[0])][4];
#define AVP_SCT AVP_SINCOS_LDS
#else
#define AVP_SCT AVP_SINCOS_TAB
#endif

#define AVP_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define AVP_PI 3.141592653589793

AVP_HD uint64_t avp_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
AVP_HD double avp_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

// ---- glibc 2.35 s_sin.c restated -------------------------------------------------------------
namespace avp_trig {
constexpr double sn3 = -1.66666666666664880952546298448555E-01;
constexpr double sn5 = 8.33333214285722277379541354343671E-03;
constexpr double cs2 = 4.99999999999999999999950396842453E-01;
constexpr double cs4 = -4.16666666666664434524222570944589E-02;
constexpr double cs6 = 1.38888874007937613028114285595617E-03;
constexpr double s1 = -0x1.5555555555555p-3;
constexpr double s2 = 0x1.1111111110ECEp-7;
constexpr double s3 = -0x1.A01A019DB08B8p-13;
constexpr double s4 = 0x1.71DE27B9A7ED9p-19;
constexpr double s5 = -0x1.ADDFFC2FCDF59p-26;
constexpr double big = 0x1.8p45;
constexpr double hp0 = 0x1.921fb54442d18p+0;    // pi/2 high
constexpr double hp1 = 0x1.1a62633145c07p-54;   // pi/2 low
constexpr double mp1 = 0x1.921FB58000000p+0;    // pi/2 in 27-bit pieces
constexpr double mp2 = -0x1.dde973c000000p-27;
constexpr double pp3 = -0x1.cb3b398000000p-55;
constexpr double pp4 = -0x1.d747f23e32ed7p-83;
constexpr double hpinv = 0x1.45f306dc9c883p-1;  // 2/pi
constexpr double toint = 0x1.8p52;

AVP_HD double taylor_sin(double xx, double x, double dx)
{
    double p = AVP_FMA(s5, xx, s4);
    p = AVP_FMA(p, xx, s3);
    p = AVP_FMA(p, xx, s2);
    p = AVP_FMA(p, xx, s1);
    double t = AVP_FMA(p, x, -(0.5 * dx));
    t = AVP_FMA(t, xx, dx);
    return x + t;
}

AVP_HD double do_sin(double x, double dx)
{
    const double xold = x;
    if (fabs(x) < 0.126) return taylor_sin(x * x, x, dx);
    if (x <= 0) dx = -dx;
    const double ux = big + fabs(x);
    const int k = (int)(avp_d2u(ux) & 0xffffffffu);
    x = fabs(x) - (ux - big);
    const double xx = x * x;
    const double s = x + AVP_FMA(x * xx, AVP_FMA(xx, sn5, sn3), dx);
    const double c = AVP_FMA(x, dx, xx * AVP_FMA(xx, AVP_FMA(xx, cs6, cs4), cs2));
    const double sn = AVP_SCT[k][0], ssn = AVP_SCT[k][1];
    const double cs = AVP_SCT[k][2], ccs = AVP_SCT[k][3];
    const double cor = AVP_FMA(cs, s, AVP_FMA(-sn, c, AVP_FMA(s, ccs, ssn)));
    return copysign(sn + cor, xold);
}

AVP_HD double do_cos(double x, double dx)
{
    if (x < 0) dx = -dx;
    const double ux = big + fabs(x);
    const int k = (int)(avp_d2u(ux) & 0xffffffffu);
    x = fabs(x) - (ux - big) + dx;
    const double xx = x * x;
    const double s = AVP_FMA(x * xx, AVP_FMA(xx, sn5, sn3), x);
    const double c = xx * AVP_FMA(xx, AVP_FMA(xx, cs6, cs4), cs2);
    const double sn = AVP_SCT[k][0], ssn = AVP_SCT[k][1];
    const double cs = AVP_SCT[k][2], ccs = AVP_SCT[k][3];
    const double cor = AVP_FMA(-sn, s, AVP_FMA(-cs, c, AVP_FMA(-s, ssn, ccs)));
    return cs + cor;
}

AVP_HD int reduce_sincos(double x, double* a, double* da)
{
    const double t = AVP_FMA(x, hpinv, toint);
    const double xn = t - toint;
    const double y = AVP_FMA(-xn, mp2, AVP_FMA(-xn, mp1, x));
    const int n = (int)(avp_d2u(t) & 3);
    const double t2 = AVP_FMA(-xn, pp3, y);
    double db = AVP_FMA(-xn, pp3, (y - t2));
    const double b = AVP_FMA(-xn, pp4, t2);
    db += AVP_FMA(-xn, pp4, (t2 - b));
    *a = b;
    *da = db;
    return n;
}

AVP_HD double do_sincos(double a, double da, int n)
{
    const double r = (n & 1) ? do_cos(a, da) : do_sin(a, da);
    return (n & 2) ? -r : r;
}
}  // namespace avp_trig

// sin / cos bit-identical to glibc 2.35 (x86-64 FMA variant) for |x| < 105414350; NaN beyond.
AVP_HD double avp_sin(double x)
{
    using namespace avp_trig;
    const int32_t k = 0x7fffffff & (int32_t)(avp_d2u(x) >> 32);
    if (k < 0x3e500000) return x;
    if (k < 0x3feb6000) return do_sin(x, 0);
    if (k < 0x400368fd) { const double t = hp0 - fabs(x); return copysign(do_cos(t, hp1), x); }
    if (k < 0x419921FB) { double a, da; const int n = reduce_sincos(x, &a, &da); return do_sincos(a, da, n); }
    return NAN;
}

AVP_HD double avp_cos(double x)
{
    using namespace avp_trig;
    const int32_t k = 0x7fffffff & (int32_t)(avp_d2u(x) >> 32);
    if (k < 0x3e400000) return 1.0;
    if (k < 0x3feb6000) return do_cos(x, 0);
    if (k < 0x400368fd) { const double y = hp0 - fabs(x); const double a = y + hp1; const double da = (y - a) + hp1; return do_sin(a, da); }
    if (k < 0x419921FB) { double a, da; const int n = reduce_sincos(x, &a, &da); return do_sincos(a, da, n + 1); }
    return NAN;
}

// sin and cos of the same argument, the values of avp_sin(x) and avp_cos(x) bit for bit, WITHOUT range branches.
// A wave whose lanes hold angles from different ranges pays every branch of a branchy dispatch (measured on MI355X:
// 2 300 cycles per wave-call with angles spread over [-pi, pi] against 610 with one common range -- and nearly every
// call of the planner is of the first kind). Every range of glibc's algorithm ends in ONE do_sin and ONE do_cos
// evaluation, only on different (argument, correction) pairs:
//   |x| < 0.855      : sin = do_sin(x, 0)                       cos = do_cos(x, 0)
//   |x| < 2.426      : sin = +-do_cos(pi/2 - |x|, hp1)          cos = do_sin(a, da),  a + da = pi/2 - |x| + hp1
//   |x| < 105414350  : (a, da, n) = pi/2 reduction;  sin, cos = +-do_sin(a, da), +-do_cos(a, da) picked by n
// So the three argument pairs are prepared for every lane (a dozen FMAs), selected per lane, do_sin / do_cos run once
// on full waves (do_sin's own |x| < 0.126 split is evaluated both ways and selected), and the outputs are routed by
// selects. Each lane performs exactly the operations of its own range: the values cannot differ from avp_sin /
// avp_cos (checked bit for bit on the host against glibc and on the device against the branchy forms).
namespace avp_trig {
AVP_HD double do_sin_sel(double x, double dx)
{
    const double ax = fabs(x);
    const double ty = taylor_sin(x * x, x, dx);                       // |x| < 0.126
    const double dxs = (x <= 0) ? -dx : dx;
    const double ux = big + ax;
    int k = (int)(avp_d2u(ux) & 0xffffffffu);
    k = k < 0 ? 0 : (k > 111 ? 111 : k);                              // (only lanes whose result is discarded can be out of range)
    const double xr = ax - (ux - big);
    const double xx = xr * xr;
    const double sp = xr + AVP_FMA(xr * xx, AVP_FMA(xx, sn5, sn3), dxs);
    const double cp = AVP_FMA(xr, dxs, xx * AVP_FMA(xx, AVP_FMA(xx, cs6, cs4), cs2));
    const double sn = AVP_SCT[k][0], ssn = AVP_SCT[k][1];
    const double cs = AVP_SCT[k][2], ccs = AVP_SCT[k][3];
    const double cor = AVP_FMA(cs, sp, AVP_FMA(-sn, cp, AVP_FMA(sp, ccs, ssn)));
    const double tb = copysign(sn + cor, x);
    return ax < 0.126 ? ty : tb;
}
AVP_HD double do_cos_sel(double x, double dx)
{
    const double dxs = (x < 0) ? -dx : dx;
    const double ux = big + fabs(x);
    int k = (int)(avp_d2u(ux) & 0xffffffffu);
    k = k < 0 ? 0 : (k > 111 ? 111 : k);
    const double xr = fabs(x) - (ux - big) + dxs;
    const double xx = xr * xr;
    const double sp = AVP_FMA(xr * xx, AVP_FMA(xx, sn5, sn3), xr);
    const double cp = xx * AVP_FMA(xx, AVP_FMA(xx, cs6, cs4), cs2);
    const double sn = AVP_SCT[k][0], ssn = AVP_SCT[k][1];
    const double cs = AVP_SCT[k][2], ccs = AVP_SCT[k][3];
    const double cor = AVP_FMA(-sn, sp, AVP_FMA(-cs, cp, AVP_FMA(-sp, ssn, ccs)));
    return cs + cor;
}
}  // namespace avp_trig

AVP_HD void avp_sincos(double x, double& sn, double& cs)
{
    using namespace avp_trig;
    const int32_t k = 0x7fffffff & (int32_t)(avp_d2u(x) >> 32);
    const double ax = fabs(x);
    // range 2 arguments
    const double t = hp0 - ax;
    const double a2 = t + hp1;
    const double da2 = (t - a2) + hp1;
    // range 3 arguments
    double a3, da3;
    const int n = reduce_sincos(x, &a3, &da3);
    const bool r1 = k < 0x3feb6000, r2 = k < 0x400368fd;             // (r2 is read only where !r1)
    const double su = r1 ? x : (r2 ? a2 : a3), sdu = r1 ? 0.0 : (r2 ? da2 : da3);
    const double cu = r1 ? x : (r2 ? t : a3), cdu = r1 ? 0.0 : (r2 ? hp1 : da3);
    const double S = do_sin_sel(su, sdu), C = do_cos_sel(cu, cdu);
    // range 3 routing: do_sincos(a, da, m) = (m & 1 ? C : S), negated when m & 2; sin uses m = n, cos m = n + 1
    const double rs = (n & 1) ? C : S;
    const double s3 = (n & 2) ? -rs : rs;
    const int m = n + 1;
    const double rc = (m & 1) ? C : S;
    const double c3 = (m & 2) ? -rc : rc;
    double so = r1 ? S : (r2 ? copysign(C, x) : s3);
    double co = r1 ? C : (r2 ? S : c3);
    if (k < 0x3e500000) so = x;                                        // tiny: sin x = x; cos x = 1 below 2^-27, do_cos(x, 0) = C above
    if (k < 0x3e400000) co = 1.0;
    if (!(k < 0x419921FB)) { so = NAN; co = NAN; }
    sn = so; cs = co;
}

// ---- Python float semantics ------------------------------------------------------------------
// CPython float %: result takes the sign of the divisor (Objects/floatobject.c float_rem)
AVP_HD double avp_pymod(double vx, double wx)
{
    double mod = fmod(vx, wx);
    if (mod != 0.0) { if ((wx < 0) != (mod < 0)) mod += wx; }
    else mod = copysign(0.0, wx);
    return mod;
}

// path_plan/rs_curve.py:649-656
AVP_HD double avp_pi_2_pi(double theta)
{
    while (theta > AVP_PI) theta -= 2.0 * AVP_PI;
    while (theta < -AVP_PI) theta += 2.0 * AVP_PI;
    return theta;
}

// theta % (2*pi) with CPython semantics. For |theta| < 4*pi the quotient is 0 or +-1 and theta -+ 2*pi
// is exact (Sterbenz), so fmod's exact remainder is obtained without the generic loop.
AVP_HD double avp_pymod_2pi(double theta)
{
    const double w = 2.0 * AVP_PI;
    const double a = fabs(theta);
    if (!(a < 2.0 * w)) return avp_pymod(theta, w);
    double mod = a < w ? theta : (theta < 0 ? theta + w : theta - w);     // == fmod(theta, w), sign of theta
    if (mod != 0.0) { if (mod < 0) mod += w; }
    else mod = 0.0;                                                         // copysign(0.0, w)
    return mod;
}

// path_plan/rs_curve.py:669-680
AVP_HD double avp_M(double theta)
{
    double phi = avp_pymod_2pi(theta);
    if (phi < -AVP_PI) phi += 2.0 * AVP_PI;
    if (phi > AVP_PI) phi -= 2.0 * AVP_PI;
    return phi;
}

// CPython 3.10 math.hypot(a, b) (Modules/mathmodule.c vector_norm, n = 2): scaled, split-accumulated
// sum of squares with one differential correction. Pure IEEE arithmetic, no libm dependence.
AVP_HD double avp_hypot(double a, double b)
{
    const double T27 = 134217729.0;
    double v0 = fabs(a), v1 = fabs(b);
    double mx = v0 > v1 ? v0 : v1;
    if (isinf(v0) || isinf(v1)) return INFINITY;
    if (v0 != v0 || v1 != v1) return NAN;
    if (mx == 0.0) return mx;
    int max_e;
    (void)frexp(mx, &max_e);
    if (max_e < -1023) return sqrt(a * a + b * b);
    const double scale = ldexp(1.0, -max_e);
    double csum = 1.0, frac1 = 0.0, frac2 = 0.0, frac3 = 0.0, x, t, hi, lo, oldcsum;
    for (int i = 0; i < 2; i++) {
        x = (i == 0 ? v0 : v1) * scale;
        t = x * T27; hi = t - (t - x); lo = x - hi;
        x = hi * hi; oldcsum = csum; csum += x; frac1 += (oldcsum - csum) + x;
        x = 2.0 * hi * lo; oldcsum = csum; csum += x; frac2 += (oldcsum - csum) + x;
        frac3 += lo * lo;
    }
    const double h = sqrt(csum - 1.0 + (frac1 + frac2 + frac3));
    x = h; t = x * T27; hi = t - (t - x); lo = x - hi;
    x = -hi * hi; oldcsum = csum; csum += x; frac1 += (oldcsum - csum) + x;
    x = -2.0 * hi * lo; oldcsum = csum; csum += x; frac2 += (oldcsum - csum) + x;
    x = -lo * lo; oldcsum = csum; csum += x; frac3 += (oldcsum - csum) + x;
    x = csum - 1.0 + (frac1 + frac2 + frac3);
    return (h + x / (2.0 * h)) / scale;
}

// ---- index searches on an ascending node table (map_position of map/costmap.py) ---------------------
// (PT: any pointer to const double -- plain, or LDS-qualified in the planner's called collision passes)
// First node index i with A[i] >= v (A ascending, n entries): exact against the table, any v.
template <class PT> AVP_HD int avp_first_ge(PT A, int n, double a0, double pitch, double v)
{
    double g = floor((v - a0) / pitch);
    int i = !(g >= 0.0) ? 0 : (g >= (double)n ? n : (int)g);
    while (i > 0 && A[i - 1] >= v) --i;
    while (i < n && A[i] < v) ++i;
    return i;
}
// Last node index i with A[i] <= v, or -1.
template <class PT> AVP_HD int avp_last_le(PT A, int n, double a0, double pitch, double v)
{
    double g = floor((v - a0) / pitch);
    int i = !(g >= 0.0) ? -1 : (g >= (double)n ? n - 1 : (int)g);
    while (i + 1 < n && A[i + 1] <= v) ++i;
    while (i >= 0 && A[i] > v) --i;
    return i;
}
// strict versions for the two-circle checker's exclusive filter (collision_check.py:119-127)
template <class PT> AVP_HD int avp_first_gt(PT A, int n, double a0, double pitch, double v)
{
    double g = floor((v - a0) / pitch);
    int i = !(g >= 0.0) ? 0 : (g >= (double)n ? n : (int)g);
    while (i > 0 && A[i - 1] > v) --i;
    while (i < n && !(A[i] > v)) ++i;
    return i;
}
template <class PT> AVP_HD int avp_last_lt(PT A, int n, double a0, double pitch, double v)
{
    double g = floor((v - a0) / pitch);
    int i = !(g >= 0.0) ? -1 : (g >= (double)n ? n - 1 : (int)g);
    while (i + 1 < n && A[i + 1] < v) ++i;
    while (i >= 0 && !(A[i] < v)) --i;
    return i;
}

// first_ge (upper = false) / last_le (upper = true) as ONE instruction stream, so that lanes searching different
// bounds of different axes do not diverge (same results as avp_first_ge / avp_last_le for finite v)
template <class PT> AVP_HD int avp_node_search(PT A, int n, double a0, double pitch, double v, bool upper)
{
    const double g = floor((v - a0) / pitch) + (upper ? 1.0 : 0.0);
    int c = !(g >= 0.0) ? 0 : (g >= (double)n ? n : (int)g);          // estimate of #{A[i] < v} resp. #{A[i] <= v}
    while (c > 0 && (upper ? A[c - 1] > v : A[c - 1] >= v)) --c;
    while (c < n && (upper ? A[c] <= v : A[c] < v)) ++c;
    return upper ? c - 1 : c;
}

// ---- numpy.linspace restated (obstacle-edge rasteriser, map/costmap.py:239) -------------------------
// numpy.linspace(0.0, stop, num)[q] (numpy/_core/function_base.py): step = stop / (num - 1);
// y = arange(num) * step (or (arange(num) / div) * stop when step == 0), y += 0.0, y[-1] = stop
AVP_HD double avp_linspace0(double stop, int num, int q)
{
    if (num > 1 && q == num - 1) return stop;
    const int div = num - 1;
    double y = (double)q;
    if (div > 0) {
        const double step = stop / (double)div;
        y = (step == 0.0) ? (y / (double)div) * stop : y * step;
    } else y = y * stop;
    return y + 0.0;
}

