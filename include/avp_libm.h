/*
 * avp_libm.h -- the scalar libm functions of the Reeds-Shepp words (plain C99, also valid C++/HIP).
 *
 * The reference's words (path_plan/rs_curve.py:159-534) call CPython math.atan2 / asin / acos / tan and float ** 2,
 * i.e. glibc 2.35 libm, and exact ties between mirror-image words (e.g. time-flipped LRL vs reflected RLR, whose
 * lengths both reduce to (-phi mod 2pi)) are decided by the last bit of those results. avp_atan2 / avp_asin /
 * avp_acos / avp_tan / avp_pow2 are therefore the bit-for-bit restatements of include/avp_glibc_libm.h (what the
 * x86-64 FMA build of that libm executes; tables in include/avp_glibc_tab.h). libavp_hip.so compiles them for the
 * device; the CPU oracle calls the platform libm by default and THIS header in its "restated libm" mode, and the two
 * modes agree on every fixture and workload (tests/test_glibc_libm.py, tests/test_oracle_restated.py).
 *
 * (Rounds 1-3 shipped a "portable", nearly correctly rounded atan2/asin/acos here and accepted the resulting tie
 * flips -- 0.01 % of queries, one golden trace and two workload problems; that implementation is gone.)
 *
 * avp_tan is glibc's for |x| <= 1e8 and for non-finite x. Beyond that glibc runs a Payne-Hanek reduction that is
 * not restated; avp_tan_fd (fdlibm-form kernel with a two-step Cody-Waite reduction, accurate to ~1e5 only) answers
 * there -- no caller on the path reaches it: the words call tan(phi) and tan(phi/2) with |phi| <= 2 pi.
 *
 * THIRD-PARTY NOTICE. avpm_ktan and avp_tan_fd below are derived from FreeBSD/Sun fdlibm (k_tan.c, s_tan.c),
 * whose notice is reproduced as required:
 *
 *   ====================================================
 *   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *   Copyright 2004 Sun Microsystems, Inc.  All Rights Reserved. (k_tan.c)
 *
 *   Developed at SunSoft, a Sun Microsystems, Inc. business.
 *   Permission to use, copy, modify, and distribute this
 *   software is freely granted, provided that this notice
 *   is preserved.
 *   ====================================================
 */
#ifndef AVP_LIBM_H
#define AVP_LIBM_H
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef AVP_LIBM_FN
#if defined(__HIPCC__)
#define AVP_LIBM_FN __host__ __device__ static inline
#else
#define AVP_LIBM_FN static inline
#endif
#endif
#include "avp_glibc_libm.h"

AVP_LIBM_FN uint32_t avpm_hi(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
AVP_LIBM_FN uint32_t avpm_lo(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
AVP_LIBM_FN double avpm_with_lo0(double x) { uint64_t u; memcpy(&u, &x, 8); u &= 0xffffffff00000000ull; memcpy(&x, &u, 8); return x; }

AVP_LIBM_FN double avp_atan2(double y, double x) { return avpg_atan2(y, x); }
AVP_LIBM_FN double avp_asin(double x) { return avpg_asin(x); }
AVP_LIBM_FN double avp_acos(double x) { return avpg_acos(x); }
/* libm pow(v, 2.0) -- CPython's v ** 2 -- which is not v*v */
AVP_LIBM_FN double avp_pow2(double v) { return avpg_pow2(v); }

/* ---- tan ------------------------------------------------------------------------------------- */
/* kernel on [-pi/4, pi/4] with tail y; iy = 1: tan, iy = -1: -1/tan */
AVP_LIBM_FN double avpm_ktan(double x, double y, int iy)
{
    const double T[13] = { 3.33333333333334091986e-01, 1.33333333333201242699e-01, 5.39682539762260521377e-02, 2.18694882948595424599e-02,
                           8.86323982359930005737e-03, 3.59207910759131235356e-03, 1.45620945432529025516e-03, 5.88041240820264096874e-04,
                           2.46463134818469906812e-04, 7.81794442939557092300e-05, 7.14072491382608190305e-05, -1.85586374855275456654e-05,
                           2.59073051863633712884e-05 };
    const double pio4 = 7.85398163397448278999e-01, pio4lo = 3.06161699786838301793e-17;
    double z, r, v, w, s;
    const uint32_t hx = avpm_hi(x);
    const uint32_t ix = hx & 0x7fffffffu;
    const int big = ix >= 0x3FE59428u;                             /* |x| >= 0.6744 */
    if (ix < 0x3e300000u) {                                        /* |x| < 2^-28 */
        if (iy == 1) return x;
        return -1.0 / x;
    }
    if (big) {
        if (hx >> 31) { x = -x; y = -y; }
        z = pio4 - x;
        w = pio4lo - y;
        x = z + w;
        y = 0.0;
    }
    z = x * x;
    w = z * z;
    r = T[1] + w * (T[3] + w * (T[5] + w * (T[7] + w * (T[9] + w * T[11]))));
    v = z * (T[2] + w * (T[4] + w * (T[6] + w * (T[8] + w * (T[10] + w * T[12])))));
    s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T[0] * s;
    w = x + r;
    if (big) {
        v = (double)iy;
        s = v - 2.0 * (x - (w * w / (w + v) - r));
        return (hx >> 31) ? -s : s;
    }
    if (iy == 1) return w;
    {
        /* -1/(x+r) with extra care */
        double a, t;
        z = avpm_with_lo0(w);
        v = r - (z - x);
        t = a = -1.0 / w;
        t = avpm_with_lo0(t);
        s = 1.0 + t * z;
        return t + a * (s + t * v);
    }
}

/* tan for |x| <= ~1e5 (the RS words use |x| < pi): two-step Cody-Waite reduction by pi/2 */
AVP_LIBM_FN double avp_tan_fd(double x)
{
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const uint32_t ix = avpm_hi(x) & 0x7fffffffu;
    double fn, r, w, t, y0, y1;
    int n;
    if (ix <= 0x3fe921fbu) return avpm_ktan(x, 0.0, 1);            /* |x| <= pi/4 */
    if (ix >= 0x7ff00000u) return x - x;
    fn = floor(fabs(x) * invpio2 + 0.5);
    n = (int)fn;
    r = fabs(x) - fn * pio2_1;
    w = fn * pio2_1t;
    y0 = r - w;
    /* second iteration when cancellation is large */
    {
        const int j = (int)(ix >> 20);
        const int i = j - (int)((avpm_hi(y0) >> 20) & 0x7ff);
        if (i > 16) {
            t = r;
            w = fn * pio2_2;
            r = t - w;
            w = fn * pio2_2t - ((t - r) - w);
            y0 = r - w;
        }
    }
    y1 = (r - y0) - w;
    if (avpm_hi(x) >> 31) { y0 = -y0; y1 = -y1; n = -n; }
    return avpm_ktan(y0, y1, 1 - ((n & 1) << 1));
}

AVP_LIBM_FN double avp_tan(double x)
{
    double r;
    if (avpg_tan_try(x, &r)) return r;
    return avp_tan_fd(x);                          /* |x| > 1e8: outside the restated range, see the header note */
}

#endif /* AVP_LIBM_H */
