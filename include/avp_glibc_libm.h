/*
 * avp_glibc_libm.h -- fp64 atan2 / asin / acos / tan / pow(., 2.0) with the results of glibc 2.35's x86-64 FMA
 * build, bit for bit (plain C99, also valid C++ / HIP device code).
 *
 * Why: the Reeds-Shepp words of the reference (path_plan/rs_curve.py:176,190,217-226,313,332,346,414,503,664)
 * call CPython math.atan2 / asin / acos / tan and float ** 2, i.e. the platform libm, and exact mathematical ties
 * between mirror-image words are decided by the last bit of those results. A device function that is merely
 * accurate (or even correctly rounded) picks the other word in 0.01 % of queries. So this header restates what
 * the libm the reference runs on EXECUTES: the selected IFUNC variants __ieee754_atan2_fma, __ieee754_asin_fma,
 * __ieee754_acos_fma, __tan_fma and __ieee754_pow_fma of Ubuntu GLIBC 2.35-0ubuntu3.x -- every operation in the
 * order of that build, every multiply-add the compiler contracted written as fma(), every separate multiply and
 * add kept separate (build with -ffp-contract=off). The polynomials' break-point tables are data of glibc
 * (include/avp_glibc_tab.h, see its notice). tests/test_glibc_libm.py compares the host build with the container's
 * libm (>= 1e9 arguments per function by scripts/glibc_libm_sweep.c), the -m gpu tests the device build with the
 * host build.
 *
 * Derived from the GNU C Library (sysdeps/ieee754/dbl-64: e_atan2.c, e_asin.c, s_tan.c, e_pow.c, e_exp.c --
 * IBM Accurate Mathematical Library and ARM optimized routines lineage), Copyright (C) Free Software
 * Foundation, Inc., distributed under the GNU Lesser General Public License, version 2.1 or (at your option)
 * any later version. This file is distributed under the same terms.
 *
 * Domain notes. Rounding mode: round-to-nearest is assumed (glibc switches to it on entry). errno and the
 * floating-point exception flags are not reproduced. Every finite and non-finite double is covered, tan's
 * Payne-Hanek range |x| > 1e8 (__branred, the generic SSE2 build: it has no FMA variant) included.
 */
#ifndef AVP_GLIBC_LIBM_H
#define AVP_GLIBC_LIBM_H
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef AVP_GLIBC_FN
#if defined(__HIPCC__)
#define AVP_GLIBC_FN __host__ __device__ static inline
#else
#define AVP_GLIBC_FN static inline
#endif
#endif
#ifndef AVP_GLIBC_TAB
#if defined(__HIP_DEVICE_COMPILE__)
#define AVP_GLIBC_TAB static __device__ const
#else
#define AVP_GLIBC_TAB static const
#endif
#endif
#include "avp_glibc_tab.h"

AVP_GLIBC_FN double avpg_from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
AVP_GLIBC_FN uint64_t avpg_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
AVP_GLIBC_FN int32_t avpg_hi(double d) { return (int32_t)(avpg_bits(d) >> 32); }
AVP_GLIBC_FN uint32_t avpg_lo(double d) { return (uint32_t)avpg_bits(d); }
AVP_GLIBC_FN double avpg_copysign(double mag, double sgn)
{
    return avpg_from_bits((avpg_bits(mag) & 0x7fffffffffffffffull) | (avpg_bits(sgn) & 0x8000000000000000ull));
}
#if defined(__HIP_DEVICE_COMPILE__)
#define AVPG_FMA(a, b, c) fma((a), (b), (c))
#else
#define AVPG_FMA(a, b, c) __builtin_fma((a), (b), (c))
#endif
#define AVPG_T(tab, i) avpg_from_bits(tab[(i)])
/* where the rows of uatan.tbl are read from: row i = 7 words {x_i, atan x_i, c1 .. c5}. A kernel that stages the table
 * elsewhere (LDS) defines AVPG_CIJ_ROW before including this header. */
#ifndef AVPG_CIJ_ROW
#define AVPG_CIJ_ROW(i) (AVP_G_CIJ + 7 * (i))
#endif

/* ================================================================================================ atan2 */
/* the odd Taylor polynomial of atan on |u| < 1/16, coefficients d3 .. d13 of e_atan2.c's atnat2.h */
AVP_GLIBC_FN double avpg_atan2_poly(double v)
{
    double p = 0x1.375f08b31cbcep-4;                 /* d13 */
    p = AVPG_FMA(p, v, -0x1.7458022b13c25p-4);       /* d11 */
    p = AVPG_FMA(p, v, 0x1.c71c6e5129a3bp-4);        /* d9  */
    p = AVPG_FMA(p, v, -0x1.24924923f7603p-3);       /* d7  */
    p = AVPG_FMA(p, v, 0x1.99999999997fdp-3);        /* d5  */
    p = AVPG_FMA(p, v, -0x1.5555555555555p-2);       /* d3  */
    return p;
}
/* row index of uatan.tbl for 1/16 <= u <= 1: round(256 u) - 16 */
AVP_GLIBC_FN int avpg_atan2_row(double u)
{
    const double two52 = 0x1p52;
    return (int)(AVPG_FMA(u, 256.0, two52) - two52) - 16;
}

/* the literal translation: every range, every special case; the slow path of avpg_atan2 below */
AVP_GLIBC_FN double avpg_atan2_ref(double y, double x)
{
    const double hpi = 0x1.921fb54442d18p+0, hpi1 = 0x1.1a62633145c07p-54;     /* pi/2 = hpi + hpi1 */
    const double opi = 0x1.921fb54442d18p+1, opi1 = 0x1.1a62633145c07p-53;     /* pi   = opi + opi1 */
    const double qpi = 0x1.921fb54442d18p-1, tqpi = 0x1.2d97c7f3321d2p+1;      /* pi/4, 3pi/4 */
    const double twom500 = 0x1p-500, two500 = 0x1p500, inv16 = 0.0625;
    const int32_t ux = avpg_hi(x), uy = avpg_hi(y);
    const uint32_t dx = avpg_lo(x), dy = avpg_lo(y);
    /* x or y NaN */
    if ((ux & 0x7ff00000) == 0x7ff00000 && (((ux & 0x000fffff) | (int32_t)dx) != 0)) return x + y;
    if ((uy & 0x7ff00000) == 0x7ff00000 && (((uy & 0x000fffff) | (int32_t)dy) != 0)) return y + y;
    /* y = +-0 */
    if (uy == 0x00000000) { if (dy == 0) return (ux & 0x80000000) == 0 ? 0.0 : opi; }
    else if ((uint32_t)uy == 0x80000000u) { if (dy == 0) return (ux & 0x80000000) == 0 ? -0.0 : -opi; }
    /* x = +-0 */
    if (x == 0.0) return (uy & 0x80000000) == 0 ? hpi : -hpi;
    /* x = +-inf */
    if (ux == 0x7ff00000) {
        if (dx == 0) {
            if (uy == 0x7ff00000) { if (dy == 0) return qpi; }
            else if ((uint32_t)uy == 0xfff00000u) { if (dy == 0) return -qpi; }
            else return (uy & 0x80000000) == 0 ? 0.0 : -0.0;
        }
    } else if ((uint32_t)ux == 0xfff00000u) {
        if (dx == 0) {
            if (uy == 0x7ff00000) { if (dy == 0) return tqpi; }
            else if ((uint32_t)uy == 0xfff00000u) { if (dy == 0) return -tqpi; }
            else return (uy & 0x80000000) == 0 ? opi : -opi;
        }
    }
    /* y = +-inf */
    if (uy == 0x7ff00000) { if (dy == 0) return hpi; }
    else if ((uint32_t)uy == 0xfff00000u) { if (dy == 0) return -hpi; }

    double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
    const int32_t de = (uy & 0x7ff00000) - (ux & 0x7ff00000);
    if (de >= 0x03900000) return y > 0.0 ? hpi : -hpi;                       /* |y/x| > 2^57 */
    if (de <= (int32_t)0xfc700000) {                                        /* |y/x| < 2^-57 */
        if (x > 0.0) return avpg_copysign(ay / ax, y);
        return y > 0.0 ? opi : -opi;
    }
    if (ax < twom500 || ay < twom500) { ax *= two500; ay *= two500; }
    if (ax > two500 || ay > two500) { ax *= twom500; ay *= twom500; }
    /* u + du = the smaller over the larger, to twice the working precision */
    double u, du, v, vv;
    if (ay < ax) { u = ay / ax; v = ax * u; vv = AVPG_FMA(ax, u, -v); du = ((ay - v) - vv) / ax; }
    else { u = ax / ay; v = ay * u; vv = AVPG_FMA(ay, u, -v); du = ((ax - v) - vv) / ay; }

    double z;
    if (x > 0.0) {
        if (ay < ax) {                                                      /* (i) atan(ay/ax) */
            if (u < inv16) {
                v = u * u;
                const double zz = AVPG_FMA(u * v, avpg_atan2_poly(v), du);
                z = u + zz;
                return avpg_copysign(z, y);
            }
            const uint64_t* c = AVPG_CIJ_ROW(avpg_atan2_row(u));
            const double t3 = u - AVPG_T(c, 0);
            double dv;
            v = t3 + du;
            if (fabs(t3) > fabs(du)) dv = (t3 - v) + du; else dv = (du - v) + t3;
            const double t2 = AVPG_T(c, 2);
            double p = AVPG_T(c, 6);
            p = AVPG_FMA(p, v, AVPG_T(c, 5));
            p = AVPG_FMA(p, v, AVPG_T(c, 4));
            p = AVPG_FMA(p, v, AVPG_T(c, 3));
            double zz = (v * v) * p;
            zz = AVPG_FMA(dv, t2, zz);
            zz = AVPG_FMA(v, t2, zz);
            z = zz + AVPG_T(c, 1);
            return avpg_copysign(z, y);
        }
        if (u < inv16) {                                                    /* (ii) pi/2 - atan(ax/ay) */
            v = u * u;
            const double zz = (u * v) * avpg_atan2_poly(v);
            const double t2 = hpi - u;
            double cor;
            if (hpi > fabs(u)) cor = (hpi - t2) - u; else cor = hpi - (u + t2);
            const double t3 = ((cor + hpi1) - du) - zz;
            z = t3 + t2;
            return avpg_copysign(z, y);
        }
        {
            const uint64_t* c = AVPG_CIJ_ROW(avpg_atan2_row(u));
            v = (u - AVPG_T(c, 0)) + du;
            double p = AVPG_T(c, 6);
            p = AVPG_FMA(p, v, AVPG_T(c, 5));
            p = AVPG_FMA(p, v, AVPG_T(c, 4));
            p = AVPG_FMA(p, v, AVPG_T(c, 3));
            p = AVPG_FMA(p, v, AVPG_T(c, 2));
            const double zz = AVPG_FMA(-p, v, hpi1);
            const double t1 = hpi - AVPG_T(c, 1);
            z = t1 + zz;
            return avpg_copysign(z, y);
        }
    }
    if (ay > ax) {                                                          /* (iii) pi/2 + atan(ax/ay) */
        if (u < inv16) {
            v = u * u;
            const double zz = (v * u) * avpg_atan2_poly(v);
            const double t2 = hpi + u;
            double cor;
            if (hpi > fabs(u)) cor = (hpi - t2) + u; else cor = (u - t2) + hpi;
            const double t3 = ((cor + hpi1) + du) + zz;
            z = t3 + t2;
            return avpg_copysign(z, y);
        }
        const uint64_t* c = AVPG_CIJ_ROW(avpg_atan2_row(u));
        v = (u - AVPG_T(c, 0)) + du;
        double p = AVPG_T(c, 6);
        p = AVPG_FMA(p, v, AVPG_T(c, 5));
        p = AVPG_FMA(p, v, AVPG_T(c, 4));
        p = AVPG_FMA(p, v, AVPG_T(c, 3));
        p = AVPG_FMA(p, v, AVPG_T(c, 2));
        const double zz = AVPG_FMA(p, v, hpi1);
        const double t1 = hpi + AVPG_T(c, 1);
        z = t1 + zz;
        return avpg_copysign(z, y);
    }
    if (u < inv16) {                                                        /* (iv) pi - atan(ay/ax) */
        v = u * u;
        const double zz = (v * u) * avpg_atan2_poly(v);
        const double t2 = opi - u;
        double cor;
        if (opi > fabs(u)) cor = (opi - t2) - u; else cor = opi - (t2 + u);
        const double t3 = ((cor + opi1) - du) - zz;
        z = t3 + t2;
        return avpg_copysign(z, y);
    }
    {
        const uint64_t* c = AVPG_CIJ_ROW(avpg_atan2_row(u));
        v = (u - AVPG_T(c, 0)) + du;
        double p = AVPG_T(c, 6);
        p = AVPG_FMA(p, v, AVPG_T(c, 5));
        p = AVPG_FMA(p, v, AVPG_T(c, 4));
        p = AVPG_FMA(p, v, AVPG_T(c, 3));
        p = AVPG_FMA(p, v, AVPG_T(c, 2));
        const double zz = AVPG_FMA(-p, v, opi1);
        const double t1 = opi - AVPG_T(c, 1);
        z = t1 + zz;
        return avpg_copysign(z, y);
    }
}

/* atan2 for wave-wide evaluation. The translation above is a tree of eight leaves -- (i) .. (iv) by the signs and the
 * larger of |x|, |y|, each with a polynomial (u < 1/16) and a table branch -- and a wave whose lanes hold directions all
 * around the circle (the Reeds-Shepp words: one word, 64 queries) executes all eight one after the other. Here the eight
 * are ONE instruction stream with per-lane selects: (ii), (iii), (iv) are the same expressions up to the constants
 * (pi/2 or pi, their tails) and exact sign flips (a - b == a + (-b), fma(-p, v, c) == fma(p, -v, c) bit for bit), (i)
 * shares the division, the table row and the first three Horner steps; both the polynomial and the table form are
 * evaluated and the lane's one is kept. Same operations on the same operands in the same order for every lane, so the
 * results are those of avpg_atan2_ref bit for bit (scripts/glibc_libm_sweep.c compares both with libm). Zeros,
 * infinities, NaNs, exponents that need glibc's rescaling and ratios beyond 2^+-57 take avpg_atan2_ref. */
AVP_GLIBC_FN double avpg_atan2(double y, double x)
{
    const double hpi = 0x1.921fb54442d18p+0, hpi1 = 0x1.1a62633145c07p-54;
    const double opi = 0x1.921fb54442d18p+1, opi1 = 0x1.1a62633145c07p-53;
    const int ex = (avpg_hi(x) >> 20) & 0x7ff, ey = (avpg_hi(y) >> 20) & 0x7ff;
    const int dxy = ey - ex;
    if (!(ex >= 524 && ex <= 1522 && ey >= 524 && ey <= 1522 && dxy >= -56 && dxy <= 56)) return avpg_atan2_ref(y, x);
    const double ax = fabs(x), ay = fabs(y);
    const int swap = !(ay < ax);                       /* u = ax / ay */
    const double num = swap ? ax : ay, den = swap ? ay : ax;
    const double u = num / den;
    /* table row (garbage but in range for u < 1/16, where it is not used): issued before the second division */
    int row = avpg_atan2_row(u);
    row = row < 0 ? 0 : row;
    const uint64_t* c = AVPG_CIJ_ROW(row);
    const double c0 = AVPG_T(c, 0), c1 = AVPG_T(c, 1), c2 = AVPG_T(c, 2), c3 = AVPG_T(c, 3), c4 = AVPG_T(c, 4), c5 = AVPG_T(c, 5), c6 = AVPG_T(c, 6);
    const double vq = den * u;
    const double vv = AVPG_FMA(den, u, -vq);
    const double du = ((num - vq) - vv) / den;
    const int xpos = x > 0.0;
    const int case1 = xpos && !swap;                   /* (i)   atan(ay/ax)          */
    const int case3 = !xpos && ay > ax;                /* (iii) pi/2 + atan(ax/ay)   */
    const int case4 = !xpos && !(ay > ax);             /* (iv)  pi - atan(ay/ax); (ii) pi/2 - atan(ax/ay) is the rest */
    const double H = case4 ? opi : hpi, H1 = case4 ? opi1 : hpi1;
    const int neg = !case3;                            /* (ii), (iv): the atan term is subtracted */
    /* polynomial form, u < 1/16 */
    const double v2 = u * u;
    const double P = avpg_atan2_poly(v2);
    const double uv = u * v2;
    const double z1s = u + AVPG_FMA(uv, P, du);                                   /* (i) */
    const double su = neg ? -u : u, sdu = neg ? -du : du;
    const double zzs = uv * P;
    const double t2s = H + su;
    const double cors = (H - t2s) + su;
    const double t3s = ((cors + H1) + sdu) + (neg ? -zzs : zzs);
    const double zos = t3s + t2s;                                                 /* (ii) .. (iv) */
    /* table form, u >= 1/16 */
    const double t3 = u - c0;
    const double v = t3 + du;
    const double dv = fabs(t3) > fabs(du) ? (t3 - v) + du : (du - v) + t3;
    double p = AVPG_FMA(c6, v, c5);
    p = AVPG_FMA(p, v, c4);
    p = AVPG_FMA(p, v, c3);
    double zz1 = (v * v) * p;
    zz1 = AVPG_FMA(dv, c2, zz1);
    zz1 = AVPG_FMA(v, c2, zz1);
    const double z1t = zz1 + c1;                                                  /* (i) */
    p = AVPG_FMA(p, v, c2);
    const double zzt = AVPG_FMA(neg ? -p : p, v, H1);
    const double zot = (H + (neg ? -c1 : c1)) + zzt;                              /* (ii) .. (iv) */
    const int small = u < 0.0625;
    const double z = case1 ? (small ? z1s : z1t) : (small ? zos : zot);
    return avpg_copysign(z, y);
}

/* ========================================================================================== asin / acos */
/* f6 .. f1 of e_asin.c: asin(x) - x on |x| < 1/8, and the tail of the sqrt branch */
AVP_GLIBC_FN double avpg_asin_fpoly(double x2)
{
    double p = 0x1.292d80f453c72p-6;                 /* f6 */
    p = AVPG_FMA(p, x2, 0x1.6e442c822d419p-6);       /* f5 */
    p = AVPG_FMA(p, x2, 0x1.f1c7e04f4ad99p-6);       /* f4 */
    p = AVPG_FMA(p, x2, 0x1.6db6dae42c0e4p-5);       /* f3 */
    p = AVPG_FMA(p, x2, 0x1.333333336127dp-4);       /* f2 */
    p = AVPG_FMA(p, x2, 0x1.55555555554f9p-3);       /* f1 */
    return p;
}
/* which block of asincos.tbl serves |x| in [1/8, 31/32): row offset n and the number of coefficients
 * c2 .. c_deg that multiply powers of xx (the row is {x_n, c1, c2 .. c_deg, c_sq, asin(x_n)}) */
AVP_GLIBC_FN int avpg_asin_row(int32_t k, int* deg)
{
    if (k < 0x3fe00000) {                            /* [0.125, 0.5) */
        *deg = 6;
        return k < 0x3fd00000 ? 11 * ((k >> 15) & 0x1f) : 11 * ((k >> 14) & 0x3f) + 352;
    }
    if (k < 0x3fe80000) { *deg = 7; return 1056 + 3 * ((k >> 11) & 0x1fc); }      /* [0.5, 0.75) */
    if (k < 0x3fed8000) { *deg = 8; return 992 + 13 * ((k >> 13) & 0x7f); }       /* [0.75, 0.921875) */
    if (k < 0x3fee8000) { *deg = 9; return 884 + 14 * ((k >> 13) & 0x7f); }       /* [0.921875, 0.953125) */
    *deg = 10;
    return 768 + 15 * ((k >> 13) & 0x7f);                                         /* [0.953125, 0.96875) */
}
/* the table polynomial: t = c1 xx + (c2 + c3 xx + .. + c_deg xx^(deg-2)) xx^2 + c_sq, Horner with fma, the last step on
 * xx^2, c1 last. Returns t; *tail = asin(x_n). The chain has the SAME length for every row -- steps above the row's degree
 * run on a zero coefficient, and fma(0, xx, 0) = 0, fma(0, xx, c) = c exactly, so the value is glibc's -- which lets the
 * lanes of a wave (rows of different degree) share one instruction stream whose 13 loads are all in flight at once. */
AVP_GLIBC_FN double avpg_asin_tabpoly(double ax_signed, int n, int deg, double* tail)
{
    const uint64_t* a = AVP_G_ASNCS + n;
    const double xx = ax_signed - AVPG_T(a, 0);
    const double k1 = AVPG_T(a, 1), csq = AVPG_T(a, deg + 1), tl = AVPG_T(a, deg + 2);
    double cf[11];
    for (int j = 2; j <= 10; ++j) cf[j] = AVPG_T(a, j);                     /* (reads within the table for every row) */
    double p = 0.0;
    for (int j = 10; j >= 2; --j) p = AVPG_FMA(p, xx, j <= deg ? cf[j] : 0.0);
    p = AVPG_FMA(p, xx * xx, csq);
    *tail = tl;
    return AVPG_FMA(xx, k1, p);
}
/* sqrt branch shared by asin and acos for 31/32 <= |x| < 1: z = (1 - |x|)/2; c ~ sqrt(z) by the inroot /
 * powtwo seed and one polynomial + one Newton step; returns through pointers */
AVP_GLIBC_FN void avpg_asin_root(double z, double* c_out, double* tc_out)
{
    const int32_t k = avpg_hi(z);
    double t = AVPG_T(AVP_G_INROOT, (k >> 14) & 0x7f) * AVPG_T(AVP_G_POWTWO, 511 - (k >> 21));
    const double r = AVPG_FMA(-(t * t), z, 1.0);
    double q = 0x1.4006318d1dab9p-2;                 /* rt3 */
    q = AVPG_FMA(q, r, 0x1.800496769c91ap-2);        /* rt2 */
    q = AVPG_FMA(q, r, 0x1.fffffff757304p-2);        /* rt1 */
    q = AVPG_FMA(q, r, 0x1.fffffffecc1ddp-1);        /* rt0 */
    t = q * t;
    const double c = z * t;
    *c_out = c;
    *tc_out = AVPG_FMA(-(t * 0.5), c, 1.5);          /* 1.5 - 0.5 t c */
}

AVP_GLIBC_FN double avpg_asin(double x)
{
    const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
    const int32_t m = avpg_hi(x);
    const int32_t k = m & 0x7fffffff;
    if (k < 0x3e500000) return x;                                           /* |x| < 2^-26 */
    if (k < 0x3fc00000) {                                                   /* |x| < 1/8 */
        const double x2 = x * x;
        return AVPG_FMA(avpg_asin_fpoly(x2), x * x2, x);
    }
    if (k < 0x3fef0000) {                                                   /* table ranges */
        int deg;
        const int n = avpg_asin_row(k, &deg);
        double tail;
        const double t = avpg_asin_tabpoly(m > 0 ? x : -x, n, deg, &tail);
        const double res = t + tail;
        return m > 0 ? res : -res;
    }
    if (k < 0x3ff00000) {                                                   /* 31/32 <= |x| < 1 */
        const double z = (m > 0 ? 1.0 - x : x + 1.0) * 0.5;
        double c, tc;
        avpg_asin_root(z, &c, &tc);
        const double t24 = 0x1p24;
        const double y = (c + t24) - t24;
        const double ty = AVPG_FMA(tc, c, y);                               /* c (1.5 - 0.5 t c) + y */
        const double cc = AVPG_FMA(-y, y, z) / ty;
        const double p = avpg_asin_fpoly(z) * z;
        const double a = AVPG_FMA(-cc, 2.0, hp1);                           /* hp1 - 2 cc */
        const double res1 = AVPG_FMA(-y, 2.0, hp0);                         /* hp0 - 2 y  */
        const double s = (y + cc) + (y + cc);
        const double cor = AVPG_FMA(-s, p, a);
        const double res = cor + res1;
        return m > 0 ? res : -res;
    }
    if (k == 0x3ff00000 && avpg_lo(x) == 0) return m > 0 ? hp0 : -hp0;      /* |x| == 1 */
    if (k > 0x7ff00000 || (k == 0x7ff00000 && avpg_lo(x) != 0)) return x + x;
    return (x - x) / (x - x);                                               /* |x| > 1: NaN */
}

AVP_GLIBC_FN double avpg_acos(double x)
{
    const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54, pi_hi = 0x1.921fb54442d18p+1;
    const int32_t m = avpg_hi(x);
    const int32_t k = m & 0x7fffffff;
    if (k < 0x3c880000) return hp0;                                         /* |x| < 2^-55 */
    if (k < 0x3fc00000) {                                                   /* |x| < 1/8 */
        const double x2 = x * x;
        const double p = avpg_asin_fpoly(x2);
        const double t = hp0 - x;
        const double cor = ((hp0 - t) - x) + hp1;
        return t + AVPG_FMA(-p, x * x2, cor);
    }
    if (k < 0x3fef0000) {
        int deg;
        const int n = avpg_asin_row(k, &deg);
        double tail;
        const double t = avpg_asin_tabpoly(m > 0 ? x : -x, n, deg, &tail);
        if (m > 0) return (hp1 - t) + (hp0 - tail);
        return (t + hp1) + (tail + hp0);
    }
    if (k < 0x3ff00000) {
        const double z = (m > 0 ? 1.0 - x : x + 1.0) * 0.5;
        double c, tc;
        avpg_asin_root(z, &c, &tc);
        const double t27 = 0x1p27;
        const double y = AVPG_FMA(-t27, c, AVPG_FMA(c, t27, c));            /* (t27 c + c) - t27 c */
        const double ty = AVPG_FMA(tc, c, y);
        const double cc = AVPG_FMA(-y, y, z) / ty;
        const double p = (avpg_asin_fpoly(z) * z) * (y + cc);
        if (m >= 0) { const double r = (cc + p) + y; return r + r; }
        { const double r = ((hp1 - cc) - p) + (hp0 - y); return r + r; }
    }
    if (k == 0x3ff00000 && avpg_lo(x) == 0) return m > 0 ? 0.0 : pi_hi;
    if (k > 0x7ff00000 || (k == 0x7ff00000 && avpg_lo(x) != 0)) return x + x;
    return (x - x) / (x - x);
}

/* ================================================================================================== tan */
AVP_GLIBC_FN double avpg_tan_poly(double a2)
{
    double t = 0x1.2385a3cf2e4eap-7;                 /* d11 */
    t = AVPG_FMA(t, a2, 0x1.664ed49cfc666p-6);       /* d9  */
    t = AVPG_FMA(t, a2, 0x1.ba1ba1cdb8745p-5);       /* d7  */
    t = AVPG_FMA(t, a2, 0x1.11111111107c6p-3);       /* d5  */
    t = AVPG_FMA(t, a2, 0x1.5555555555555p-2);       /* d3  */
    return t;
}
/* -1 / (b + db) as glibc's DIV2(1, 0, b, db) followed by y = c + dc */
AVP_GLIBC_FN double avpg_tan_mcot(double a, double t2)
{
    const double b = a + t2;
    double db;
    if (fabs(a) > fabs(t2)) db = (a - b) + t2; else db = (t2 - b) + a;
    const double c = 1.0 / b;
    const double u = c * b;
    const double uu = AVPG_FMA(c, b, -u);
    double w = ((1.0 - u) - uu) + 0.0;
    w = AVPG_FMA(-db, c, w);
    const double dc = w / b;
    const double s = c + dc;
    return -(((c - s) + dc) + s);
}
/* utan.tbl branch: ya + yya in (0.0608, pi/4], n = parity of the quadrant, sy = sign. tan: fi + num/(gi - s), -cot:
 * gi - num/(s + fi) -- one division on selected operands, so that lanes in even and odd quadrants share the stream. */
AVP_GLIBC_FN double avpg_tan_table(double ya, double yya, int n, double sy)
{
    const int i = (int)AVPG_FMA(256.0, ya, -15.5);
    const uint64_t* row = AVP_G_XFG + 4 * i;
    const double x0 = AVPG_T(row, 0), fi = AVPG_T(row, 1), gi = AVPG_T(row, 2);
    const double z = (ya - x0) + yya;
    const double z2 = z * z;
    const double s = AVPG_FMA(z * z2, AVPG_FMA(z2, 0x1.11112e0a6b45fp-3 /* e1 */, 0x1.5555555554dbdp-2 /* e0 */), z);
    const double num = (fi + gi) * s;
    const double q = num / (n ? s + fi : gi - s);
    return n ? (gi - q) * -sy : (q + fi) * sy;
}
/* common tail of the reduced ranges: a + da = x - n pi/2 */
AVP_GLIBC_FN double avpg_tan_reduced(double a, double da, int n)
{
    const double g2 = 0x1.f212dp-5;
    double ya = a, yya = da, sy = 1.0;
    if (a < 0.0) { ya = -a; yya = -da; sy = -1.0; }
    if (ya <= g2) {
        const double a2 = a * a;
        const double t2 = AVPG_FMA(a * a2, avpg_tan_poly(a2), da);
        if (n == 0) return a + t2;
        return avpg_tan_mcot(a, t2);
    }
    return avpg_tan_table(ya, yya, n, sy);
}

/* __branred (branred.c): x - n pi/2 for |x| > 1e8 as a + aa, n mod 4 returned. x is scaled by 2^-600, split in two
 * 27-bit halves, each multiplied by six 24-bit pieces of 2/pi picked by its exponent; the integer parts are peeled off
 * with the 1.5 * 2^52 trick. The library's build of this file is plain SSE2: no contraction anywhere. */
AVP_GLIBC_FN void avpg_branred_half(double xh, double* b_out, double* bb_out, double* sum_out)
{
    const double tm24 = 0x1p-24, big = 0x1.8p52, big1 = 0x1.8p54;
    int k = (int)((avpg_bits(xh) >> 52) & 2047);
    k = (k - 450) / 24;
    if (k < 0) k = 0;
    double gor = avpg_from_bits(((uint64_t)(0x63f00000u - (uint32_t)((k * 24) << 20))) << 32);      /* 2^(576 - 24 k) */
    double r[6];
    for (int i = 0; i < 6; i++) { r[i] = xh * AVPG_T(AVP_G_TOVERP, k + i) * gor; gor *= tm24; }
    double sum = 0.0, s;
    for (int i = 0; i < 3; i++) { s = (r[i] + big) - big; sum += s; r[i] -= s; }
    double t = 0.0;
    for (int i = 0; i < 6; i++) t += r[5 - i];
    double bb = (((((r[0] - t) + r[1]) + r[2]) + r[3]) + r[4]) + r[5];
    s = (t + big) - big;
    sum += s;
    t -= s;
    const double b = t + bb;
    bb = (t - b) + bb;
    s = (sum + big1) - big1;
    sum -= s;
    *b_out = b; *bb_out = bb; *sum_out = sum;
}
AVP_GLIBC_FN int avpg_branred(double x, double* a, double* aa)
{
    const double split = 134217729.0, hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
    const double mp1 = 0x1.921fb58p+0, mp2 = -0x1.dde974p-27;                  /* branred.h's split of pi/2 (not s_tan.c's) */
    x *= 0x1p-600;
    double t = x * split;
    const double x1 = t - (t - x), x2 = x - x1;
    double b1, bb1, sum1, b2, bb2, sum2;
    avpg_branred_half(x1, &b1, &bb1, &sum1);
    avpg_branred_half(x2, &b2, &bb2, &sum2);
    double sum = sum1 + sum2;
    double b = b1 + b2;
    double bb = fabs(b1) > fabs(b2) ? (b1 - b) + b2 : (b2 - b) + b1;
    if (b > 0.5) { b -= 1.0; sum += 1.0; }
    else if (b < -0.5) { b += 1.0; sum -= 1.0; }
    double s = b + (bb + bb1 + bb2);
    t = ((b - s) + bb) + (bb1 + bb2);
    b = s * split;
    const double t1 = b - (b - s), t2 = s - t1;
    b = s * hp0;
    bb = (((t1 * mp1 - b) + t1 * mp2) + t2 * mp1) + (t2 * mp2 + s * hp1 + t * hp0);
    s = b + bb;
    t = (b - s) + bb;
    *a = s; *aa = t;
    return ((int)sum) & 3;
}

/* tan(x); always returns 1 (kept as a "try" for the callers written when the Payne-Hanek range was not restated) */
AVP_GLIBC_FN int avpg_tan_try(double x, double* res)
{
    const double g1 = 0x1.b096cp-27, g2 = 0x1.f212dp-5, g3 = 0x1.92f1ap-1, g4 = 25.0, g5 = 1e8;
    const double hpinv = 0x1.45f306dc9c883p-1, toint = 0x1.8p52;
    const double mp1 = 0x1.921fb58p+0, mp2 = -0x1.dde973cp-27, mp3 = -0x1.cb3b399d747f2p-55;
    const double pp3 = -0x1.cb3b398p-55, pp4 = -0x1.d747f23e32ed7p-83;
    if ((avpg_hi(x) & 0x7ff00000) == 0x7ff00000) { *res = x - x; return 1; }
    const double w = x < 0.0 ? -x : x;
    if (w <= g1) { *res = x; return 1; }
    if (w <= g2) {
        const double x2 = x * x;
        *res = AVPG_FMA(x * x2, avpg_tan_poly(x2), x);
        return 1;
    }
    if (w <= g3) {                                  /* (w - x_i) + 0.0 == w - x_i bit for bit: a difference is never -0.0 */
        *res = avpg_tan_table(w, 0.0, 0, x < 0.0 ? -1.0 : 1.0);
        return 1;
    }
    if (w <= g4) {
        const double t = AVPG_FMA(x, hpinv, toint);
        const double xn = t - toint;
        const int n = (int)(avpg_bits(t) & 1u);
        double t1 = AVPG_FMA(-xn, mp1, x);
        t1 = AVPG_FMA(-xn, mp2, t1);
        const double a = AVPG_FMA(-xn, mp3, t1);
        const double da = AVPG_FMA(-xn, mp3, t1 - a);
        *res = avpg_tan_reduced(a, da, n);
        return 1;
    }
    if (w <= g5) {
        const double t = AVPG_FMA(x, hpinv, toint);
        const double xn = t - toint;
        const int n = (int)(avpg_bits(t) & 1u);
        double t1 = AVPG_FMA(-xn, mp1, x);
        t1 = AVPG_FMA(-xn, mp2, t1);
        const double a1 = AVPG_FMA(-xn, pp3, t1);
        double da = AVPG_FMA(-xn, pp3, t1 - a1);
        double a = AVPG_FMA(-xn, pp4, a1);
        da = da + AVPG_FMA(-xn, pp4, a1 - a);
        const double s = a + da;
        double ds;
        if (fabs(a) > fabs(da)) ds = (a - s) + da; else ds = (da - s) + a;
        *res = avpg_tan_reduced(s, ds, n);
        return 1;
    }
    {                                               /* |x| > 1e8: __branred, then the same tail */
        double a, da;
        const int n = avpg_branred(x, &a, &da) & 1;
        const double s = a + da;
        double ds;
        if (fabs(a) > fabs(da)) ds = (a - s) + da; else ds = (da - s) + a;
        *res = avpg_tan_reduced(s, ds, n);
        return 1;
    }
}

/* ========================================================================================== pow(x, 2.0) */
/* CPython's float ** 2 is libm pow(x, 2.0), which is NOT x*x (glibc evaluates exp(2 log x) to ~2^-68 and rounds:
 * 0.08 % of arguments differ from the correctly rounded square). __ieee754_pow_fma specialised to y = 2. */
AVP_GLIBC_FN double avpg_pow2(double x)
{
    uint64_t ix = avpg_bits(x);
    uint32_t topx = (uint32_t)(ix >> 52);
    if (topx - 1u >= 0x7fdu) {
        if (2 * ix - 1 >= 2 * 0x7ff0000000000000ull - 1) return x * x;      /* +-0, +-inf, NaN */
        if (ix >> 63) { ix &= 0x7fffffffffffffffull; topx &= 0x7ff; }      /* y = 2 is an even integer */
        if (topx == 0) {                                                    /* subnormal */
            ix = avpg_bits(x * 0x1p52) & 0x7fffffffffffffffull;
            ix -= 52ull << 52;
        }
    }
    /* log_inline */
    const uint64_t tmp = ix - 0x3fe6955500000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    const double z = avpg_from_bits(iz), kd = (double)k;
    const uint64_t* T = AVP_G_POWLOG_TAB + 4 * i;
    const double invc = AVPG_T(T, 0), logc = AVPG_T(T, 2), logctail = AVPG_T(T, 3);
    /* __pow_log_data: ln2hi, ln2lo and the polynomial A[0..6] (literals: the device keeps them in registers / as
     * instruction constants instead of loading them; include/gen_glibc_tab.py checks them against the archive) */
    const double ln2hi = 0x1.62e42fefa3800p-1, ln2lo = 0x1.ef35793c76730p-45;
    const double A0 = -0x1p-1, A1 = -0x1.555555555556p-1, A2 = 0x1.0000000000006p-1, A3 = 0x1.999999959554ep-1,
                 A4 = -0x1.555555529a47ap-1, A5 = -0x1.2495b9b4845e9p+0, A6 = 0x1.0002b8b263fc3p+0;
    const double r = AVPG_FMA(z, invc, -1.0);
    const double t1 = AVPG_FMA(kd, ln2hi, logc);
    const double t2 = t1 + r;
    const double lo1 = AVPG_FMA(kd, ln2lo, logctail);
    const double lo2 = (t1 - t2) + r;
    const double ar = A0 * r;
    const double ar2 = r * ar;
    const double ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = AVPG_FMA(ar, r, -ar2);
    const double lo4 = (t2 - hi) + ar2;
    double q = AVPG_FMA(r, A6, A5);
    q = AVPG_FMA(q, ar2, AVPG_FMA(A4, r, A3));
    q = AVPG_FMA(ar2, q, AVPG_FMA(A2, r, A1));
    const double lo = AVPG_FMA(ar3, q, ((lo1 + lo2) + lo3) + lo4);
    const double lhi = hi + lo;
    const double llo = (hi - lhi) + lo;
    /* y * log x, y = 2 */
    const double ehi = 2.0 * lhi;
    const double elo = AVPG_FMA(2.0, llo, AVPG_FMA(lhi, 2.0, -ehi));
    /* exp_inline(ehi, elo, 0) */
    uint32_t abstop = (uint32_t)(avpg_bits(ehi) >> 52) & 0x7ff;
    if (abstop - 0x3c9u >= 0x3fu) {
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + ehi;               /* |2 log x| < 2^-54 */
        if (abstop >= 0x409u) return (avpg_bits(ehi) >> 63) ? 0x1p-767 * 0x1p-767 : 0x1p769 * 0x1p769;
        abstop = 0;
    }
    /* __exp_data: InvLn2N, Shift, NegLn2hiN, NegLn2loN, C2 .. C5 */
    const double invln2N = 0x1.71547652b82fep+7, shift = 0x1.8p+52;
    const double negln2hiN = -0x1.62e42fefa0000p-8, negln2loN = -0x1.cf79abc9e3b3ap-47;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    double kd2 = AVPG_FMA(ehi, invln2N, shift);
    const uint64_t ki = avpg_bits(kd2);
    kd2 -= shift;
    double rr = AVPG_FMA(kd2, negln2hiN, ehi);
    rr = AVPG_FMA(kd2, negln2loN, rr);
    rr = elo + rr;
    const unsigned idx = 2 * (unsigned)(ki % 128);
    uint64_t sbits = AVP_G_EXP_TAB[idx + 1] + (ki << 45);
    const double tail = AVPG_T(AVP_G_EXP_TAB, idx);
    const double r2 = rr * rr;
    double e = AVPG_FMA(C3, rr, C2);
    e = AVPG_FMA(e, r2, rr + tail);
    e = AVPG_FMA(AVPG_FMA(rr, C5, C4), r2 * r2, e);
    if (abstop == 0) {                                                      /* specialcase(): |2 log x| >= 512 */
        if ((ki & 0x80000000u) == 0) {
            sbits -= 1009ull << 52;
            const double scale = avpg_from_bits(sbits);
            return AVPG_FMA(scale, e, scale) * 0x1p1009;
        }
        sbits += 1022ull << 52;
        const double scale = avpg_from_bits(sbits);
        const double se = e * scale;
        double y = scale + se;
        if (fabs(y) < 1.0) {
            const double one = y < 0.0 ? -1.0 : 1.0;
            double l = (scale - y) + se;
            const double h = y + one;
            l = ((one - h) + y) + l;
            y = (l + h) - one;
            if (y == 0.0) y = avpg_from_bits(sbits & 0x8000000000000000ull);
        }
        return y * 0x1p-1022;
    }
    const double scale = avpg_from_bits(sbits);
    return AVPG_FMA(e, scale, scale);
}

#endif /* AVP_GLIBC_LIBM_H */
