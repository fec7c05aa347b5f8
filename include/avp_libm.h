/*
 * avp_libm.h -- portable IEEE fp64 atan / atan2 / asin / acos / tan (plain C99, also valid C++/HIP).
 *
 * Why this exists: the Reeds-Shepp words of the reference (path_plan/rs_curve.py:159-534) call
 * CPython math.atan2/asin/acos/tan, i.e. glibc 2.35 libm, whose table-driven IBM-lineage kernels are
 * neither correctly rounded nor reproducible from first principles; the ROCm device libm differs
 * from them by <= 1-2 ulp. Exact ties between mirror-image words (e.g. time-flipped LRL vs reflected
 * RLR, whose lengths both reduce to (-phi mod 2pi)) are broken by those last bits, so no device
 * implementation can follow the reference through every tie. To keep the device path verifiable bit
 * for bit, the RS words use THIS implementation on every backend: libavp_hip.so (device) and the
 * CPU oracle's "portable" mode compile the same functions (no FMA contraction; every operation is
 * written out), so GPU == CPU-port exactly, and the port-vs-glibc difference is confined to the
 * last ulp of these four functions (tests/test_oracle_portable.py quantifies it).
 *
 * Algorithms: atan2/asin/acos go through one double-double atan core (65-entry table of atan(k/64),
 * double-double argument reduction, one final rounding) and are correctly rounded except within
 * ~2^-15 ulp of a rounding boundary; they differ from glibc essentially only where glibc itself is
 * not correctly rounded (~0.05-0.2 % of calls). tan is the fdlibm kernel (< 1 ulp; used by the two SLS
 * words only). Special cases use the classic fdlibm plumbing. Checked in tests/test_math_host.py.
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

#ifndef AVP_LIBM_TAB
#if defined(__HIP_DEVICE_COMPILE__)
#define AVP_LIBM_TAB static __device__ const
#else
#define AVP_LIBM_TAB static const
#endif
#endif
#include "avp_atan_tab.h"

/* the device build redirects the lookups to an LDS copy of the table (csrc/avp_device.h) */
#if defined(__HIP_DEVICE_COMPILE__)
__shared__ double AVP_ATAN_LDS[65][2];
#define AVP_ATAN_TAB_REF AVP_ATAN_LDS
#else
#define AVP_ATAN_TAB_REF AVP_ATAN_TAB
#endif

AVP_LIBM_FN uint32_t avpm_hi(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
AVP_LIBM_FN uint32_t avpm_lo(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
AVP_LIBM_FN double avpm_with_lo0(double x) { uint64_t u; memcpy(&u, &x, 8); u &= 0xffffffff00000000ull; memcpy(&x, &u, 8); return x; }

/* ---- atan ------------------------------------------------------------------------------------ */
AVP_LIBM_FN double avp_atan(double x)
{
    const double atanhi[4] = { 4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00 };
    const double atanlo[4] = { 2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17 };
    const double aT[11] = { 3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
                            9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
                            4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02 };
    double w, s1, s2, z;
    int id;
    const uint32_t hx = avpm_hi(x);
    const uint32_t ix = hx & 0x7fffffffu;
    const int neg = (hx >> 31) != 0;
    if (ix >= 0x44100000u) {                 /* |x| >= 2^66 */
        if (x != x) return x + x;
        return neg ? -(atanhi[3] + atanlo[3]) : (atanhi[3] + atanlo[3]);
    }
    if (ix < 0x3fdc0000u) {                  /* |x| < 0.4375 */
        if (ix < 0x3e200000u) return x;      /* |x| < 2^-29 */
        id = -1;
    } else {
        x = fabs(x);
        if (ix < 0x3ff30000u) {              /* |x| < 1.1875 */
            if (ix < 0x3fe60000u) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); }      /* 7/16 <= |x| < 11/16 */
            else { id = 1; x = (x - 1.0) / (x + 1.0); }                             /* 11/16 <= |x| < 19/16 */
        } else {
            if (ix < 0x40038000u) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); }      /* |x| < 2.4375 */
            else { id = 3; x = -1.0 / x; }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return neg ? -z : z;
}

/* ---- nearly correctly rounded atan2 / asin / acos (double-double core) ----------------------------
 * avpm_atan_dd: atan(u), u = uh + ul >= 0, as an unevaluated sum zh + zl with relative error ~2^-68.
 *   u <= 1: k = round(64 u), c = k/64, t = (u - c)/(1 + u c) in double-double (|t| <= 2^-7),
 *           atan u = atan c (table, double-double) + t - t^3/3 + ... - t^11/11
 *   u  > 1: atan u = pi/2 - atan(1/u), 1/u in double-double.
 * The callers keep double-double until one final rounding, so results differ from the correctly
 * rounded value only when the exact value lies within ~2^-15 ulp of a rounding boundary. */
AVP_LIBM_FN double avp_atan2_fd(double y, double x);
AVP_LIBM_FN double avp_asin_fd(double x);
AVP_LIBM_FN double avp_acos_fd(double x);
AVP_LIBM_FN void avpm_two_sum(double a, double b, double* s, double* e)
{
    const double t = a + b;
    const double bb = t - a;
    *e = (a - (t - bb)) + (b - bb);
    *s = t;
}
AVP_LIBM_FN void avpm_div_dd(double nh, double nl, double dh, double dl, double* qh, double* ql)
{
    const double q1 = nh / dh;
    const double r = __builtin_fma(-q1, dh, nh);
    const double q2 = ((r + nl) - q1 * dl) / dh;
    const double t = q1 + q2;
    *ql = q2 - (t - q1);
    *qh = t;
}
AVP_LIBM_FN void avpm_atan_dd(double uh, double ul, double* zh, double* zl)
{
    const double pio2_hi = 0x1.921fb54442d18p+0, pio2_lo = 0x1.1a62633145c07p-54;
    int inv = 0;
    if (uh > 1.0) { double rh, rl; avpm_div_dd(1.0, 0.0, uh, ul, &rh, &rl); uh = rh; ul = rl; inv = 1; }
    const double kf = floor(uh * 64.0 + 0.5);
    const int k = (int)kf;
    const double c = kf * 0.015625;
    /* t = (u - c) / (1 + u c) */
    double nh = uh - c, nl = ul;                       /* uh - c is exact */
    { const double t_ = nh + nl; nl = nl - (t_ - nh); nh = t_; }
    const double ph = uh * c;
    const double pl = __builtin_fma(uh, c, -ph) + ul * c;
    const double dh = 1.0 + ph;
    const double dl = ((1.0 - dh) + ph) + pl;
    double th, tl;
    avpm_div_dd(nh, nl, dh, dl, &th, &tl);
    const double z = th * th;
    const double P = z * (-0x1.5555555555555p-2 + z * (0x1.999999999999ap-3 + z * (-0x1.2492492492492p-3 + z * (0x1.c71c71c71c71cp-4 + z * -0x1.745d1745d1746p-4))));
    const double corr = th * P;
    double sh, se;
    avpm_two_sum(AVP_ATAN_TAB_REF[k][0], th, &sh, &se);
    double low = ((se + AVP_ATAN_TAB_REF[k][1]) + tl) + corr;
    if (inv) {
        double vh, ve;
        avpm_two_sum(pio2_hi, -sh, &vh, &ve);
        low = (ve + pio2_lo) - low;
        sh = vh;
    }
    const double r_ = sh + low;
    *zl = low - (r_ - sh);
    *zh = r_;
}

AVP_LIBM_FN double avp_atan2(double y, double x)
{
    const double pi_hi = 0x1.921fb54442d18p+1, pi_lo_ = 0x1.1a62633145c07p-53;
    if (x == x && y == y && x != 0.0 && y != 0.0 && fabs(x) < 0x1p1000 && fabs(y) < 0x1p1000 && fabs(x) > 0x1p-1000 && fabs(y) > 0x1p-1000) {
        const double ax = fabs(x), ay = fabs(y);
        const int ey = (int)((avpm_hi(ay) >> 20) & 0x7ff), ex = (int)((avpm_hi(ax) >> 20) & 0x7ff);
        if (ey - ex <= 60 && ex - ey <= 60) {
            double uh, ul, zh, zl;
            avpm_div_dd(ay, 0.0, ax, 0.0, &uh, &ul);
            avpm_atan_dd(uh, ul, &zh, &zl);
            double r;
            if (x < 0) { double vh, ve; avpm_two_sum(pi_hi, -zh, &vh, &ve); r = vh + ((ve + pi_lo_) - zl); }
            else r = zh + zl;
            return y < 0 ? -r : r;
        }
    }
    return avp_atan2_fd(y, x);      /* zeros, infinities, NaN, extreme ratios: the classic special-case plumbing */
}

/* asin x = atan2(x, sqrt(1 - x^2)), acos x = atan2(sqrt(1 - x^2), x), with 1 - x^2 and its root in double-double */
AVP_LIBM_FN void avpm_sqrt1mx2_dd(double x, double* sh, double* sl)
{
    const double p = x * x, e = __builtin_fma(x, x, -p);
    double wh, we;
    avpm_two_sum(1.0, -p, &wh, &we);
    const double wl = we - e;
    const double s = sqrt(wh);
    const double rem = __builtin_fma(-s, s, wh) + wl;
    *sh = s;
    *sl = rem / (2.0 * s);
}
AVP_LIBM_FN double avp_asin(double x)
{
    const double ax = fabs(x);
    if (!(ax < 1.0) || ax < 0x1p-26) return avp_asin_fd(x);
    double sh, sl, uh, ul, zh, zl;
    avpm_sqrt1mx2_dd(ax, &sh, &sl);
    avpm_div_dd(ax, 0.0, sh, sl, &uh, &ul);
    avpm_atan_dd(uh, ul, &zh, &zl);
    const double r = zh + zl;
    return x < 0 ? -r : r;
}
AVP_LIBM_FN double avp_acos(double x)
{
    const double pi_hi = 0x1.921fb54442d18p+1, pi_lo_ = 0x1.1a62633145c07p-53;
    const double ax = fabs(x);
    if (!(ax < 1.0) || ax < 0x1p-26) return avp_acos_fd(x);
    double sh, sl, uh, ul, zh, zl;
    avpm_sqrt1mx2_dd(ax, &sh, &sl);
    avpm_div_dd(sh, sl, ax, 0.0, &uh, &ul);
    avpm_atan_dd(uh, ul, &zh, &zl);
    if (x < 0) { double vh, ve; avpm_two_sum(pi_hi, -zh, &vh, &ve); return vh + ((ve + pi_lo_) - zl); }
    return zh + zl;
}

/* ---- fdlibm-form atan2 / asin / acos: special cases (zeros, infinities, NaN, |x| >= 1, tiny x) ---------- */
AVP_LIBM_FN double avp_atan2_fd(double y, double x)
{
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    const double pi_o_2 = 1.5707963267948965580E+00, pi_o_4 = 7.8539816339744827900E-01;
    double z;
    int k, m;
    const uint32_t hx = avpm_hi(x), lx = avpm_lo(x), hy = avpm_hi(y), ly = avpm_lo(y);
    const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (x != x || y != y) return x + y;
    if (hx == 0x3ff00000u && lx == 0) return avp_atan(y);          /* x = 1.0 */
    m = (int)((hy >> 31) & 1) | (int)((hx >> 30) & 2);             /* 2*sign(x) + sign(y) */
    if ((iy | ly) == 0) {                                          /* y = 0 */
        switch (m) {
            case 0: case 1: return y;                              /* atan(+-0, +anything) = +-0 */
            case 2: return pi;
            default: return -pi;
        }
    }
    if ((ix | lx) == 0) return (hy >> 31) ? -pi_o_2 : pi_o_2;      /* x = 0 */
    if (ix == 0x7ff00000u) {                                       /* x = inf */
        if (iy == 0x7ff00000u) {
            switch (m) {
                case 0: return pi_o_4;
                case 1: return -pi_o_4;
                case 2: return 3.0 * pi_o_4;
                default: return -3.0 * pi_o_4;
            }
        } else {
            switch (m) {
                case 0: return 0.0;
                case 1: return -0.0;
                case 2: return pi;
                default: return -pi;
            }
        }
    }
    if (iy == 0x7ff00000u) return (hy >> 31) ? -pi_o_2 : pi_o_2;   /* y = inf */
    k = (int)(iy >> 20) - (int)(ix >> 20);
    if (k > 60) { z = pi_o_2 + 0.5 * pi_lo; m &= 1; }              /* |y/x| > 2^60 */
    else if ((hx >> 31) && k < -60) z = 0.0;                       /* 0 > |y|/x > -2^-60 */
    else z = avp_atan(fabs(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

/* ---- asin / acos ----------------------------------------------------------------------------- */
AVP_LIBM_FN double avpm_asin_R(double t)
{
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    const double p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
    const double q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
    return p / q;
}

AVP_LIBM_FN double avp_asin_fd(double x)
{
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pio4_hi = 7.85398163397448278999e-01;
    double t, w, p, q, c, r, s;
    const uint32_t hx = avpm_hi(x);
    const uint32_t ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {                                       /* |x| >= 1 */
        if (((ix - 0x3ff00000u) | avpm_lo(x)) == 0) return x * pio2_hi + x * pio2_lo;
        return (x - x) / (x - x);                                  /* NaN */
    }
    if (ix < 0x3fe00000u) {                                        /* |x| < 0.5 */
        if (ix < 0x3e500000u) return x;
        return x + x * avpm_asin_R(x * x);
    }
    w = 1.0 - fabs(x);
    t = w * 0.5;
    r = avpm_asin_R(t);
    s = sqrt(t);
    if (ix >= 0x3fef3333u) {                                       /* |x| > 0.975 */
        t = pio2_hi - (2.0 * (s + s * r) - pio2_lo);
    } else {
        w = avpm_with_lo0(s);
        c = (t - w * w) / (s + w);
        p = 2.0 * s * r - (pio2_lo - 2.0 * c);
        q = pio4_hi - 2.0 * w;
        t = pio4_hi - (p - q);
    }
    return (hx >> 31) ? -t : t;
}

AVP_LIBM_FN double avp_acos_fd(double x)
{
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
    double z, r, s, w, c, df;
    const uint32_t hx = avpm_hi(x);
    const uint32_t ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {
        if (((ix - 0x3ff00000u) | avpm_lo(x)) == 0) return (hx >> 31) ? pi + 2.0 * pio2_lo : 0.0;
        return (x - x) / (x - x);
    }
    if (ix < 0x3fe00000u) {                                        /* |x| < 0.5 */
        if (ix <= 0x3c600000u) return pio2_hi + pio2_lo;
        z = x * x;
        r = avpm_asin_R(z);
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx >> 31) {                                                /* x < -0.5 */
        z = (1.0 + x) * 0.5;
        s = sqrt(z);
        r = avpm_asin_R(z);
        w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    z = (1.0 - x) * 0.5;                                           /* x > 0.5 */
    s = sqrt(z);
    df = avpm_with_lo0(s);
    c = (z - df * df) / (s + df);
    r = avpm_asin_R(z);
    w = r * s + c;
    return 2.0 * (df + w);
}

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

/* tan, nearly correctly rounded on |x| < 3.2 (the SLS words use 0 < x < pi): one Newton step on
 * atan(t) = x with the double-double atan core: t = t0 + (y - atan t0)(1 + t0^2), y = x reduced by +-pi. */
AVP_LIBM_FN double avp_tan(double x)
{
    const double pi_hi = 0x1.921fb54442d18p+1, pi_lo_ = 0x1.1a62633145c07p-53, pio2 = 0x1.921fb54442d18p+0;
    const double t0 = avp_tan_fd(x);
    const double a = fabs(t0);
    if (!(fabs(x) < 3.2) || !(a > 0x1p-20) || !(a < 0x1p20)) return t0;
    double zh, zl;
    avpm_atan_dd(a, 0.0, &zh, &zl);
    if (t0 < 0) { zh = -zh; zl = -zl; }
    /* y = x - n*pi with n in {-1, 0, 1} so that y is in (-pi/2, pi/2) */
    double yh = x, yl = 0.0;
    if (x > pio2) { double vh, ve; avpm_two_sum(x, -pi_hi, &vh, &ve); yh = vh; yl = ve - pi_lo_; }
    else if (x < -pio2) { double vh, ve; avpm_two_sum(x, pi_hi, &vh, &ve); yh = vh; yl = ve + pi_lo_; }
    double eh, ee;
    avpm_two_sum(yh, -zh, &eh, &ee);
    const double err = eh + ((ee + yl) - zl);
    return t0 + err * (1.0 + t0 * t0);
}

#endif /* AVP_LIBM_H */
