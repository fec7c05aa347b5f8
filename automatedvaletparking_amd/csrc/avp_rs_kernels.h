// avp_rs_kernels.h -- Reeds-Shepp analytic shots on the device.
//
// Replaces rs_curve.calc_optimal_path (path_plan/rs_curve.py:99-134) and everything under it:
// generate_path :627-644 (46 word evaluations in 6 families, source order), set_path :137-156
// (signed-sum duplicate filter, L >= 1000 reject, L >= 0.01 assertion), arg-min with "<=" (last of
// equal minima wins :106-108), generate_local_course/interpolate :537-624 for the winner, world
// transform :125-131.
//
// Every libm call of the reference is reproduced bit for bit: sin/cos are the glibc-exact avp_sin/avp_cos, hypot is
// CPython's algorithm (avp_hypot), float % is CPython's (avp_pymod), and tan/atan2/asin/acos and the libm pow(v, 2.0)
// behind Python's v ** 2 are the restatements of what glibc 2.35's x86-64 FMA build executes (include/avp_glibc_libm.h
// via include/avp_libm.h; host build == platform libm on 1e9 arguments per function, device == host build in
// tests/test_gpu_rs.py). Exact ties between mirror-image words therefore fall the way they fall in the reference.
#pragma once
#include "avp_device.h"
#include "../../include/avp_libm.h"

enum { RS_S = 0, RS_L = 1, RS_R = 2 };

// ---- scalar maths as CALLED leaf functions (PL_LIBM_CALLS) ----------------------------------------------------
// Fully inlined, the double-double sin/cos, atan2, asin/acos, tan, fmod and hypot bodies make up 60 % of the planner
// kernel's 290 KB of code -- several times the instruction cache a pair of CUs shares -- and every wave of a
// workgroup runs a different part of it. As leaf functions (no stack use: arguments and results in registers)
// each body exists once. Same arithmetic, same results; only the code layout changes. The collision kernels
// (avp_check_kernels.h, included before this header) keep the inlined forms.
#ifndef PL_LIBM_CALLS
#define PL_LIBM_CALLS 1
#endif
#if PL_LIBM_CALLS && defined(__HIP_DEVICE_COMPILE__)
struct AvpSinCos { double s, c; };
__device__ __noinline__ AvpSinCos avp_sincos_fn(double x) { AvpSinCos r; avp_sincos(x, r.s, r.c); return r; }
__device__ __noinline__ double avp_sin_fn(double x) { return avp_sin(x); }
__device__ __noinline__ double avp_cos_fn(double x) { return avp_cos(x); }
__device__ __noinline__ double avp_atan2_fn(double y, double x) { return avp_atan2(y, x); }
__device__ __noinline__ double avp_asin_fn(double x) { return avp_asin(x); }
__device__ __noinline__ double avp_acos_fn(double x) { return avp_acos(x); }
__device__ __noinline__ double avp_tan_fn(double x) { return avp_tan(x); }
__device__ __noinline__ double avp_pow2_fn(double x) { return avp_pow2(x); }
__device__ __noinline__ double avp_M_fn(double x) { return avp_M(x); }
__device__ __noinline__ double avp_pi_2_pi_fn(double x) { return avp_pi_2_pi(x); }
__device__ __noinline__ double avp_hypot_fn(double a, double b) { return avp_hypot(a, b); }
#define avp_sincos(x, s_, c_) do { const AvpSinCos r_ = avp_sincos_fn(x); (s_) = r_.s; (c_) = r_.c; } while (0)
#define avp_sin(x) avp_sin_fn(x)
#define avp_cos(x) avp_cos_fn(x)
#define avp_atan2(y, x) avp_atan2_fn(y, x)
#define avp_asin(x) avp_asin_fn(x)
#define avp_acos(x) avp_acos_fn(x)
#define avp_tan(x) avp_tan_fn(x)
#define avp_pow2(x) avp_pow2_fn(x)
#define avp_M(x) avp_M_fn(x)
#define avp_pi_2_pi(x) avp_pi_2_pi_fn(x)
#define avp_hypot(a, b) avp_hypot_fn(a, b)
#endif

struct RsPath {
    int n;                 // segments (3..5); 0 = none
    int8_t t[AVP_RS_MAXSEG];
    double l[AVP_RS_MAXSEG];   // normalised (unit turning radius) inside rs_generate; metres after rs_optimal
    double L;
};

AVP_D void rs_polar(double x, double y, double& r, double& th) { r = avp_hypot(x, y); th = avp_atan2(y, x); }

// rs_curve.py:159-167
AVP_D bool rs_LSL(double x, double y, double phi, double sp, double cp, double& t, double& u, double& v)
{
    double uu, tt;
    rs_polar(x - sp, y - 1.0 + cp, uu, tt);
    if (tt >= 0.0) {
        const double vv = avp_M(phi - tt);
        if (vv >= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
// rs_curve.py:170-183
AVP_D bool rs_LSR(double x, double y, double phi, double sp, double cp, double& t, double& u, double& v)
{
    double u1, t1;
    rs_polar(x + sp, y - 1.0 - cp, u1, t1);
    u1 = avp_pow2(u1);                                            // u1 ** 2: libm pow
    if (u1 >= 4.0) {
        const double uu = sqrt(u1 - 4.0);
        const double theta = avp_atan2(2.0, uu);
        const double tt = avp_M(t1 + theta);
        const double vv = avp_M(tt - phi);
        if (tt >= 0.0 && vv >= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
// rs_curve.py:186-197
AVP_D bool rs_LRL(double x, double y, double phi, double sp, double cp, double& t, double& u, double& v)
{
    double u1, t1;
    rs_polar(x - sp, y - 1.0 + cp, u1, t1);
    if (u1 <= 4.0) {
        const double uu = -2.0 * avp_asin(0.25 * u1);
        const double tt = avp_M(t1 + 0.5 * uu + AVP_PI);
        const double vv = avp_M(phi - tt + uu);
        if (tt >= 0.0 && uu <= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
// rs_curve.py:213-229
AVP_D bool rs_SLS(double x, double y, double phi, double& t, double& u, double& v)
{
    // rs_curve.py:213-229: the y > 0 and y < 0 branches differ only in the sign of the square root, and each
    // evaluates tan(phi) once and tan(phi / 2) twice: one evaluation of each serves both (same values).
    phi = avp_M(phi);
    if (!(0.0 < phi && phi < AVP_PI * 0.99) || !(y > 0.0 || y < 0.0)) return false;
    const double tan_phi = avp_tan(phi), tan_half = avp_tan(phi / 2.0);
    const double xd = -y / tan_phi + x;
    const double r = sqrt(avp_pow2(x - xd) + avp_pow2(y));          // math.sqrt((x - xd) ** 2 + y ** 2): ** is libm pow
    t = xd - tan_half;
    u = phi;
    v = (y > 0.0 ? r : -r) - tan_half;
    return true;
}
// rs_curve.py:308-323
AVP_D void rs_tauOmega(double u, double v, double xi, double eta, double phi, double& tau, double& omega)
{
    const double delta = avp_M(u - v);
    double sin_u, cos_u, sin_d, cos_d;                            // each cosine is used twice below
    avp_sincos(u, sin_u, cos_u);
    avp_sincos(delta, sin_d, cos_d);
    const double A = sin_u - sin_d;
    const double B = cos_u - cos_d - 1.0;
    const double t1 = avp_atan2(eta * A - xi * B, xi * A + eta * B);
    double sin_v, cos_v;                                          // (the branch-free fused form; sin_v unused)
    avp_sincos(v, sin_v, cos_v);
    const double t2 = 2.0 * (cos_d - cos_v - cos_u) + 3.0;
    tau = t2 < 0 ? avp_M(t1 + AVP_PI) : avp_M(t1);
    omega = avp_M(tau - u + v - phi);
}
// rs_curve.py:326-337
AVP_D bool rs_LRLRn(double x, double y, double phi, double sp, double cp, double& t, double& u, double& v)
{
    const double xi = x + sp, eta = y - 1.0 - cp;
    const double rho = 0.25 * (2.0 + sqrt(xi * xi + eta * eta));
    if (rho <= 1.0) {
        const double uu = avp_acos(rho);
        double tt, vv;
        rs_tauOmega(uu, -uu, xi, eta, phi, tt, vv);
        if (tt >= 0.0 && vv <= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
// rs_curve.py:340-352
AVP_D bool rs_LRLRp(double x, double y, double phi, double sp, double cp, double& t, double& u, double& v)
{
    const double xi = x + sp, eta = y - 1.0 - cp;
    const double rho = (20.0 - xi * xi - eta * eta) / 16.0;
    if (0.0 <= rho && rho <= 1.0) {
        const double uu = -avp_acos(rho);
        if (uu >= -0.5 * AVP_PI) {
            double tt, vv;
            rs_tauOmega(uu, uu, xi, eta, phi, tt, vv);
            if (tt >= 0.0 && vv >= 0.0) { t = tt; u = uu; v = vv; return true; }
        }
    }
    return false;
}
// rs_curve.py:391-403
AVP_D bool rs_LRSR(double x, double y, double phi, double sp, double cp, double& t, double& u, double& v)
{
    const double xi = x + sp, eta = y - 1.0 - cp;
    double rho, theta;
    rs_polar(-eta, xi, rho, theta);
    if (rho >= 2.0) {
        const double tt = theta, uu = 2.0 - rho, vv = avp_M(tt + 0.5 * AVP_PI - phi);
        if (tt >= 0.0 && uu <= 0.0 && vv <= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
// rs_curve.py:406-419
AVP_D bool rs_LRSL(double x, double y, double phi, double sp, double cp, double& t, double& u, double& v)
{
    const double xi = x - sp, eta = y - 1.0 + cp;
    double rho, theta;
    rs_polar(xi, eta, rho, theta);
    if (rho >= 2.0) {
        const double r = sqrt(rho * rho - 4.0);
        const double uu = 2.0 - r;
        const double tt = avp_M(theta + avp_atan2(r, -2.0));
        const double vv = avp_M(phi - 0.5 * AVP_PI - tt);
        if (tt >= 0.0 && uu <= 0.0 && vv <= 0.0) { t = tt; u = uu; v = vv; return true; }
    }
    return false;
}
// rs_curve.py:494-510
AVP_D bool rs_LRSLR(double x, double y, double phi, double sp, double cp, double& t, double& u, double& v)
{
    const double xi = x + sp, eta = y - 1.0 - cp;
    double rho, theta;
    rs_polar(xi, eta, rho, theta);
    if (rho >= 2.0) {
        const double uu = 4.0 - sqrt(rho * rho - 4.0);
        if (uu <= 0.0) {
            const double tt = avp_M(avp_atan2((4.0 - uu) * xi - 2.0 * eta, -2.0 * xi + (uu - 4.0) * eta));
            const double vv = avp_M(tt - phi);
            if (tt >= 0.0 && vv >= 0.0) { t = tt; u = uu; v = vv; return true; }
        }
    }
    return false;
}

// Word table: the 46 evaluations of generate_path in source order. Each row: solver id, sign of x
// (the mirrored/time-flipped variants call f(-x, y, -phi) etc.), sign of y, sign of phi, "backwards"
// frame flag, output permutation/sign recipe and the type word.
//   solver: 0 SLS 1 LSL 2 LSR 3 LRL 4 LRLRn 5 LRLRp 6 LRSL 7 LRSR 8 LRSLR
//   recipe: 0 (t,u,v) 1 -(t,u,v) 2 (v,u,t) 3 -(v,u,t)
//           4 (t,u,-u,v) 5 (-t,-u,u,-v) 6 (t,u,u,v) 7 -(t,u,u,v)
//           8 (t,-hp,u,v) 9 (-t,hp,-u,-v) 10 (v,u,-hp,t) 11 (-v,-u,hp,-t)
//           12 (t,-hp,u,-hp,v) 13 (-t,hp,-u,hp,-v)
struct RsWord { int8_t solver, sx, sy, sphi, back, recipe, n, a, b, c, d, e; };
#define W_(solver, sx, sy, sp, back, rec, n, a, b, c, d, e) { solver, sx, sy, sp, back, rec, n, a, b, c, d, e }
static __device__ const RsWord RS_WORDS_G[46] = {
    // SCS :200-210
    W_(0, 1, 1, 1, 0, 0, 3, RS_S, RS_L, RS_S, 0, 0), W_(0, 1, -1, -1, 0, 0, 3, RS_S, RS_R, RS_S, 0, 0),
    // CSC :232-265
    W_(1, 1, 1, 1, 0, 0, 3, RS_L, RS_S, RS_L, 0, 0), W_(1, -1, 1, -1, 0, 1, 3, RS_L, RS_S, RS_L, 0, 0),
    W_(1, 1, -1, -1, 0, 0, 3, RS_R, RS_S, RS_R, 0, 0), W_(1, -1, -1, 1, 0, 1, 3, RS_R, RS_S, RS_R, 0, 0),
    W_(2, 1, 1, 1, 0, 0, 3, RS_L, RS_S, RS_R, 0, 0), W_(2, -1, 1, -1, 0, 1, 3, RS_L, RS_S, RS_R, 0, 0),
    W_(2, 1, -1, -1, 0, 0, 3, RS_R, RS_S, RS_L, 0, 0), W_(2, -1, -1, 1, 0, 1, 3, RS_R, RS_S, RS_L, 0, 0),
    // CCC :268-305
    W_(3, 1, 1, 1, 0, 0, 3, RS_L, RS_R, RS_L, 0, 0), W_(3, -1, 1, -1, 0, 1, 3, RS_L, RS_R, RS_L, 0, 0),
    W_(3, 1, -1, -1, 0, 0, 3, RS_R, RS_L, RS_R, 0, 0), W_(3, -1, -1, 1, 0, 1, 3, RS_R, RS_L, RS_R, 0, 0),
    W_(3, 1, 1, 1, 1, 2, 3, RS_L, RS_R, RS_L, 0, 0), W_(3, -1, 1, -1, 1, 3, 3, RS_L, RS_R, RS_L, 0, 0),
    W_(3, 1, -1, -1, 1, 2, 3, RS_R, RS_L, RS_R, 0, 0), W_(3, -1, -1, 1, 1, 3, 3, RS_R, RS_L, RS_R, 0, 0),
    // CCCC :355-388
    W_(4, 1, 1, 1, 0, 4, 4, RS_L, RS_R, RS_L, RS_R, 0), W_(4, -1, 1, -1, 0, 5, 4, RS_L, RS_R, RS_L, RS_R, 0),
    W_(4, 1, -1, -1, 0, 4, 4, RS_R, RS_L, RS_R, RS_L, 0), W_(4, -1, -1, 1, 0, 5, 4, RS_R, RS_L, RS_R, RS_L, 0),
    W_(5, 1, 1, 1, 0, 6, 4, RS_L, RS_R, RS_L, RS_R, 0), W_(5, -1, 1, -1, 0, 7, 4, RS_L, RS_R, RS_L, RS_R, 0),
    W_(5, 1, -1, -1, 0, 6, 4, RS_R, RS_L, RS_R, RS_L, 0), W_(5, -1, -1, 1, 0, 7, 4, RS_R, RS_L, RS_R, RS_L, 0),
    // CCSC :422-491
    W_(6, 1, 1, 1, 0, 8, 4, RS_L, RS_R, RS_S, RS_L, 0), W_(6, -1, 1, -1, 0, 9, 4, RS_L, RS_R, RS_S, RS_L, 0),
    W_(6, 1, -1, -1, 0, 8, 4, RS_R, RS_L, RS_S, RS_R, 0), W_(6, -1, -1, 1, 0, 9, 4, RS_R, RS_L, RS_S, RS_R, 0),
    W_(7, 1, 1, 1, 0, 8, 4, RS_L, RS_R, RS_S, RS_R, 0), W_(7, -1, 1, -1, 0, 9, 4, RS_L, RS_R, RS_S, RS_R, 0),
    W_(7, 1, -1, -1, 0, 8, 4, RS_R, RS_L, RS_S, RS_L, 0), W_(7, -1, -1, 1, 0, 9, 4, RS_R, RS_L, RS_S, RS_L, 0),
    W_(6, 1, 1, 1, 1, 10, 4, RS_L, RS_S, RS_R, RS_L, 0), W_(6, -1, 1, -1, 1, 11, 4, RS_L, RS_S, RS_R, RS_L, 0),
    W_(6, 1, -1, -1, 1, 10, 4, RS_R, RS_S, RS_L, RS_R, 0), W_(6, -1, -1, 1, 1, 11, 4, RS_R, RS_S, RS_L, RS_R, 0),
    W_(7, 1, 1, 1, 1, 10, 4, RS_R, RS_S, RS_R, RS_L, 0), W_(7, -1, 1, -1, 1, 11, 4, RS_R, RS_S, RS_R, RS_L, 0),
    W_(7, 1, -1, -1, 1, 10, 4, RS_L, RS_S, RS_L, RS_R, 0), W_(7, -1, -1, 1, 1, 11, 4, RS_L, RS_S, RS_L, RS_R, 0),
    // CCSCC :513-534
    W_(8, 1, 1, 1, 0, 12, 5, RS_L, RS_R, RS_S, RS_L, RS_R), W_(8, -1, 1, -1, 0, 13, 5, RS_L, RS_R, RS_S, RS_L, RS_R),
    W_(8, 1, -1, -1, 0, 12, 5, RS_R, RS_L, RS_S, RS_R, RS_L), W_(8, -1, -1, 1, 0, 13, 5, RS_R, RS_L, RS_S, RS_R, RS_L),
};
#undef W_

// Words grouped by type sequence (set_path only compares candidates of identical ctypes), ascending
// word index inside a group, -1 padded.
static __device__ const int8_t RS_GROUPS_G[20][4] = {
    { 0, -1, -1, -1 }, { 1, -1, -1, -1 }, { 2, 3, -1, -1 }, { 4, 5, -1, -1 }, { 6, 7, -1, -1 }, { 8, 9, -1, -1 },
    { 10, 11, 14, 15 }, { 12, 13, 16, 17 }, { 18, 19, 22, 23 }, { 20, 21, 24, 25 },
    { 26, 27, -1, -1 }, { 28, 29, -1, -1 }, { 30, 31, -1, -1 }, { 32, 33, -1, -1 },
    { 34, 35, -1, -1 }, { 36, 37, -1, -1 }, { 38, 39, -1, -1 }, { 40, 41, -1, -1 }, { 42, 43, -1, -1 }, { 44, 45, -1, -1 },
};
// Like the trig tables, the word / type-group tables are read through LDS copies (filled by rs_lds_tables_fill): set_path
// walks them inside dependent loops, where a global read costs several hundred cycles each.
__shared__ RsWord RS_WORDS[46];
__shared__ int8_t RS_GROUPS[20][4];
__device__ __forceinline__ void rs_lds_tables_fill()
{
    for (int i = threadIdx.x; i < 46; i += blockDim.x) RS_WORDS[i] = RS_WORDS_G[i];
    for (int i = threadIdx.x; i < 80; i += blockDim.x) (&RS_GROUPS[0][0])[i] = (&RS_GROUPS_G[0][0])[i];
    __syncthreads();
}


// Start-frame normalisation of generate_path (rs_curve.py:627-634) + the "backwards" frame
// (:286-287, :456-457).
struct RsFrame { double x0, y0, phi0, xb, yb, sphi, cphi; };
AVP_D RsFrame rs_frame(double q0x, double q0y, double q0t, double q1x, double q1y, double q1t, double maxc)
{
    RsFrame f;
    const double dx = q1x - q0x, dy = q1y - q0y;
    f.phi0 = q1t - q0t;
    double c, s;
    avp_sincos(q0t, s, c);
    f.x0 = (c * dx + s * dy) * maxc;
    f.y0 = (-s * dx + c * dy) * maxc;
    avp_sincos(f.phi0, f.sphi, f.cphi);   // sin is odd and cos even bit for bit, so the mirrored / time-flipped
                                          // words (phi -> -phi) reuse these two values
    f.xb = f.x0 * f.cphi + f.y0 * f.sphi;
    f.yb = f.x0 * f.sphi - f.y0 * f.cphi;
    return f;
}

// Word w of the 46 (source order): returns validity and the signed normalised segment lengths.
__device__ __forceinline__ bool rs_word(int w, const RsFrame& f, double l[5])
{
    const RsWord W = RS_WORDS[w];
    const double bx = W.back ? f.xb : f.x0, by = W.back ? f.yb : f.y0;
    const double x = W.sx < 0 ? -bx : bx, y = W.sy < 0 ? -by : by, phi = W.sphi < 0 ? -f.phi0 : f.phi0;
    const double sp = W.sphi < 0 ? -f.sphi : f.sphi, cp = f.cphi;
    const double hp = 0.5 * AVP_PI;
    double t = 0, u = 0, v = 0;
    bool ok;
    switch (W.solver) {
        case 0: ok = rs_SLS(x, y, phi, t, u, v); break;
        case 1: ok = rs_LSL(x, y, phi, sp, cp, t, u, v); break;
        case 2: ok = rs_LSR(x, y, phi, sp, cp, t, u, v); break;
        case 3: ok = rs_LRL(x, y, phi, sp, cp, t, u, v); break;
        case 4: ok = rs_LRLRn(x, y, phi, sp, cp, t, u, v); break;
        case 5: ok = rs_LRLRp(x, y, phi, sp, cp, t, u, v); break;
        case 6: ok = rs_LRSL(x, y, phi, sp, cp, t, u, v); break;
        case 7: ok = rs_LRSR(x, y, phi, sp, cp, t, u, v); break;
        default: ok = rs_LRSLR(x, y, phi, sp, cp, t, u, v); break;
    }
    l[0] = l[1] = l[2] = l[3] = l[4] = 0.0;
    if (!ok) return false;
    switch (W.recipe) {
        case 0: l[0] = t; l[1] = u; l[2] = v; break;
        case 1: l[0] = -t; l[1] = -u; l[2] = -v; break;
        case 2: l[0] = v; l[1] = u; l[2] = t; break;
        case 3: l[0] = -v; l[1] = -u; l[2] = -t; break;
        case 4: l[0] = t; l[1] = u; l[2] = -u; l[3] = v; break;
        case 5: l[0] = -t; l[1] = -u; l[2] = u; l[3] = -v; break;
        case 6: l[0] = t; l[1] = u; l[2] = u; l[3] = v; break;
        case 7: l[0] = -t; l[1] = -u; l[2] = -u; l[3] = -v; break;
        case 8: l[0] = t; l[1] = -hp; l[2] = u; l[3] = v; break;
        case 9: l[0] = -t; l[1] = hp; l[2] = -u; l[3] = -v; break;
        case 10: l[0] = v; l[1] = u; l[2] = -hp; l[3] = t; break;
        case 11: l[0] = -v; l[1] = -u; l[2] = hp; l[3] = -t; break;
        case 12: l[0] = t; l[1] = -hp; l[2] = u; l[3] = -hp; l[4] = v; break;
        default: l[0] = -t; l[1] = hp; l[2] = -u; l[3] = hp; l[4] = -v; break;
    }
    return true;
}

// One word as a CALLED leaf function (value in, value out: no stack): rs_optimal evaluates the words in four unrolled
// call sites of a loop over the type groups, and inlining the nine solvers four times would only bloat the kernel.
struct RsWordOut { double l0, l1, l2, l3, l4; int ok; };
__device__ __noinline__ RsWordOut rs_word_fn(int w, RsFrame f)
{
    double l[5];
    RsWordOut o;
    o.ok = rs_word(w, f, l) ? 1 : 0;
    o.l0 = l[0]; o.l1 = l[1]; o.l2 = l[2]; o.l3 = l[3]; o.l4 = l[4];
    return o;
}

// calc_optimal_path (rs_curve.py:99-134 with generate_path :627-644 and set_path :137-156) for one query in one thread,
// everything in registers: set_path only ever compares a candidate with kept candidates of the SAME type sequence, so the
// 46 words are visited type group by type group (RS_GROUPS, word order inside a group = source order), the <= 3 kept
// predecessors of a group live in registers, and the optimum is a running (length, word) pair -- the later word wins a
// tie, which is what "<=" over the kept list in source order does (:103-108). No candidate list, no stack object.
// Returns 0 ok (normalised lengths in `out`), 1 no candidate, 2 the reference's assertion L >= 0.01 fails.
AVP_D int rs_optimal(double q0x, double q0y, double q0t, double q1x, double q1y, double q1t, double maxc, RsPath& out)
{
    const RsFrame f = rs_frame(q0x, q0y, q0t, q1x, q1y, q1t, maxc);
    double bestLm = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0, b4 = 0.0;
    int bestW = -1;
    bool err = false;
    for (int g = 0; g < 20; g++) {
        unsigned accmask = 0;
        double kept[3][5];
        bool live = true;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int wd = RS_GROUPS[g][j];
            if (wd < 0) live = false;
            RsWordOut o;
            o.l0 = o.l1 = o.l2 = o.l3 = o.l4 = 0.0; o.ok = 0;
            if (live) o = rs_word_fn(wd, f);
            const double l[5] = { o.l0, o.l1, o.l2, o.l3, o.l4 };
            if (j < 3) {
#pragma unroll
                for (int i = 0; i < 5; i++) kept[j][i] = l[i];
            }
            bool dup = false;
#pragma unroll
            for (int e = 0; e < 3; e++) {
                if (e >= j) continue;
                double sum = 0;
#pragma unroll
                for (int i = 0; i < 5; i++) sum = sum + (kept[e][i] - l[i]);          // the unused tail of both is 0.0
                if ((accmask & (1u << e)) && sum <= 0.01) dup = true;
            }
            double L = 0;
#pragma unroll
            for (int i = 0; i < 5; i++) L = L + fabs(l[i]);
            if (!o.ok || dup || L >= 1000.0) continue;
            if (!(L >= 0.01)) { err = true; continue; }
            accmask |= 1u << j;
            const double Lm = L / maxc;
            if (bestW < 0 || Lm < bestLm || (Lm == bestLm && wd > bestW)) { bestLm = Lm; bestW = wd; b0 = l[0]; b1 = l[1]; b2 = l[2]; b3 = l[3]; b4 = l[4]; }
        }
    }
    out.n = 0; out.L = 0;
    if (err) return 2;
    if (bestW < 0) return 1;
    const RsWord W = RS_WORDS[bestW];
    out.n = W.n;
    out.t[0] = 0 < W.n ? W.a : (int8_t)-1; out.t[1] = 1 < W.n ? W.b : (int8_t)-1; out.t[2] = 2 < W.n ? W.c : (int8_t)-1;
    out.t[3] = 3 < W.n ? W.d : (int8_t)-1; out.t[4] = 4 < W.n ? W.e : (int8_t)-1;
    out.l[0] = b0; out.l[1] = b1; out.l[2] = b2; out.l[3] = b3; out.l[4] = b4;
    out.L = ((((0.0 + fabs(b0)) + fabs(b1)) + fabs(b2)) + fabs(b3)) + fabs(b4);
    return 0;
}

// rs_curve.py:597-624
AVP_D void rs_interpolate(double l, int m, double maxc, double ox, double oy, double oyaw, double& px, double& py, double& pyaw)
{
    // One evaluation of sincos(oyaw) serves both segment kinds: the straight segment uses it directly (:599-601), the arcs
    // use sin(-oyaw) = -sin(oyaw), cos(-oyaw) = cos(oyaw) -- exact identities of the restated glibc algorithm (every
    // step is odd / even in its argument) -- so a wave whose lanes hold both kinds runs two sincos bodies, not three.
    double sy, cy;
    avp_sincos(oyaw, sy, cy);
    double sl = 0.0, cl = 1.0;
    if (m != RS_S) avp_sincos(l, sl, cl);                                              // each used twice (:605-612)
    if (m == RS_S) {
        px = ox + l / maxc * cy;
        py = oy + l / maxc * sy;
        pyaw = oyaw;
    } else {
        const double so = -sy, co = cy;                                                // sin, cos of -oyaw
        const double ldx = sl / maxc;
        const double ldy = (m == RS_L) ? (1.0 - cl) / maxc : (1.0 - cl) / (-maxc);
        const double gdx = co * ldx + so * ldy;
        const double gdy = -so * ldx + co * ldy;
        px = ox + gdx;
        py = oy + gdy;
        pyaw = (m == RS_L) ? oyaw + l : oyaw - l;
    }
}

// rs_curve.py:537-594 + :125-131. `p` holds normalised lengths. Writes world-frame samples
// (x, y, pi_2_pi(yaw)) with stride `stride` doubles and directions; returns the number of points,
// or -1 when cap is too small. The index bookkeeping (segment ends overwritten by the next
// segment's first step, trailing px == 0.0 entries dropped) follows the reference exactly.
__device__ __noinline__ int rs_sample(const RsPath& p, double maxc, double q0x, double q0y, double q0t, double* xyyaw, int stride,
                    int8_t* dir, int cap)
{
    const double step = 0.5 * maxc;
    const int point_num = (int)(p.L / step) + p.n + 3;
    if (point_num > cap) return -1;
    for (int i = 0; i < point_num; i++) { xyyaw[i * stride] = 0.0; xyyaw[i * stride + 1] = 0.0; xyyaw[i * stride + 2] = 0.0; if (dir) dir[i] = 0; }
    int ind = 1;
    if (dir) dir[0] = p.l[0] > 0.0 ? 1 : -1;
    double d = p.l[0] > 0.0 ? step : -step;
    double pd = d, ll = 0.0;
    for (int i = 0; i < p.n; i++) {
        const double l = p.l[i];
        const int m = p.t[i];
        d = l > 0.0 ? step : -step;
        const double ox = xyyaw[ind * stride], oy = xyyaw[ind * stride + 1], oyaw = xyyaw[ind * stride + 2];
        ind -= 1;
        if (i >= 1 && (p.l[i - 1] * p.l[i]) > 0) pd = -d - ll; else pd = d - ll;
        while (fabs(pd) <= fabs(l)) {
            ind += 1;
            rs_interpolate(pd, m, maxc, ox, oy, oyaw, xyyaw[ind * stride], xyyaw[ind * stride + 1], xyyaw[ind * stride + 2]);
            if (dir) dir[ind] = pd > 0.0 ? 1 : -1;
            pd += d;
        }
        ll = l - pd - d;
        ind += 1;
        rs_interpolate(l, m, maxc, ox, oy, oyaw, xyyaw[ind * stride], xyyaw[ind * stride + 1], xyyaw[ind * stride + 2]);
        if (dir) dir[ind] = l > 0.0 ? 1 : -1;
    }
    int np = point_num;
    while (np > 0 && xyyaw[(np - 1) * stride] == 0.0) np--;
    double cm, sm;
    avp_sincos(-q0t, sm, cm);
    for (int i = 0; i < np; i++) {
        const double ix = xyyaw[i * stride], iy = xyyaw[i * stride + 1];
        xyyaw[i * stride] = cm * ix + sm * iy + q0x;
        xyyaw[i * stride + 1] = -sm * ix + cm * iy + q0y;
        xyyaw[i * stride + 2] = avp_pi_2_pi(xyyaw[i * stride + 2] + q0t);
    }
    return np;
}

// One thread = one (q0, q1) query.
__global__ __launch_bounds__(64) void rs_optimal_kernel(const double* __restrict__ q0, const double* __restrict__ q1,
                                                        double maxc, int64_t n, int32_t maxpts, int32_t* __restrict__ status,
                                                        double* __restrict__ L, int8_t* __restrict__ types,
                                                        double* __restrict__ lens, int32_t* __restrict__ npts,
                                                        double* __restrict__ xyyaw, int8_t* __restrict__ dir)
{
    avp_lds_tables_fill<true>();
    rs_lds_tables_fill();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the winning word lives in LDS (its segments are indexed at run time by the sampler: a private copy would be a
    // scratch array), one record per lane at an odd stride in doubles
    static_assert(sizeof(RsPath) <= 8 * 8, "RsPath record");
    __shared__ double lp[64 * 9];
    RsPath& p = *reinterpret_cast<RsPath*>(lp + 9 * threadIdx.x);
    const double ax = q0[3 * i], ay = q0[3 * i + 1], at = q0[3 * i + 2];
    const double bx = q1[3 * i], by = q1[3 * i + 1], bt = q1[3 * i + 2];
    // status 7: a pose the reference itself never returns on -- pi_2_pi's subtract-2-pi loop (rs_curve.py:649-656) on an infinite heading,
    // headings up to 1e6 rad are wrapped by that very loop, bit for bit; coordinates must be finite (NaN: refused too)
    if (!(fabs(ax) <= 1.7e308 && fabs(ay) <= 1.7e308 && fabs(at) <= 1e6 && fabs(bx) <= 1.7e308 && fabs(by) <= 1.7e308 && fabs(bt) <= 1e6)) {
        for (int k = 0; k < 5; k++) { types[i * 5 + k] = (int8_t)-1; lens[i * 5 + k] = 0.0; }
        L[i] = 0.0; npts[i] = 0; status[i] = 7;
        return;
    }
    int st = rs_optimal(ax, ay, at, bx, by, bt, maxc, p);
    for (int k = 0; k < 5; k++) { types[i * 5 + k] = st ? (int8_t)-1 : p.t[k]; lens[i * 5 + k] = st ? 0.0 : p.l[k] / maxc; }
    L[i] = st ? 0.0 : p.L / maxc;
    int np = 0;
    if (!st && maxpts > 0 && xyyaw) {
        np = rs_sample(p, maxc, ax, ay, at, xyyaw + (size_t)i * maxpts * 3, 3, dir ? dir + (size_t)i * maxpts : nullptr, maxpts);
        if (np < 0) { st = 3; np = (int)(p.L / (0.5 * maxc)) + p.n + 3; }
    }
    npts[i] = np;
    status[i] = st;
}
