/*
 * avp_oracle.c -- CPU restatement of the reference hybrid-A* hot path.
 *
 * TEST INFRASTRUCTURE ONLY. This file is the parity oracle for the MI355X HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it. The product
 * path (automatedvaletparking_amd/, libavp_hip.so) never links or calls anything in oracle/.
 *
 * It restates, function by function, the algorithm of wenqing-2021/AutomatedValetParking
 * (pure Python + numpy; citations are file:line under /root/reference). Arithmetic that the
 * reference delegates to third-party code is reproduced through the same dependency where it is
 * available to C (glibc 2.35 libm: sin cos tan atan2 asin acos sqrt pow fmod, called un-folded,
 * this file is built with -fno-builtin -ffp-contract=off) and restated where it is not:
 *   - CPython 3.10 heapq (Lib/heapq.py:130-160,205-276)          -> heap_* below
 *   - CPython 3.10 float %  (Objects/floatobject.c float_rem)    -> py_fmod
 *   - CPython 3.10 math.hypot (Modules/mathmodule.c vector_norm) -> py_hypot
 *   - numpy/OpenBLAS 2x2 dgemv and 2-element ddot as executed in the build container
 *     (acc = a0*b0; acc = fma(a1, b1, acc)), measured, see DESIGN.md
 *   - scipy 1.15 spatial.distance.cosine (1 - u.v/sqrt(u.u*v.v), clipped to [0,2])
 * Parity is pinned: tests/test_oracle_golden.py checks every function here against golden vectors
 * captured from the unmodified reference (oracle/gen_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define ORC_API __attribute__((visibility("default")))

/* "restated libm" mode (check instrumentation; default 0 = the platform's glibc libm, the reference's arithmetic,
 * pinned by the golden vectors): atan2 / asin / acos / tan / pow(v, 2.0) come from include/avp_libm.h -- the bit-for-bit
 * restatement of glibc 2.35's kernels that libavp_hip.so compiles for the device. The two modes must agree everywhere
 * (tests/test_oracle_restated.py runs every trace fixture and the BASELINE workloads in both); the switch exists so that
 * a disagreement between the restatement and the libm the goldens were captured with shows up on the CPU, by name. */
#include "../include/avp_libm.h"
static int g_restated = 0;
ORC_API void orc_set_restated_libm(int v) { g_restated = v; }
ORC_API int orc_get_restated_libm(void) { return g_restated; }
/* What-if switch for the heuristic Dijkstra (test instrumentation; default 0 = the reference's behaviour): the
 * reference lowers the distance of an open cell IN PLACE without restoring the heap order (compute_h.py:226-227), so a
 * cell is occasionally popped one step early. With g_dij_reheap = 1 the decrease is followed by the sift-up a correct
 * decrease-key performs, i.e. cells are popped in exact (distance, id) order -- the order the device's bucketed sweep
 * realises. tests/test_dijkstra_stale_key.py measures what the two orders can differ in (nothing but the hit/miss
 * classification of a few later queries; never a distance, never a pop trace). */
static int g_dij_reheap = 0;
ORC_API void orc_set_dij_reheap(int v) { g_dij_reheap = v; }
ORC_API int orc_get_dij_reheap(void) { return g_dij_reheap; }
#define ATAN2(y, x) (g_restated ? avp_atan2((y), (x)) : atan2((y), (x)))
#define ASIN(x) (g_restated ? avp_asin(x) : asin(x))
#define ACOS(x) (g_restated ? avp_acos(x) : acos(x))
#define TAN(x) (g_restated ? avp_tan(x) : tan(x))
#define POW2(v) (g_restated ? avp_pow2(v) : pow((v), 2.0))

/* ------------------------------------------------------------------------------------------ */
/* context: map + vehicle + config                                                            */
typedef struct {
    int32_t nx, ny;          /* cost_map.shape                         map/costmap.py:182-186 */
    int32_t S, Sy;           /* int((b1-b0)/dx), int((b3-b2)/dy)       map/costmap.py:328     */
    double b[4];             /* boundary = floor(xmin,xmax,ymin,ymax)  map/costmap.py:169-172 */
    double dx, dy;           /* _discrete_x/_y                         map/costmap.py:190-191 */
    const double *X, *Y;     /* map_position                           map/costmap.py:188-193 */
    const uint8_t *occ;      /* cost_map==255, [ix*ny+iy]                                      */
    int32_t P;               /* obstacle points in np.where (row-major) order                  */
    const double *ox, *oy;
    /* vehicle: map/costmap.py:52-63 */
    double lw, lf, lr, lb, max_v, max_steer, min_radius;
    /* config: config/config.yaml */
    double safe_side, safe_fr;
    int32_t n_steer;
    double steer[64];        /* np.linspace(-max_steer, max_steer, n)  hybrid_a_star.py:81-83 (the reference takes any n; 64 here) */
    double steer_tan[64];    /* np.tan(steer[i]) computed on the host in numpy                 */
    double dt, ddt, flag_radius;
    double cost_gear, cost_heading, cost_scale;
    int32_t extended_num;
    int32_t checker_kind;    /* 0 distance, 1 circle */
    int64_t max_pops;        /* safety cap (the reference has none) */
} orc_ctx;

static const double PI = 3.141592653589793;

/* ---- Python float semantics ------------------------------------------------------------- */
static double py_fmod(double vx, double wx)
{   /* CPython float_rem */
    double mod = fmod(vx, wx);
    if (mod) {
        if ((wx < 0) != (mod < 0)) mod += wx;
    } else {
        mod = copysign(0.0, wx);
    }
    return mod;
}

static double py_hypot(double a, double b)
{   /* CPython 3.10 math.hypot -> vector_norm(n=2) */
    const double T27 = 134217729.0;
    double vec[2], max = 0.0, x, scale, oldcsum, csum = 1.0, frac1 = 0.0, frac2 = 0.0, frac3 = 0.0;
    double t, hi, lo, h;
    int max_e, i, found_nan = 0;
    vec[0] = fabs(a); vec[1] = fabs(b);
    for (i = 0; i < 2; i++) { found_nan |= isnan(vec[i]); if (vec[i] > max) max = vec[i]; }
    if (isinf(vec[0]) || isinf(vec[1])) return INFINITY;
    if (found_nan) return NAN;
    if (max == 0.0) return max;
    frexp(max, &max_e);
    if (max_e >= -1023) {
        scale = ldexp(1.0, -max_e);
        for (i = 0; i < 2; i++) {
            x = vec[i];
            x *= scale;
            t = x * T27; hi = t - (t - x); lo = x - hi;
            x = hi * hi; oldcsum = csum; csum += x; frac1 += (oldcsum - csum) + x;
            x = 2.0 * hi * lo; oldcsum = csum; csum += x; frac2 += (oldcsum - csum) + x;
            frac3 += lo * lo;
        }
        h = sqrt(csum - 1.0 + (frac1 + frac2 + frac3));
        x = h; t = x * T27; hi = t - (t - x); lo = x - hi;
        x = -hi * hi; oldcsum = csum; csum += x; frac1 += (oldcsum - csum) + x;
        x = -2.0 * hi * lo; oldcsum = csum; csum += x; frac2 += (oldcsum - csum) + x;
        x = -lo * lo; oldcsum = csum; csum += x; frac3 += (oldcsum - csum) + x;
        x = csum - 1.0 + (frac1 + frac2 + frac3);
        return (h + x / (2.0 * h)) / scale;
    }
    /* subnormal inputs: not reachable on this path */
    return sqrt(a * a + b * b);
}

/* path_plan/rs_curve.py:649-656 */
ORC_API double orc_pi_2_pi(double theta)
{
    while (theta > PI) theta -= 2.0 * PI;
    while (theta < -PI) theta += 2.0 * PI;
    return theta;
}

/* path_plan/rs_curve.py:669-680 */
ORC_API double orc_M(double theta)
{
    double phi = py_fmod(theta, 2.0 * PI);
    if (phi < -PI) phi += 2.0 * PI;
    if (phi > PI) phi -= 2.0 * PI;
    return phi;
}

/* ---- map index maths --------------------------------------------------------------------- */
/* map/costmap.py:319-329 */
ORC_API int64_t orc_pos_to_index(const orc_ctx *c, double gx, double gy)
{
    int64_t i0 = (int64_t)floor((gx - c->b[0]) / c->dx);
    int64_t i1 = (int64_t)floor((c->b[3] - gy) / c->dy) * (int64_t)c->S;
    return i0 + i1;
}

/* path_plan/compute_h.py:237-255 (Python negative indices wrap) */
ORC_API int32_t orc_is_obstacle(const orc_ctx *c, double gx, double gy)
{
    int64_t xi = (int64_t)floor((gx - c->b[0]) / c->dx) - 1;
    int64_t yi = (int64_t)floor((gy - c->b[2]) / c->dy) - 1;
    if (xi >= c->S) xi = c->S - 1;
    if (yi >= c->Sy) yi = c->Sy - 1;
    if (xi < 0) xi += c->nx;
    if (yi < 0) yi += c->ny;
    if (xi < 0 || yi < 0 || xi >= c->nx || yi >= c->ny) return -1; /* IndexError in the reference */
    return c->occ[xi * c->ny + yi] == 255;
}

/* ---- vehicle footprint: map/costmap.py:85-121 --------------------------------------------- */
/* out: rr, rf, lf, lr as (x,y) pairs. The 2x2 dot is evaluated as numpy/OpenBLAS does it in the
 * build container: acc = a0*b0; acc = fma(a1, b1, acc). */
ORC_API void orc_corners(const orc_ctx *c, double x, double y, double th, double out[8])
{
    double cs = cos(th), sn = sin(th);
    double lxr = -c->lr - c->safe_fr, lxf = c->lw + c->lf + c->safe_fr;
    double lyr = -c->lb / 2 - c->safe_side, lyl = c->lb / 2 + c->safe_side;
    const double lx[4] = { lxr, lxf, lxf, lxr };
    const double ly[4] = { lyr, lyr, lyl, lyl };
    for (int i = 0; i < 4; i++) {
        out[2 * i] = fma(-sn, ly[i], cs * lx[i]) + x;
        out[2 * i + 1] = fma(cs, ly[i], sn * lx[i]) + y;
    }
}

/* collision_check/collision_check.py:144-240 (get_near_obstacles :29-73 inlined) */
ORC_API int32_t orc_check_distance(const orc_ctx *c, double x, double y, double th, int32_t *near_out)
{
    double vb[8];
    orc_corners(c, x, y, th, vb);
    double xmax = vb[0], xmin = vb[0], ymax = vb[1], ymin = vb[1];
    for (int i = 1; i < 4; i++) {
        if (vb[2 * i] > xmax) xmax = vb[2 * i];
        if (vb[2 * i] < xmin) xmin = vb[2 * i];
        if (vb[2 * i + 1] > ymax) ymax = vb[2 * i + 1];
        if (vb[2 * i + 1] < ymin) ymin = vb[2 * i + 1];
    }
    double t0 = vb[0] - vb[6], t1 = vb[1] - vb[7];
    double v_lb = sqrt(t0 * t0 + t1 * t1);
    t0 = vb[6] - vb[4]; t1 = vb[7] - vb[5];
    double v_len = sqrt(t0 * t0 + t1 * t1);
    double k[4], bb[4];
    for (int i = 0; i < 4; i++) {
        int j = (i + 1) & 3;
        k[i] = (vb[2 * j + 1] - vb[2 * i + 1]) / (vb[2 * j] - vb[2 * i]);
        bb[i] = vb[2 * i + 1] - k[i] * vb[2 * i];
    }
    double den[4];
    for (int i = 0; i < 4; i++) den[i] = sqrt(1 + k[i] * k[i]);
    int32_t near = 0, hit = 0;
    for (int32_t p = 0; p < c->P; p++) {
        double px = c->ox[p], py = c->oy[p];
        if (!(px >= xmin && px <= xmax)) continue;
        if (!(py >= ymin && py <= ymax)) continue;
        near++;
        if (hit) continue;
        double d[4];
        for (int i = 0; i < 4; i++) d[i] = fabs(k[i] * px + bb[i] - py) / den[i];
        int c1 = fabs(d[0] - d[2]) < v_lb - 0.01;
        int c2 = fabs(d[1] - d[3]) < v_len - 0.01;
        if (c1 && c2) { hit = 1; continue; }
        int on_x = 0, on_y = 0;
        for (int i = 0; i < 4; i++) if (px == vb[2 * i]) on_x = 1;
        if (on_x) for (int i = 0; i < 4; i++) if (py == vb[2 * i + 1]) on_y = 1;
        if (on_x && on_y) { hit = 1; continue; }
        for (int i = 0; i < 4; i++) {
            double k1 = (vb[2 * i + 1] - py) / (vb[2 * i] - px);
            if (k1 == k[i]) { hit = 1; break; }
        }
    }
    if (near_out) *near_out = near;
    return hit;
}

/* collision_check/collision_check.py:88-137 */
ORC_API int32_t orc_check_circle(const orc_ctx *c, double x, double y, double th)
{
    double Rd = 0.5 * sqrt(pow((c->lr + c->lw + c->lf) / 2, 2.0) + pow(c->lb, 2.0));
    double cf = 1.0 / 4 * (3 * c->lw + 3 * c->lf - c->lr);
    double cr = 1.0 / 4 * (c->lw + c->lf - 3 * c->lr);
    double cs = cos(th), sn = sin(th);
    double fx = x + cf * cs, fy = y + cf * sn, rx = x + cr * cs, ry = y + cr * sn;
    double right, left, upper, down;
    if (fx >= rx) { right = fx + Rd; left = rx - Rd; } else { right = rx + Rd; left = fx - Rd; }
    if (fy >= ry) { upper = fy + Rd; down = ry - Rd; } else { upper = ry + Rd; down = fy - Rd; }
    int32_t hit = 0;
    for (int32_t p = 0; p < c->P; p++) {
        double px = c->ox[p], py = c->oy[p];
        if (!(px > left && px < right)) continue;
        if (!(py > down && py < upper)) continue;
        if (sqrt(POW2(px - fx) + POW2(py - fy)) <= Rd) { hit = 1; break; }
        else if (sqrt(POW2(px - rx) + POW2(py - ry)) <= Rd) { hit = 1; break; }
    }
    return hit;
}

static int32_t orc_check(const orc_ctx *c, double x, double y, double th)
{
    return c->checker_kind == 1 ? orc_check_circle(c, x, y, th) : orc_check_distance(c, x, y, th, NULL);
}

ORC_API void orc_check_batch(const orc_ctx *c, int32_t kind, const double *x, const double *y, const double *th,
                             int64_t n, uint8_t *out, int32_t *near)
{
    for (int64_t i = 0; i < n; i++) {
        int32_t nn = 0;
        out[i] = kind == 1 ? orc_check_circle(c, x[i], y[i], th[i]) : orc_check_distance(c, x[i], y[i], th[i], &nn);
        if (near) near[i] = nn;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Reeds-Shepp: path_plan/rs_curve.py                                                         */
enum { T_S = 0, T_L = 1, T_R = 2 };
#define RS_MAXSEG 5
#define RS_MAXCAND 46
typedef struct { int n; int8_t t[RS_MAXSEG]; double l[RS_MAXSEG]; double L; } rs_cand;
typedef struct { rs_cand c[RS_MAXCAND]; int n; int err; } rs_set;

/* rs_curve.py:137-156 */
static void set_path(rs_set *s, int n, const double *len, const int8_t *ty)
{
    for (int e = 0; e < s->n; e++) {
        const rs_cand *pe = &s->c[e];
        if (pe->n != n) continue;
        int same = 1;
        for (int i = 0; i < n; i++) if (pe->t[i] != ty[i]) same = 0;
        if (!same) continue;
        double sum = 0;
        for (int i = 0; i < n; i++) sum = sum + (pe->l[i] - len[i]);
        if (sum <= 0.01) return;
    }
    double L = 0;
    for (int i = 0; i < n; i++) L = L + fabs(len[i]);
    if (L >= 1000.0) return;
    if (!(L >= 0.01)) { s->err = 1; return; }   /* AssertionError rs_curve.py:153 */
    rs_cand *p = &s->c[s->n++];
    p->n = n; p->L = L;
    for (int i = 0; i < n; i++) { p->t[i] = ty[i]; p->l[i] = len[i]; }
}

static void polar(double x, double y, double *r, double *th) { *r = py_hypot(x, y); *th = ATAN2(y, x); }

/* rs_curve.py:159-167 */
static int LSL(double x, double y, double phi, double *t, double *u, double *v)
{
    double uu, tt;
    polar(x - sin(phi), y - 1.0 + cos(phi), &uu, &tt);
    if (tt >= 0.0) {
        double vv = orc_M(phi - tt);
        if (vv >= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
    }
    return 0;
}
/* rs_curve.py:170-183 */
static int LSR(double x, double y, double phi, double *t, double *u, double *v)
{
    double u1, t1;
    polar(x + sin(phi), y - 1.0 - cos(phi), &u1, &t1);
    u1 = POW2(u1);
    if (u1 >= 4.0) {
        double uu = sqrt(u1 - 4.0);
        double theta = ATAN2(2.0, uu);
        double tt = orc_M(t1 + theta);
        double vv = orc_M(tt - phi);
        if (tt >= 0.0 && vv >= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
    }
    return 0;
}
/* rs_curve.py:186-197 */
static int LRL(double x, double y, double phi, double *t, double *u, double *v)
{
    double u1, t1;
    polar(x - sin(phi), y - 1.0 + cos(phi), &u1, &t1);
    if (u1 <= 4.0) {
        double uu = -2.0 * ASIN(0.25 * u1);
        double tt = orc_M(t1 + 0.5 * uu + PI);
        double vv = orc_M(phi - tt + uu);
        if (tt >= 0.0 && uu <= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
    }
    return 0;
}
/* rs_curve.py:213-229 */
static int SLS(double x, double y, double phi, double *t, double *u, double *v)
{
    phi = orc_M(phi);
    if (y > 0.0 && 0.0 < phi && phi < PI * 0.99) {
        double xd = -y / TAN(phi) + x;
        *t = xd - TAN(phi / 2.0);
        *u = phi;
        *v = sqrt(POW2(x - xd) + POW2(y)) - TAN(phi / 2.0);
        return 1;
    } else if (y < 0.0 && 0.0 < phi && phi < PI * 0.99) {
        double xd = -y / TAN(phi) + x;
        *t = xd - TAN(phi / 2.0);
        *u = phi;
        *v = -sqrt(POW2(x - xd) + POW2(y)) - TAN(phi / 2.0);
        return 1;
    }
    return 0;
}
/* rs_curve.py:308-323 */
static void calc_tauOmega(double u, double v, double xi, double eta, double phi, double *tau, double *omega)
{
    double delta = orc_M(u - v);
    double A = sin(u) - sin(delta);
    double B = cos(u) - cos(delta) - 1.0;
    double t1 = ATAN2(eta * A - xi * B, xi * A + eta * B);
    double t2 = 2.0 * (cos(delta) - cos(v) - cos(u)) + 3.0;
    if (t2 < 0) *tau = orc_M(t1 + PI); else *tau = orc_M(t1);
    *omega = orc_M(*tau - u + v - phi);
}
/* rs_curve.py:326-337 */
static int LRLRn(double x, double y, double phi, double *t, double *u, double *v)
{
    double xi = x + sin(phi), eta = y - 1.0 - cos(phi);
    double rho = 0.25 * (2.0 + sqrt(xi * xi + eta * eta));
    if (rho <= 1.0) {
        double uu = ACOS(rho), tt, vv;
        calc_tauOmega(uu, -uu, xi, eta, phi, &tt, &vv);
        if (tt >= 0.0 && vv <= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
    }
    return 0;
}
/* rs_curve.py:340-352 */
static int LRLRp(double x, double y, double phi, double *t, double *u, double *v)
{
    double xi = x + sin(phi), eta = y - 1.0 - cos(phi);
    double rho = (20.0 - xi * xi - eta * eta) / 16.0;
    if (0.0 <= rho && rho <= 1.0) {
        double uu = -ACOS(rho);
        if (uu >= -0.5 * PI) {
            double tt, vv;
            calc_tauOmega(uu, uu, xi, eta, phi, &tt, &vv);
            if (tt >= 0.0 && vv >= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
        }
    }
    return 0;
}
/* rs_curve.py:391-403 */
static int LRSR(double x, double y, double phi, double *t, double *u, double *v)
{
    double xi = x + sin(phi), eta = y - 1.0 - cos(phi), rho, theta;
    polar(-eta, xi, &rho, &theta);
    if (rho >= 2.0) {
        double tt = theta, uu = 2.0 - rho, vv = orc_M(tt + 0.5 * PI - phi);
        if (tt >= 0.0 && uu <= 0.0 && vv <= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
    }
    return 0;
}
/* rs_curve.py:406-419 */
static int LRSL(double x, double y, double phi, double *t, double *u, double *v)
{
    double xi = x - sin(phi), eta = y - 1.0 + cos(phi), rho, theta;
    polar(xi, eta, &rho, &theta);
    if (rho >= 2.0) {
        double r = sqrt(rho * rho - 4.0);
        double uu = 2.0 - r;
        double tt = orc_M(theta + ATAN2(r, -2.0));
        double vv = orc_M(phi - 0.5 * PI - tt);
        if (tt >= 0.0 && uu <= 0.0 && vv <= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
    }
    return 0;
}
/* rs_curve.py:494-510 */
static int LRSLR(double x, double y, double phi, double *t, double *u, double *v)
{
    double xi = x + sin(phi), eta = y - 1.0 - cos(phi), rho, theta;
    polar(xi, eta, &rho, &theta);
    if (rho >= 2.0) {
        double uu = 4.0 - sqrt(rho * rho - 4.0);
        if (uu <= 0.0) {
            double tt = orc_M(ATAN2((4.0 - uu) * xi - 2.0 * eta, -2.0 * xi + (uu - 4.0) * eta));
            double vv = orc_M(tt - phi);
            if (tt >= 0.0 && vv >= 0.0) { *t = tt; *u = uu; *v = vv; return 1; }
        }
    }
    return 0;
}

#define SETP(n, ...) do { const double L_[] = { __VA_ARGS__ }; set_path(s, n, L_, ty); } while (0)
#define TY(...) do { const int8_t T_[] = { __VA_ARGS__ }; memcpy(ty, T_, sizeof(T_)); } while (0)

/* rs_curve.py:627-644 with SCS :200, CSC :232, CCC :268, CCCC :355, CCSC :422, CCSCC :513 */
static void rs_generate(const double q0[3], const double q1[3], double maxc, rs_set *s)
{
    double dx = q1[0] - q0[0], dy = q1[1] - q0[1], phi = q1[2] - q0[2];
    double c = cos(q0[2]), sn = sin(q0[2]);
    double x = (c * dx + sn * dy) * maxc;
    double y = (-sn * dx + c * dy) * maxc;
    double t, u, v;
    int8_t ty[RS_MAXSEG];
    s->n = 0; s->err = 0;
    const double hp = 0.5 * PI;
    /* SCS */
    if (SLS(x, y, phi, &t, &u, &v)) { TY(T_S, T_L, T_S); SETP(3, t, u, v); }
    if (SLS(x, -y, -phi, &t, &u, &v)) { TY(T_S, T_R, T_S); SETP(3, t, u, v); }
    /* CSC */
    if (LSL(x, y, phi, &t, &u, &v)) { TY(T_L, T_S, T_L); SETP(3, t, u, v); }
    if (LSL(-x, y, -phi, &t, &u, &v)) { TY(T_L, T_S, T_L); SETP(3, -t, -u, -v); }
    if (LSL(x, -y, -phi, &t, &u, &v)) { TY(T_R, T_S, T_R); SETP(3, t, u, v); }
    if (LSL(-x, -y, phi, &t, &u, &v)) { TY(T_R, T_S, T_R); SETP(3, -t, -u, -v); }
    if (LSR(x, y, phi, &t, &u, &v)) { TY(T_L, T_S, T_R); SETP(3, t, u, v); }
    if (LSR(-x, y, -phi, &t, &u, &v)) { TY(T_L, T_S, T_R); SETP(3, -t, -u, -v); }
    if (LSR(x, -y, -phi, &t, &u, &v)) { TY(T_R, T_S, T_L); SETP(3, t, u, v); }
    if (LSR(-x, -y, phi, &t, &u, &v)) { TY(T_R, T_S, T_L); SETP(3, -t, -u, -v); }
    /* CCC */
    if (LRL(x, y, phi, &t, &u, &v)) { TY(T_L, T_R, T_L); SETP(3, t, u, v); }
    if (LRL(-x, y, -phi, &t, &u, &v)) { TY(T_L, T_R, T_L); SETP(3, -t, -u, -v); }
    if (LRL(x, -y, -phi, &t, &u, &v)) { TY(T_R, T_L, T_R); SETP(3, t, u, v); }
    if (LRL(-x, -y, phi, &t, &u, &v)) { TY(T_R, T_L, T_R); SETP(3, -t, -u, -v); }
    {
        double xb = x * cos(phi) + y * sin(phi);
        double yb = x * sin(phi) - y * cos(phi);
        if (LRL(xb, yb, phi, &t, &u, &v)) { TY(T_L, T_R, T_L); SETP(3, v, u, t); }
        if (LRL(-xb, yb, -phi, &t, &u, &v)) { TY(T_L, T_R, T_L); SETP(3, -v, -u, -t); }
        if (LRL(xb, -yb, -phi, &t, &u, &v)) { TY(T_R, T_L, T_R); SETP(3, v, u, t); }
        if (LRL(-xb, -yb, phi, &t, &u, &v)) { TY(T_R, T_L, T_R); SETP(3, -v, -u, -t); }
    }
    /* CCCC */
    if (LRLRn(x, y, phi, &t, &u, &v)) { TY(T_L, T_R, T_L, T_R); SETP(4, t, u, -u, v); }
    if (LRLRn(-x, y, -phi, &t, &u, &v)) { TY(T_L, T_R, T_L, T_R); SETP(4, -t, -u, u, -v); }
    if (LRLRn(x, -y, -phi, &t, &u, &v)) { TY(T_R, T_L, T_R, T_L); SETP(4, t, u, -u, v); }
    if (LRLRn(-x, -y, phi, &t, &u, &v)) { TY(T_R, T_L, T_R, T_L); SETP(4, -t, -u, u, -v); }
    if (LRLRp(x, y, phi, &t, &u, &v)) { TY(T_L, T_R, T_L, T_R); SETP(4, t, u, u, v); }
    if (LRLRp(-x, y, -phi, &t, &u, &v)) { TY(T_L, T_R, T_L, T_R); SETP(4, -t, -u, -u, -v); }
    if (LRLRp(x, -y, -phi, &t, &u, &v)) { TY(T_R, T_L, T_R, T_L); SETP(4, t, u, u, v); }
    if (LRLRp(-x, -y, phi, &t, &u, &v)) { TY(T_R, T_L, T_R, T_L); SETP(4, -t, -u, -u, -v); }
    /* CCSC */
    if (LRSL(x, y, phi, &t, &u, &v)) { TY(T_L, T_R, T_S, T_L); SETP(4, t, -hp, u, v); }
    if (LRSL(-x, y, -phi, &t, &u, &v)) { TY(T_L, T_R, T_S, T_L); SETP(4, -t, hp, -u, -v); }
    if (LRSL(x, -y, -phi, &t, &u, &v)) { TY(T_R, T_L, T_S, T_R); SETP(4, t, -hp, u, v); }
    if (LRSL(-x, -y, phi, &t, &u, &v)) { TY(T_R, T_L, T_S, T_R); SETP(4, -t, hp, -u, -v); }
    if (LRSR(x, y, phi, &t, &u, &v)) { TY(T_L, T_R, T_S, T_R); SETP(4, t, -hp, u, v); }
    if (LRSR(-x, y, -phi, &t, &u, &v)) { TY(T_L, T_R, T_S, T_R); SETP(4, -t, hp, -u, -v); }
    if (LRSR(x, -y, -phi, &t, &u, &v)) { TY(T_R, T_L, T_S, T_L); SETP(4, t, -hp, u, v); }
    if (LRSR(-x, -y, phi, &t, &u, &v)) { TY(T_R, T_L, T_S, T_L); SETP(4, -t, hp, -u, -v); }
    {
        double xb = x * cos(phi) + y * sin(phi);
        double yb = x * sin(phi) - y * cos(phi);
        if (LRSL(xb, yb, phi, &t, &u, &v)) { TY(T_L, T_S, T_R, T_L); SETP(4, v, u, -hp, t); }
        if (LRSL(-xb, yb, -phi, &t, &u, &v)) { TY(T_L, T_S, T_R, T_L); SETP(4, -v, -u, hp, -t); }
        if (LRSL(xb, -yb, -phi, &t, &u, &v)) { TY(T_R, T_S, T_L, T_R); SETP(4, v, u, -hp, t); }
        if (LRSL(-xb, -yb, phi, &t, &u, &v)) { TY(T_R, T_S, T_L, T_R); SETP(4, -v, -u, hp, -t); }
        if (LRSR(xb, yb, phi, &t, &u, &v)) { TY(T_R, T_S, T_R, T_L); SETP(4, v, u, -hp, t); }
        if (LRSR(-xb, yb, -phi, &t, &u, &v)) { TY(T_R, T_S, T_R, T_L); SETP(4, -v, -u, hp, -t); }
        if (LRSR(xb, -yb, -phi, &t, &u, &v)) { TY(T_L, T_S, T_L, T_R); SETP(4, v, u, -hp, t); }
        if (LRSR(-xb, -yb, phi, &t, &u, &v)) { TY(T_L, T_S, T_L, T_R); SETP(4, -v, -u, hp, -t); }
    }
    /* CCSCC */
    if (LRSLR(x, y, phi, &t, &u, &v)) { TY(T_L, T_R, T_S, T_L, T_R); SETP(5, t, -hp, u, -hp, v); }
    if (LRSLR(-x, y, -phi, &t, &u, &v)) { TY(T_L, T_R, T_S, T_L, T_R); SETP(5, -t, hp, -u, hp, -v); }
    if (LRSLR(x, -y, -phi, &t, &u, &v)) { TY(T_R, T_L, T_S, T_R, T_L); SETP(5, t, -hp, u, -hp, v); }
    if (LRSLR(-x, -y, phi, &t, &u, &v)) { TY(T_R, T_L, T_S, T_R, T_L); SETP(5, -t, hp, -u, hp, -v); }
}

/* rs_curve.py:597-624 */
static void interpolate(int ind, double l, int m, double maxc, double ox, double oy, double oyaw,
                        double *px, double *py, double *pyaw, int8_t *dir)
{
    if (m == T_S) {
        px[ind] = ox + l / maxc * cos(oyaw);
        py[ind] = oy + l / maxc * sin(oyaw);
        pyaw[ind] = oyaw;
    } else {
        double ldx = sin(l) / maxc, ldy;
        if (m == T_L) ldy = (1.0 - cos(l)) / maxc; else ldy = (1.0 - cos(l)) / (-maxc);
        double gdx = cos(-oyaw) * ldx + sin(-oyaw) * ldy;
        double gdy = -sin(-oyaw) * ldx + cos(-oyaw) * ldy;
        px[ind] = ox + gdx;
        py[ind] = oy + gdy;
    }
    if (m == T_L) pyaw[ind] = oyaw + l; else if (m == T_R) pyaw[ind] = oyaw - l;
    dir[ind] = l > 0.0 ? 1 : -1;
}

/* rs_curve.py:537-594; returns number of points kept, or -1 if cap too small */
static int local_course(const rs_cand *p, double maxc, double step, double *px, double *py, double *pyaw,
                        int8_t *dir, int cap)
{
    int point_num = (int)(p->L / step) + p->n + 3;
    if (point_num > cap) return -1;
    for (int i = 0; i < point_num; i++) { px[i] = 0.0; py[i] = 0.0; pyaw[i] = 0.0; dir[i] = 0; }
    int ind = 1;
    dir[0] = p->l[0] > 0.0 ? 1 : -1;
    double d = p->l[0] > 0.0 ? step : -step;
    double pd = d, ll = 0.0;
    for (int i = 0; i < p->n; i++) {
        double l = p->l[i];
        int m = p->t[i];
        d = l > 0.0 ? step : -step;
        double ox = px[ind], oy = py[ind], oyaw = pyaw[ind];
        ind -= 1;
        if (i >= 1 && (p->l[i - 1] * p->l[i]) > 0) pd = -d - ll; else pd = d - ll;
        while (fabs(pd) <= fabs(l)) {
            ind += 1;
            interpolate(ind, pd, m, maxc, ox, oy, oyaw, px, py, pyaw, dir);
            pd += d;
        }
        ll = l - pd - d;
        ind += 1;
        interpolate(ind, l, m, maxc, ox, oy, oyaw, px, py, pyaw, dir);
    }
    while (point_num > 0 && px[point_num - 1] == 0.0) point_num--;
    return point_num;
}

#define RS_MAXPTS 4096
typedef struct {
    int n; int8_t t[RS_MAXSEG]; double l[RS_MAXSEG]; double L;  /* metres (divided by maxc) */
    int npts; double *x, *y, *yaw; int8_t *dir;                 /* world frame               */
} rs_path;

/* rs_curve.py:99-134: every candidate is sampled in the reference; sampling only the winner gives
 * the same returned PATH (candidates are independent). status: 0 ok, 1 no candidate, 2 assertion */
static int rs_optimal(const double q0[3], const double q1[3], double maxc, rs_path *out, int cap)
{
    rs_set s;
    rs_generate(q0, q1, maxc, &s);
    if (s.err) return 2;
    if (s.n == 0) return 1;     /* IndexError paths[0] in the reference */
    double minL = s.c[0].L / maxc;
    int mini = 0;
    for (int i = 0; i < s.n; i++) {
        double Li = s.c[i].L / maxc;
        if (Li <= minL) { minL = Li; mini = i; }
    }
    const rs_cand *p = &s.c[mini];
    out->n = p->n;
    for (int i = 0; i < p->n; i++) { out->t[i] = p->t[i]; out->l[i] = p->l[i] / maxc; }
    out->L = p->L / maxc;
    out->npts = 0;
    if (cap > 0) {
        int np = local_course(p, maxc, 0.5 * maxc, out->x, out->y, out->yaw, out->dir, cap);
        if (np < 0) return 3;
        double cm = cos(-q0[2]), sm = sin(-q0[2]);
        for (int i = 0; i < np; i++) {
            double ix = out->x[i], iy = out->y[i];
            out->x[i] = cm * ix + sm * iy + q0[0];
            out->y[i] = -sm * ix + cm * iy + q0[1];
            out->yaw[i] = orc_pi_2_pi(out->yaw[i] + q0[2]);
        }
        out->npts = np;
    }
    return 0;
}

/* batch interface used by the tests: candidates (generate_path) and optimum (calc_optimal_path) */
ORC_API void orc_rs_candidates(const double *q0, const double *q1, double maxc, int64_t n, int32_t *ncand,
                               int8_t *types /* n*12*5 */, double *lens /* n*12*5 */)
{
    for (int64_t i = 0; i < n; i++) {
        rs_set s;
        rs_generate(q0 + 3 * i, q1 + 3 * i, maxc, &s);
        ncand[i] = s.err ? -1 : s.n;
        for (int j = 0; j < 12; j++)
            for (int k = 0; k < 5; k++) {
                types[(i * 12 + j) * 5 + k] = (j < s.n && k < s.c[j].n) ? s.c[j].t[k] : -1;
                lens[(i * 12 + j) * 5 + k] = (j < s.n && k < s.c[j].n) ? s.c[j].l[k] : 0.0;
            }
    }
}

ORC_API void orc_rs_optimal_batch(const double *q0, const double *q1, double maxc, int64_t n, int32_t maxpts,
                                  int32_t *status, double *L, int8_t *types /* n*5 */, double *lens /* n*5 */,
                                  int32_t *npts, double *xyyaw /* n*maxpts*3 */, int8_t *dir /* n*maxpts */)
{
    double *bx = malloc(sizeof(double) * RS_MAXPTS * 3);
    int8_t *bd = malloc(RS_MAXPTS);
    for (int64_t i = 0; i < n; i++) {
        rs_path p;
        p.x = bx; p.y = bx + RS_MAXPTS; p.yaw = bx + 2 * RS_MAXPTS; p.dir = bd;
        int st = rs_optimal(q0 + 3 * i, q1 + 3 * i, maxc, &p, RS_MAXPTS);
        status[i] = st;
        for (int k = 0; k < 5; k++) { types[i * 5 + k] = -1; lens[i * 5 + k] = 0.0; }
        L[i] = 0; npts[i] = 0;
        if (st) continue;
        L[i] = p.L;
        for (int k = 0; k < p.n; k++) { types[i * 5 + k] = p.t[k]; lens[i * 5 + k] = p.l[k]; }
        npts[i] = p.npts;
        if (p.npts > maxpts) { status[i] = 3; continue; }
        for (int k = 0; k < p.npts; k++) {
            xyyaw[(i * maxpts + k) * 3 + 0] = p.x[k];
            xyyaw[(i * maxpts + k) * 3 + 1] = p.y[k];
            xyyaw[(i * maxpts + k) * 3 + 2] = p.yaw[k];
            dir[i * maxpts + k] = p.dir[k];
        }
    }
    free(bx); free(bd);
}

/* ------------------------------------------------------------------------------------------ */
/* int64 -> int32 open-addressing hash map                                                    */
typedef struct { int64_t *k; int32_t *v; int64_t cap, n; } imap;
static void imap_init(imap *m, int64_t cap) { m->cap = 1; while (m->cap < cap) m->cap <<= 1;
    m->k = malloc(sizeof(int64_t) * m->cap); m->v = malloc(sizeof(int32_t) * m->cap); m->n = 0;
    for (int64_t i = 0; i < m->cap; i++) m->v[i] = -1; }
static void imap_free(imap *m) { free(m->k); free(m->v); }
static uint64_t mix64(uint64_t z) { z ^= z >> 33; z *= 0xff51afd7ed558ccdULL; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ULL; z ^= z >> 33; return z; }
static int32_t *imap_slot(imap *m, int64_t key, int create);
static void imap_grow(imap *m)
{
    imap o = *m;
    imap_init(m, o.cap * 2);
    for (int64_t i = 0; i < o.cap; i++) if (o.v[i] != -1) *imap_slot(m, o.k[i], 1) = o.v[i];
    imap_free(&o);
}
static int32_t *imap_slot(imap *m, int64_t key, int create)
{
    if (create && m->n * 2 >= m->cap) imap_grow(m);
    uint64_t h = mix64((uint64_t)key) & (uint64_t)(m->cap - 1);
    for (;;) {
        if (m->v[h] == -1) {
            if (!create) return NULL;
            m->k[h] = key; m->v[h] = -2; m->n++;
            return &m->v[h];
        }
        if (m->k[h] == key) return &m->v[h];
        h = (h + 1) & (uint64_t)(m->cap - 1);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Dijkstra heuristic field: path_plan/compute_h.py                                           */
typedef struct { int64_t id; double x, y; int64_t dist; int64_t father; int in_heap; } grid_t;
typedef struct {
    const orc_ctx *c;
    double gx, gy;                   /* final_point */
    grid_t *pool; int64_t npool, cpool;
    int32_t *heap; int64_t nheap, cheap;       /* open_list.queue (indices into pool)  */
    int32_t *closed; int64_t nclosed, cclosed; /* closedlist (indices into pool)       */
    imap seen;                       /* openlist_index: id -> pool idx of the pushed Grid */
    imap first_closed;               /* id -> first position in closedlist               */
    int64_t terminate_id;
    int64_t n_pops, n_calls;
} dij_t;

static int grid_lt(const grid_t *a, const grid_t *b)
{   /* compute_h.py:33-38 */
    if (a->dist == b->dist) return a->id < b->id;
    return a->dist < b->dist;
}
/* CPython heapq._siftdown / _siftup / heappush / heappop */
static void dheap_siftdown(dij_t *d, int64_t startpos, int64_t pos)
{
    int32_t newitem = d->heap[pos];
    while (pos > startpos) {
        int64_t parentpos = (pos - 1) >> 1;
        int32_t parent = d->heap[parentpos];
        if (grid_lt(&d->pool[newitem], &d->pool[parent])) { d->heap[pos] = parent; pos = parentpos; continue; }
        break;
    }
    d->heap[pos] = newitem;
}
static void dheap_siftup(dij_t *d, int64_t pos)
{
    int64_t endpos = d->nheap, startpos = pos;
    int32_t newitem = d->heap[pos];
    int64_t childpos = 2 * pos + 1;
    while (childpos < endpos) {
        int64_t rightpos = childpos + 1;
        if (rightpos < endpos && !grid_lt(&d->pool[d->heap[childpos]], &d->pool[d->heap[rightpos]])) childpos = rightpos;
        d->heap[pos] = d->heap[childpos];
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    d->heap[pos] = newitem;
    dheap_siftdown(d, startpos, pos);
}
static void dheap_push(dij_t *d, int32_t item)
{
    if (d->nheap == d->cheap) { d->cheap *= 2; d->heap = realloc(d->heap, sizeof(int32_t) * d->cheap); }
    d->heap[d->nheap++] = item;
    dheap_siftdown(d, 0, d->nheap - 1);
}
static int32_t dheap_pop(dij_t *d)
{
    int32_t lastelt = d->heap[--d->nheap];
    if (d->nheap) {
        int32_t ret = d->heap[0];
        d->heap[0] = lastelt;
        dheap_siftup(d, 0);
        return ret;
    }
    return lastelt;
}
static int32_t dij_new_grid(dij_t *d, int64_t id, double x, double y, int64_t dist, int64_t father)
{
    if (d->npool == d->cpool) { d->cpool *= 2; d->pool = realloc(d->pool, sizeof(grid_t) * d->cpool); }
    grid_t *g = &d->pool[d->npool];
    g->id = id; g->x = x; g->y = y; g->dist = dist; g->father = father; g->in_heap = 0;
    return (int32_t)d->npool++;
}
static void dij_close(dij_t *d, int32_t gi)
{
    if (d->nclosed == d->cclosed) { d->cclosed *= 2; d->closed = realloc(d->closed, sizeof(int32_t) * d->cclosed); }
    int32_t *s = imap_slot(&d->first_closed, d->pool[gi].id, 1);
    if (*s == -2) *s = (int32_t)d->nclosed;
    d->closed[d->nclosed++] = gi;
}

ORC_API dij_t *orc_dij_create(const orc_ctx *c, double gx, double gy)
{   /* compute_h.py:42-48 */
    dij_t *d = calloc(1, sizeof(dij_t));
    d->c = c; d->gx = gx; d->gy = gy;
    d->cpool = 1 << 16; d->pool = malloc(sizeof(grid_t) * d->cpool);
    d->cheap = 1 << 12; d->heap = malloc(sizeof(int32_t) * d->cheap);
    d->cclosed = 1 << 16; d->closed = malloc(sizeof(int32_t) * d->cclosed);
    imap_init(&d->seen, 1 << 17);
    imap_init(&d->first_closed, 1 << 17);
    return d;
}
ORC_API void orc_dij_destroy(dij_t *d)
{
    if (!d) return;
    free(d->pool); free(d->heap); free(d->closed); imap_free(&d->seen); imap_free(&d->first_closed); free(d);
}

/* compute_h.py:216-235 */
static void dij_add(dij_t *d, double gx, double gy, int64_t priority, int64_t father_id)
{
    int64_t index = orc_pos_to_index(d->c, gx, gy);
    int32_t *s = imap_slot(&d->seen, index, 0);
    if (s) {
        grid_t *g = &d->pool[*s];
        if (g->in_heap && g->dist > priority) {
            g->dist = priority; g->father = father_id;
            if (g_dij_reheap) {
                for (int64_t q = 0; q < d->nheap; q++) if (d->heap[q] == *s) { dheap_siftdown(d, 0, q); break; }
            }
        }
    } else {
        int32_t gi = dij_new_grid(d, index, gx, gy, priority, father_id);
        d->pool[gi].in_heap = 1;
        dheap_push(d, gi);
        *imap_slot(&d->seen, index, 1) = gi;
    }
}

/* compute_h.py:84-195 */
static void dij_update_openlist(dij_t *d, int32_t cur)
{
    const orc_ctx *c = d->c;
    static const int sx[8] = { -1, 0, 1, -1, 1, -1, 0, 1 };
    static const int sy[8] = { 1, 1, 1, 0, 0, -1, -1, -1 };
    static const int cost[8] = { 14, 10, 14, 10, 10, 14, 10, 14 };
    for (int i = 0; i < 8; i++) {
        double cx = d->pool[cur].x, cy = d->pool[cur].y;
        double gx = sx[i] < 0 ? cx - c->dx : (sx[i] > 0 ? cx + c->dx : cx);
        double gy = sy[i] < 0 ? cy - c->dy : (sy[i] > 0 ? cy + c->dy : cy);
        if (orc_is_obstacle(c, gx, gy)) continue;
        int ok = 1;
        if (sx[i] < 0 && !(gx >= c->b[0])) ok = 0;
        if (sx[i] > 0 && !(gx <= c->b[1])) ok = 0;
        if (sy[i] > 0 && !(gy <= c->b[3])) ok = 0;
        if (sy[i] < 0 && !(gy >= c->b[2])) ok = 0;
        if (ok) dij_add(d, gx, gy, d->pool[cur].dist + cost[i], d->pool[cur].id);
    }
}

/* compute_h.py:198-214. returns distance, or -1 when the open list runs dry (the reference would
 * block forever inside PriorityQueue.get(), compute_h.py:77) */
ORC_API int64_t orc_dij_compute_path(dij_t *d, double node_x, double node_y)
{
    const orc_ctx *c = d->c;
    d->n_calls++;
    /* initial_map :50-72 */
    int64_t init_id = orc_pos_to_index(c, d->gx, d->gy);
    int32_t cur = dij_new_grid(d, init_id, d->gx, d->gy, 0, 0);
    dij_close(d, cur);
    d->terminate_id = orc_pos_to_index(c, node_x, node_y);
    int found = 0;
    while (!found) {
        dij_update_openlist(d, cur);
        if (d->nheap == 0) return -1;
        cur = dheap_pop(d);
        d->pool[cur].in_heap = 0;
        d->n_pops++;
        if (d->pool[cur].id == d->terminate_id) found = 1;
        dij_close(d, cur);
    }
    return d->pool[cur].dist;
}

/* hybrid_a_star.py:272-280: first closedlist entry with this id; -1 if absent */
ORC_API int64_t orc_dij_lookup(const dij_t *d, int64_t id)
{
    int32_t *s = imap_slot((imap *)&d->first_closed, id, 0);
    if (!s) return -1;
    return d->pool[d->closed[*s]].dist;
}
ORC_API int64_t orc_dij_nclosed(const dij_t *d) { return d->nclosed; }
ORC_API void orc_dij_dump(const dij_t *d, int64_t *ids, int64_t *dist, double *x, double *y)
{
    for (int64_t i = 0; i < d->nclosed; i++) {
        const grid_t *g = &d->pool[d->closed[i]];
        ids[i] = g->id; dist[i] = g->dist;
        if (x) x[i] = g->x;
        if (y) y[i] = g->y;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* hybrid A*: path_plan/hybrid_a_star.py + path_plan/path_planner.py                          */
typedef struct {
    int64_t index, parent_index;
    double x, y, theta, h, g, f;
    int8_t forward, in_open, in_closed, steer_i;  /* steer_i = -1 for the root (steering None) */
} node_t;

typedef struct { double x, y, th; } pose_key;
typedef struct { pose_key *k; int32_t *v; int64_t cap, n; } pmap;
static double nz(double v) { return v == 0.0 ? 0.0 : v; }
static uint64_t dbits(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
static uint64_t pose_hash(double x, double y, double th)
{ return mix64(dbits(nz(x)) * 0x9E3779B97F4A7C15ULL ^ mix64(dbits(nz(y)) + 0x632BE59BD9B4E019ULL) ^ mix64(dbits(nz(th)) * 3)); }
static void pmap_init(pmap *m, int64_t cap) { m->cap = 1; while (m->cap < cap) m->cap <<= 1;
    m->k = malloc(sizeof(pose_key) * m->cap); m->v = malloc(sizeof(int32_t) * m->cap); m->n = 0;
    for (int64_t i = 0; i < m->cap; i++) m->v[i] = -1; }
static void pmap_free(pmap *m) { free(m->k); free(m->v); }
static int32_t pmap_get(const pmap *m, double x, double y, double th)
{
    if (x != x || y != y || th != th) return -1;
    uint64_t h = pose_hash(x, y, th) & (uint64_t)(m->cap - 1);
    for (;;) {
        if (m->v[h] == -1) return -1;
        if (m->k[h].x == x && m->k[h].y == y && m->k[h].th == th) return m->v[h];
        h = (h + 1) & (uint64_t)(m->cap - 1);
    }
}
static void pmap_put(pmap *m, double x, double y, double th, int32_t v);
static void pmap_grow(pmap *m)
{
    pmap o = *m;
    pmap_init(m, o.cap * 2);
    for (int64_t i = 0; i < o.cap; i++) if (o.v[i] != -1) pmap_put(m, o.k[i].x, o.k[i].y, o.k[i].th, o.v[i]);
    pmap_free(&o);
}
static void pmap_put(pmap *m, double x, double y, double th, int32_t v)
{
    if (x != x || y != y || th != th) return;
    if (m->n * 2 >= m->cap) pmap_grow(m);
    uint64_t h = pose_hash(x, y, th) & (uint64_t)(m->cap - 1);
    while (m->v[h] != -1) h = (h + 1) & (uint64_t)(m->cap - 1);
    m->k[h].x = x; m->k[h].y = y; m->k[h].th = th; m->v[h] = v; m->n++;
}

typedef struct {
    const orc_ctx *c;
    node_t *pool; int64_t npool, cpool;
    int32_t *heap; int64_t nheap, cheap;        /* open_list.queue */
    int32_t *closed; int64_t nclosed, cclosed;  /* closed_list     */
    pmap poses;                                 /* pose -> node (open or closed) */
    imap by_index;                              /* node.index -> pool idx (closed nodes only) */
    dij_t *dij;
    double goal[3];
    int64_t global_index;
    int64_t n_checks, n_rs, n_dij_resume, n_closed_hit, n_open_hit, n_improved, n_collided, n_pushed;
    int status;
} astar_t;

static void aheap_siftdown(astar_t *a, int64_t startpos, int64_t pos)
{
    int32_t newitem = a->heap[pos];
    while (pos > startpos) {
        int64_t parentpos = (pos - 1) >> 1;
        int32_t parent = a->heap[parentpos];
        if (a->pool[newitem].f < a->pool[parent].f) { a->heap[pos] = parent; pos = parentpos; continue; }
        break;
    }
    a->heap[pos] = newitem;
}
static void aheap_siftup(astar_t *a, int64_t pos)
{
    int64_t endpos = a->nheap, startpos = pos;
    int32_t newitem = a->heap[pos];
    int64_t childpos = 2 * pos + 1;
    while (childpos < endpos) {
        int64_t rightpos = childpos + 1;
        if (rightpos < endpos && !(a->pool[a->heap[childpos]].f < a->pool[a->heap[rightpos]].f)) childpos = rightpos;
        a->heap[pos] = a->heap[childpos];
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    a->heap[pos] = newitem;
    aheap_siftdown(a, startpos, pos);
}
static void aheap_push(astar_t *a, int32_t item)
{
    if (a->nheap == a->cheap) { a->cheap *= 2; a->heap = realloc(a->heap, sizeof(int32_t) * a->cheap); }
    a->heap[a->nheap++] = item;
    aheap_siftdown(a, 0, a->nheap - 1);
}
static int32_t aheap_pop(astar_t *a)
{
    int32_t lastelt = a->heap[--a->nheap];
    if (a->nheap) { int32_t ret = a->heap[0]; a->heap[0] = lastelt; aheap_siftup(a, 0); return ret; }
    return lastelt;
}
static int32_t a_new_node(astar_t *a)
{
    if (a->npool == a->cpool) { a->cpool *= 2; a->pool = realloc(a->pool, sizeof(node_t) * a->cpool); }
    memset(&a->pool[a->npool], 0, sizeof(node_t));
    return (int32_t)a->npool++;
}
static void a_close(astar_t *a, int32_t ni)
{
    if (a->nclosed == a->cclosed) { a->cclosed *= 2; a->closed = realloc(a->closed, sizeof(int32_t) * a->cclosed); }
    a->closed[a->nclosed++] = ni;
    int32_t *s = imap_slot(&a->by_index, a->pool[ni].index, 1);
    if (*s == -2) *s = ni;
}

/* hybrid_a_star.py:243-259 */
static double calc_node_cost(const orc_ctx *c, int node_forward, double node_theta, double father_theta, int father_gear)
{
    double cost_gear = 0;
    if (node_forward != father_gear) cost_gear = c->cost_gear;
    double cost_heading = fabs(node_theta - father_theta);
    double cost = cost_gear + c->cost_heading * cost_heading;
    return c->cost_scale * cost;
}

/* hybrid_a_star.py:261-298; returns 0 ok, <0 error */
static int calc_node_heuristic(astar_t *a, double x, double y, double theta, double *h_out)
{
    const orc_ctx *c = a->c;
    int64_t id = orc_pos_to_index(c, x, y);
    int64_t h1 = orc_dij_lookup(a->dij, id);
    if (h1 < 0) {
        a->n_dij_resume++;
        h1 = orc_dij_compute_path(a->dij, x, y);
        if (h1 < 0) return -2;   /* H_UNREACHABLE */
    }
    double max_c = 1 / c->min_radius;
    double q0[3] = { x, y, theta };
    rs_path p; p.x = p.y = p.yaw = NULL; p.dir = NULL;
    a->n_rs++;
    int st = rs_optimal(q0, a->goal, max_c, &p, 0);
    if (st) return -3;
    double hv1 = (double)h1 / 100;
    double hv2 = p.L;
    *h_out = hv1 > hv2 ? hv1 : (hv2 > hv1 ? hv2 : hv1);  /* Python max(): first wins on ties (same value) */
    return 0;
}

/* pop trace record layout (doubles): index, parent, grid_id, x, y, theta, g, h, f, forward, steer */
#define TRACE_W 11

typedef struct {
    int32_t status;           /* 0 ok; 1 NO_PATH (open list empty); 2 H_UNREACHABLE; 3 RS error; 4 ITER_LIMIT; 5 capacity */
    int32_t in_radius_last;   /* info['in_radius'] of the last pop  */
    int32_t rs_valid;         /* an rs_path object exists (not None) */
    int32_t rs_collision;     /* collision flag of the last try_rs_curve */
    int64_t n_pops, n_closed, n_open, global_index;
    int64_t n_checks, n_rs, n_dij_calls, n_dij_closed;
    int64_t n_closed_hit, n_open_hit, n_improved, n_collided, n_pushed;
    int32_t n_astar, n_rs_pts, n_final;
    int32_t rs_n; int8_t rs_types[8]; double rs_lengths[5]; double rs_L;
} orc_plan_out;

/* path_planner.py:58-110 + hybrid_a_star.py. Buffers: trace[max_trace*11], astar_path[max_path*3],
 * rs_xyyaw[max_path*3], rs_dir[max_path], final_path[max_path*3]. */
ORC_API int32_t orc_plan(const orc_ctx *c, const double start[3], const double goal[3], orc_plan_out *out,
                         double *trace, int64_t max_trace, double *astar_path, double *rs_xyyaw, int8_t *rs_dir,
                         double *final_path, int32_t max_path,
                         int64_t *h_closed_id, int64_t *h_closed_dist, int64_t max_h)
{
    memset(out, 0, sizeof(*out));
    astar_t A; memset(&A, 0, sizeof(A));
    astar_t *a = &A;
    a->c = c;
    a->cpool = 1 << 12; a->pool = malloc(sizeof(node_t) * a->cpool);
    a->cheap = 1 << 10; a->heap = malloc(sizeof(int32_t) * a->cheap);
    a->cclosed = 1 << 12; a->closed = malloc(sizeof(int32_t) * a->cclosed);
    pmap_init(&a->poses, 1 << 12);
    imap_init(&a->by_index, 1 << 12);
    /* hybrid_a_star.__init__ :72-124 */
    a->dij = orc_dij_create(c, goal[0], goal[1]);
    int64_t d0 = orc_dij_compute_path(a->dij, start[0], start[1]);
    double *rsx = malloc(sizeof(double) * RS_MAXPTS * 3);
    int8_t *rsd = malloc(RS_MAXPTS);
    rs_path rs; rs.x = rsx; rs.y = rsx + RS_MAXPTS; rs.yaw = rsx + 2 * RS_MAXPTS; rs.dir = rsd; rs.npts = 0; rs.n = 0;
    int rs_valid = 0, collision = 0, in_radius = 0;
    int32_t cur = -1;
    int status = 0;
    if (d0 < 0) { status = 2; goto done; }
    a->goal[0] = goal[0]; a->goal[1] = goal[1]; a->goal[2] = orc_pi_2_pi(goal[2]);
    {
        int32_t n0 = a_new_node(a);
        node_t *n = &a->pool[n0];
        n->index = 0; n->parent_index = -1; n->x = start[0]; n->y = start[1]; n->theta = orc_pi_2_pi(start[2]);
        n->forward = 1; n->steer_i = -1; n->in_open = 1;
        aheap_push(a, n0);
        pmap_put(&a->poses, n->x, n->y, n->theta, n0);
    }
    const int nst = c->n_steer;
    const int next_index = 2 * nst;
    const int nsub = (int)ceil(c->dt / c->ddt);
    const double max_c = 1 / c->min_radius;
    int reach_goal = 0;
    /* path_planner.py:68 */
    while (a->nheap > 0 && !reach_goal) {
        if (c->max_pops > 0 && out->n_pops >= c->max_pops) { status = 4; break; }
        cur = aheap_pop(a);
        node_t cn = a->pool[cur];
        if (trace && out->n_pops < max_trace) {
            double *t = trace + out->n_pops * TRACE_W;
            t[0] = (double)cn.index; t[1] = (double)cn.parent_index; t[2] = (double)orc_pos_to_index(c, cn.x, cn.y);
            t[3] = cn.x; t[4] = cn.y; t[5] = cn.theta; t[6] = cn.g; t[7] = cn.h; t[8] = cn.f;
            t[9] = cn.forward; t[10] = cn.steer_i < 0 ? NAN : c->steer[cn.steer_i];
        }
        out->n_pops++;
        /* try_reach_goal :300-316 */
        collision = 0; in_radius = 0; rs_valid = 0;
        double ddx = cn.x - a->goal[0], ddy = cn.y - a->goal[1];
        double distance = sqrt(POW2(ddx) + POW2(ddy));
        if (distance < c->flag_radius) {
            in_radius = 1;
            /* try_rs_curve :318-349 */
            double q0[3] = { cn.x, cn.y, cn.theta };
            a->n_rs++;
            int st = rs_optimal(q0, a->goal, max_c, &rs, RS_MAXPTS);
            if (st) { status = 3; break; }
            rs_valid = 1;
            for (int i = 0; i < rs.npts; i++) {
                a->n_checks++;
                collision = orc_check(c, rs.x[i], rs.y[i], orc_pi_2_pi(rs.yaw[i]));
                if (collision) break;
            }
        }
        if (!collision && in_radius) { reach_goal = 1; break; }
        /* expand_node :126-241 */
        for (int i = 0; i < next_index; i++) {
            int si = i % nst;
            double speed; int is_forward;
            if (i < next_index / 2.0) { speed = c->max_v; is_forward = 1; } else { speed = -c->max_v; is_forward = 0; }
            double travel = speed * c->dt;
            double theta_ = cn.theta + (c->max_v * c->steer_tan[si]) / c->lw * c->dt;
            theta_ = orc_pi_2_pi(theta_);
            double x_ = cn.x + travel * cos(theta_);
            double y_ = cn.y + travel * sin(theta_);
            int32_t hit = pmap_get(&a->poses, x_, y_, theta_);
            int find_closed = 0;
            if (a->nclosed > 0) {
                /* :155-163: the first closed node either matches or triggers the bounds test */
                int oob = (x_ > c->b[1] || x_ < c->b[0] || y_ > c->b[3] || y_ < c->b[2]);
                if (hit >= 0 && a->pool[hit].in_closed) {
                    /* an equal closed node exists; if the node is also out of bounds the result is the same */
                    find_closed = 1; a->n_closed_hit++;
                } else if (oob) find_closed = 1;
            }
            if (find_closed) continue;
            int32_t child = -1;
            int find_open = 0;
            if (hit >= 0 && a->pool[hit].in_open && !a->pool[hit].in_closed) { child = hit; find_open = 1; }
            if (!find_open) {
                child = a_new_node(a);
                node_t *ch = &a->pool[child];
                ch->x = x_; ch->y = y_; ch->theta = theta_;
                ch->index = a->global_index + i + 1; ch->parent_index = cn.index;
                ch->forward = (int8_t)is_forward; ch->steer_i = (int8_t)si;
                int coll = 0;
                for (int j = 0; j < nsub; j++) {
                    double td = speed * c->ddt * (j + 1);
                    double th_i = cn.theta + (c->max_v * c->steer_tan[si]) / c->lw * c->ddt * (j + 1);
                    th_i = orc_pi_2_pi(th_i);
                    double x_i = cn.x + td * cos(th_i);
                    double y_i = cn.y + td * sin(th_i);
                    a->n_checks++;
                    coll = orc_check(c, x_i, y_i, th_i);
                    if (coll) {
                        ch = &a->pool[child];
                        ch->in_closed = 1;
                        a_close(a, child);
                        pmap_put(&a->poses, x_, y_, theta_, child);
                        a->n_collided++;
                        break;
                    }
                }
                if (!coll) {
                    double g = calc_node_cost(c, is_forward, theta_, cn.theta, cn.forward);
                    double h;
                    int hs = calc_node_heuristic(a, x_, y_, theta_, &h);
                    if (hs) { status = hs == -2 ? 2 : 3; goto done; }
                    ch = &a->pool[child];
                    ch->g = g; ch->h = h; ch->f = g + h;
                    ch->in_open = 1;
                    aheap_push(a, child);
                    pmap_put(&a->poses, x_, y_, theta_, child);
                    a->n_pushed++;
                }
            } else {
                a->n_open_hit++;
                node_t *ch = &a->pool[child];
                double new_h;
                int hs = calc_node_heuristic(a, ch->x, ch->y, ch->theta, &new_h);
                if (hs) { status = hs == -2 ? 2 : 3; goto done; }
                ch = &a->pool[child];
                double new_g = calc_node_cost(c, ch->forward, ch->theta, cn.theta, cn.forward);
                double new_f = new_h + new_g;
                if (new_f < ch->f) {
                    ch->f = new_f; ch->g = new_g; ch->h = new_h;
                    ch->parent_index = cn.index; ch->forward = (int8_t)is_forward; ch->steer_i = (int8_t)si;
                    a->n_improved++;
                }
            }
        }
        a->pool[cur].in_closed = 1; a->pool[cur].in_open = 0;
        a_close(a, cur);
        a->global_index += next_index;
    }
    if (status == 0 && !reach_goal) status = 1;
done:
    out->status = status;
    out->in_radius_last = in_radius; out->rs_valid = rs_valid; out->rs_collision = collision;
    out->n_closed = a->nclosed; out->n_open = a->nheap; out->global_index = a->global_index;
    out->n_checks = a->n_checks; out->n_rs = a->n_rs;
    out->n_dij_calls = a->dij->n_calls; out->n_dij_closed = a->dij->nclosed;
    out->n_closed_hit = a->n_closed_hit; out->n_open_hit = a->n_open_hit; out->n_improved = a->n_improved;
    out->n_collided = a->n_collided; out->n_pushed = a->n_pushed;
    /* finish_path :351-389 */
    if (cur >= 0 && (status == 0 || status == 1) && astar_path) {
        int64_t cap = 1024, np = 0;
        int32_t *chain = malloc(sizeof(int32_t) * cap);
        int32_t node = cur;
        while (a->pool[node].index != 0) {
            if (np == cap) { cap *= 2; chain = realloc(chain, sizeof(int32_t) * cap); }
            chain[np++] = node;
            int32_t *s = imap_slot(&a->by_index, a->pool[node].parent_index, 0);
            if (!s) break;
            node = *s;
        }
        if (np == cap) { cap *= 2; chain = realloc(chain, sizeof(int32_t) * cap); }
        chain[np++] = node;
        int32_t cnt = 0;
        int overflow = 0;
        #define PUSH_PT(buf, X, Y, T) do { if (cnt < max_path) { buf[3 * cnt] = X; buf[3 * cnt + 1] = Y; buf[3 * cnt + 2] = T; cnt++; } else overflow = 1; } while (0)
        PUSH_PT(astar_path, a->pool[node].x, a->pool[node].y, a->pool[node].theta);
        for (int64_t i = 0; i < np; i++) {
            int64_t k = np - 1 - i;
            if (k == 0) break;
            const node_t *par = &a->pool[chain[k]], *chd = &a->pool[chain[k - 1]];
            for (int j = 0; j < nsub; j++) {
                double speed = chd->forward ? c->max_v : -c->max_v;
                double td = speed * c->ddt * (j + 1);
                double th_j = par->theta + (c->max_v * c->steer_tan[chd->steer_i]) / c->lw * c->ddt * (j + 1);
                th_j = orc_pi_2_pi(th_j);
                double x_j = par->x + td * cos(th_j);
                double y_j = par->y + td * sin(th_j);
                PUSH_PT(astar_path, x_j, y_j, th_j);
            }
        }
        out->n_astar = cnt;
        free(chain);
        /* path_planner.py:100-108 */
        if (rs_valid) {
            out->rs_n = rs.n; out->rs_L = rs.L;
            for (int k = 0; k < rs.n; k++) { out->rs_types[k] = rs.t[k]; out->rs_lengths[k] = rs.l[k]; }
            out->n_rs_pts = rs.npts;
            for (int k = 0; k < rs.npts && k < max_path; k++) {
                rs_xyyaw[3 * k] = rs.x[k]; rs_xyyaw[3 * k + 1] = rs.y[k]; rs_xyyaw[3 * k + 2] = rs.yaw[k];
                rs_dir[k] = rs.dir[k];
            }
            if (rs.npts > max_path) overflow = 1;
            int32_t na = cnt;
            for (int k = 0; k < na && k < max_path; k++) { final_path[3 * k] = astar_path[3 * k]; final_path[3 * k + 1] = astar_path[3 * k + 1]; final_path[3 * k + 2] = astar_path[3 * k + 2]; }
            for (int k = 1; k < rs.npts; k++) PUSH_PT(final_path, rs.x[k], rs.y[k], rs.yaw[k]);
            out->n_final = cnt;
        }
        if (overflow) out->status = 5;
    }
    if (h_closed_id) {
        int64_t n = a->dij->nclosed < max_h ? a->dij->nclosed : max_h;
        for (int64_t i = 0; i < n; i++) { const grid_t *g = &a->dij->pool[a->dij->closed[i]]; h_closed_id[i] = g->id; h_closed_dist[i] = g->dist; }
    }
    orc_dij_destroy(a->dij);
    free(a->pool); free(a->heap); free(a->closed); pmap_free(&a->poses); imap_free(&a->by_index);
    free(rsx); free(rsd);
    return out->status;
}

/* ------------------------------------------------------------------------------------------ */
/* split_path: path_planner.py:112-192. Output: concatenated points + per-segment lengths.     */
static double cosine_dist(double u0, double u1, double v0, double v1)
{   /* scipy.spatial.distance.cosine with numpy/OpenBLAS ddot rounding */
    double uv = fma(u1, v1, u0 * v0), uu = fma(u1, u1, u0 * u0), vv = fma(v1, v1, v0 * v0);
    double dist = 1.0 - uv / sqrt(uu * vv);
    if (dist < 0.0) dist = 0.0; else if (dist > 2.0) dist = 2.0;   /* np.clip keeps NaN */
    return dist;
}

ORC_API int32_t orc_split_path(const orc_ctx *c, const double *fp, int32_t n, double *out_pts, int32_t max_pts,
                               int32_t *seg_len, int32_t max_seg, int32_t *n_seg_out, int32_t *change_gear_out)
{
    int32_t nseg = 0, npts = 0, change_gear = 0, start = 0, have_ext = 0;
    int32_t last_seg_start = 0;
    const int extend_num = c->extended_num;
    for (int32_t i = 0; i < n - 2; i++) {
        double v1x = fp[3 * (i + 1)] - fp[3 * i], v1y = fp[3 * (i + 1) + 1] - fp[3 * i + 1];
        double v2x = fp[3 * (i + 2)] - fp[3 * (i + 1)], v2y = fp[3 * (i + 2) + 1] - fp[3 * (i + 1) + 1];
        double cosin = 1 - cosine_dist(v1x, v1y, v2x, v2y);
        if (cosin < 0) {
            change_gear++;
            int32_t end = i + 2;
            if (nseg >= max_seg) return -1;
            int32_t seg_start = npts, cnt = 0;
            if (change_gear > 1 && have_ext > 0) {
                /* insert(0, ...) of pre_path[-(h-j)] for j=0..h-1 => reversed order of the last h points */
                int32_t prev_end = last_seg_start + seg_len[nseg - 1];
                for (int j = have_ext - 1; j >= 0; j--) {
                    const double *p = &out_pts[3 * (prev_end - (have_ext - j))];
                    if (npts >= max_pts) return -1;
                    double px = p[0], py = p[1], pt = p[2];
                    out_pts[3 * npts] = px; out_pts[3 * npts + 1] = py; out_pts[3 * npts + 2] = pt; npts++; cnt++;
                }
                have_ext = 0;
            }
            for (int32_t k = start; k < end; k++) {
                if (npts >= max_pts) return -1;
                out_pts[3 * npts] = fp[3 * k]; out_pts[3 * npts + 1] = fp[3 * k + 1]; out_pts[3 * npts + 2] = fp[3 * k + 2]; npts++; cnt++;
            }
            for (int j = 0; j < extend_num; j++) {
                double th_i = fp[3 * i + 2];
                int f1 = (fp[3 * (i + 1)] > fp[3 * i]) && (th_i > -PI / 2 && th_i < PI / 2);
                int f2 = (fp[3 * (i + 1)] < fp[3 * i]) && ((th_i > PI / 2 && th_i < PI) || (th_i > -PI && th_i < -PI / 2));
                double speed = (f1 || f2) ? c->max_v : -c->max_v;
                double td = speed * c->ddt * (j + 1);
                double th_j = fp[3 * (i + 1) + 2];
                double x_j = fp[3 * (i + 1)] + td * cos(th_j);
                double y_j = fp[3 * (i + 1) + 1] + td * sin(th_j);
                if (!orc_check(c, x_j, y_j, th_j)) {
                    if (npts >= max_pts) return -1;
                    out_pts[3 * npts] = x_j; out_pts[3 * npts + 1] = y_j; out_pts[3 * npts + 2] = th_j; npts++; cnt++;
                    have_ext++;
                }
            }
            seg_len[nseg++] = cnt;
            last_seg_start = seg_start;
            start = i + 1;
        }
    }
    if (nseg == 0) return -2;   /* IndexError path_planner.py:181 */
    if (nseg >= max_seg) return -1;
    {
        int32_t cnt = 0;
        if (have_ext > 0) {
            int32_t prev_end = last_seg_start + seg_len[nseg - 1];
            for (int j = have_ext - 1; j >= 0; j--) {
                const double *p = &out_pts[3 * (prev_end - (have_ext - j))];
                if (npts >= max_pts) return -1;
                double px = p[0], py = p[1], pt = p[2];
                out_pts[3 * npts] = px; out_pts[3 * npts + 1] = py; out_pts[3 * npts + 2] = pt; npts++; cnt++;
            }
        }
        for (int32_t k = start; k < n; k++) {
            if (npts >= max_pts) return -1;
            out_pts[3 * npts] = fp[3 * k]; out_pts[3 * npts + 1] = fp[3 * k + 1]; out_pts[3 * npts + 2] = fp[3 * k + 2]; npts++; cnt++;
        }
        seg_len[nseg++] = cnt;
    }
    *n_seg_out = nseg;
    *change_gear_out = change_gear;
    return npts;
}

/* ------------------------------------------------------------------------------------------ */
/* Corridor bounds: path_opti.compute_collision_H, optimization/path_optimazition.py:221-409.   */
/* Per way-point: free distance (capped at expand_dis) to the nearest obstacle point in +x, +y,  */
/* -x, -y, measured along the axes from the vehicle edge the point faces. The 4 heading cases x */
/* 4 areas of the reference are one rotation pattern: area k (0 right, 1 front, 2 left, 3 rear)  */
/* in heading case c looks towards quadrant (k + c - 1) mod 4 of [(x+,y-), (x+,y+), (x-,y+),     */
/* (x-,y-)]. out[i] = {x_max + x, y_max + y, x - x_min, y - y_min} (H_max / H_min rows).          */
ORC_API void orc_corridor_batch(const orc_ctx *c, double expand, const double *px_, const double *py_, const double *pt_,
                                int64_t n, double *out)
{
    for (int64_t q = 0; q < n; q++) {
        const double x = px_[q], y = py_[q], theta = pt_[q];
        double vb[8];
        orc_corners(c, x, y, theta, vb);
        double xhi = vb[0], xlo = vb[0], yhi = vb[1], ylo = vb[1];
        for (int i = 1; i < 4; i++) {
            if (vb[2 * i] > xhi) xhi = vb[2 * i];
            if (vb[2 * i] < xlo) xlo = vb[2 * i];
            if (vb[2 * i + 1] > yhi) yhi = vb[2 * i + 1];
            if (vb[2 * i + 1] < ylo) ylo = vb[2 * i + 1];
        }
        xhi = xhi + expand; xlo = xlo - expand; yhi = yhi + expand; ylo = ylo - expand;      /* :254-257 */
        double k[4], b[4], area[4][4];
        for (int i = 0; i < 4; i++) {
            const int j = (i + 1) & 3;
            k[i] = (vb[2 * j + 1] - vb[2 * i + 1]) / (vb[2 * j] - vb[2 * i]);
            b[i] = vb[2 * i + 1] - k[i] * vb[2 * i];
            area[i][0] = vb[2 * i] < vb[2 * j] ? vb[2 * i] : vb[2 * j];                      /* get_area_boundary :287-292 */
            area[i][1] = vb[2 * i] > vb[2 * j] ? vb[2 * i] : vb[2 * j];
            area[i][2] = vb[2 * i + 1] < vb[2 * j + 1] ? vb[2 * i + 1] : vb[2 * j + 1];
            area[i][3] = vb[2 * i + 1] > vb[2 * j + 1] ? vb[2 * i + 1] : vb[2 * j + 1];
        }
        int cs = 0;                                                                          /* :341-348 */
        if (theta >= -PI && theta < -PI / 2) cs = 3;
        else if (theta >= -PI / 2 && theta < 0) cs = 4;
        else if (theta >= 0 && theta < PI / 2) cs = 1;
        else if (theta >= PI / 2 && theta <= PI) cs = 2;
        double x_min = expand, x_max = expand, y_min = expand, y_max = expand;
        const double ac = fabs(cos(theta)), as = fabs(sin(theta));
        if (cs) {
            for (int32_t o = 0; o < c->P; o++) {
                const double ox = c->ox[o], oy = c->oy[o];
                if (!(ox >= xlo && ox <= xhi)) continue;
                if (!(oy >= ylo && oy <= yhi)) continue;
                for (int kk = 0; kk < 4; kk++) {
                    const int quad = (kk + cs - 1) & 3;
                    const int xpos = quad == 0 || quad == 1, ypos = quad == 1 || quad == 2;
                    const double ax0 = xpos ? area[kk][0] : area[kk][0] - expand, ax1 = xpos ? area[kk][1] + expand : area[kk][1];
                    const double ay0 = ypos ? area[kk][2] : area[kk][2] - expand, ay1 = ypos ? area[kk][3] + expand : area[kk][3];
                    if (ox > ax0 && ox < ax1 && oy > ay0 && oy < ay1) {
                        const double sd = fabs(k[kk] * ox + b[kk] - oy) / sqrt(1 + k[kk] * k[kk]);   /* :294-296 */
                        const double ver = sd / ac, hor = sd / as;                                    /* :299-303 */
                        if (xpos) { if (hor < x_max) x_max = hor; } else { if (hor < x_min) x_min = hor; }
                        if (ypos) { if (ver < y_max) y_max = ver; } else { if (ver < y_min) y_min = ver; }
                        break;
                    }
                }
            }
        }
        out[4 * q] = x_max + x; out[4 * q + 1] = y_max + y; out[4 * q + 2] = x - x_min; out[4 * q + 3] = y - y_min;
    }
}

ORC_API int32_t orc_sizeof_ctx(void) { return (int32_t)sizeof(orc_ctx); }
ORC_API int32_t orc_sizeof_plan_out(void) { return (int32_t)sizeof(orc_plan_out); }
ORC_API double orc_py_hypot(double a, double b) { return py_hypot(a, b); }

/* ------------------------------------------------------------------------------------------ */
/* Per-sample stage of Map.detect_obstacle_edge, map/costmap.py:236-261. */
/* edges[e] = {p1x, p1y, cos, sin, length, count}: the host (numpy) half of the function (np.unique, */
/* argsort of centroid angles, arctan2/cos/sin of the edge, np.dot for the length) stays in numpy. */
/* For q < count: t = np.linspace(0, length, count)[q]; np.dot(rot.T, [[t],[0]]) = (cos*t, sin*t)   */
/* (the second product of each 2-term dot is an exact zero); + p1; np.where((X < p) & (X > p - dx)) */
/* scanned over the whole axis as the reference does. Returns the number of samples that matched   */
/* more than one node on an axis (the reference raises TypeError there, :260).                      */
ORC_API int orc_rasterize_edges(const double *xs, const double *ys, int nx, int ny, double dx, double dy,
                                const double *edges, long n_edges, unsigned char *occ /* nx*ny */)
{
    int multi = 0;
    for (long e = 0; e < n_edges; e++) {
        const double *r = edges + 6 * e;
        const int count = (int)r[5];
        for (int q = 0; q < count; q++) {
            /* numpy.linspace (numpy/_core/function_base.py:linspace) with start = 0 */
            double t = (double)q;
            const int div = count - 1;
            if (div > 0) {
                const double step = r[4] / (double)div;
                if (step == 0.0) { t = t / (double)div; t = t * r[4]; }
                else t = t * step;
            } else t = t * r[4];
            t = t + 0.0;
            if (count > 1 && q == count - 1) t = r[4];
            const double lx = r[2] * t, ly = r[3] * t;
            const double px = lx + r[0], py = ly + r[1];
            int ni = 0, nj = 0, i0 = -1, j0 = -1;
            for (int i = 0; i < nx; i++) if (xs[i] < px && xs[i] > px - dx) { if (!ni) i0 = i; ni++; }
            for (int j = 0; j < ny; j++) if (ys[j] < py && ys[j] > py - dy) { if (!nj) j0 = j; nj++; }
            if (ni == 0 || nj == 0) continue;
            if (ni > 1 || nj > 1) { multi++; continue; }
            occ[(long)i0 * ny + j0] = 255;
        }
    }
    return multi;
}

/* ------------------------------------------------------------------------------------------ */
/* All-core CPU baseline (bench.py's cpu_baseline_all_cores leg; tests/test_oracle_batch.py): the same orc_plan, one     */
/* problem per thread drawn from an atomic ticket counter, no Python in the loop. Two modes:                            */
/*   min_seconds <= 0 : ONE pass over the n problems; status[i] / pops[i] of every problem are written.                 */
/*   min_seconds  > 0 : steady state -- tickets cycle over the problems (ticket % n, in the caller's `order` when given) */
/*                      until the deadline; plans in flight are finished and counted, the clock stops when the last one  */
/*                      ends; status / pops hold the LAST result of each problem that was planned.                       */
/* totals = { plans, completed (status 0 or 1), pops }; *elapsed = wall seconds from the first ticket to the last plan.  */
/* orc_plan keeps no mutable global state (g_restated / g_dij_reheap are read only while it runs), so the threads share  */
/* nothing but the read-only context, the counter and glibc's per-thread malloc arenas.                                  */
#include <pthread.h>
#include <stdatomic.h>
#include <time.h>
#include <malloc.h>

typedef struct {
    const orc_ctx *c; const double *starts, *goals; const int32_t *order;
    int64_t n; double deadline; int steady;
    atomic_llong ticket; atomic_llong plans, completed, pops;
    int32_t *status; int64_t *pops_out;
} orc_batch_t;

static double orc_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static void *orc_batch_worker(void *arg)
{
    orc_batch_t *b = (orc_batch_t *)arg;
    enum { MAXP = 1024 };
    double *ap = malloc(sizeof(double) * MAXP * 3), *rp = malloc(sizeof(double) * MAXP * 3), *fp = malloc(sizeof(double) * MAXP * 3);
    int8_t *rd = malloc(MAXP);
    for (;;) {
        if (b->steady && orc_now() >= b->deadline) break;
        const long long t = atomic_fetch_add(&b->ticket, 1);
        if (!b->steady && t >= b->n) break;
        int64_t i = t % b->n;
        if (b->order) i = b->order[i];
        orc_plan_out out;
        orc_plan(b->c, b->starts + 3 * i, b->goals + 3 * i, &out, NULL, 0, ap, rp, rd, fp, MAXP, NULL, NULL, 0);
        if (b->status) b->status[i] = out.status;
        if (b->pops_out) b->pops_out[i] = out.n_pops;
        atomic_fetch_add(&b->plans, 1);
        atomic_fetch_add(&b->completed, (out.status == 0 || out.status == 1) ? 1 : 0);
        atomic_fetch_add(&b->pops, out.n_pops);
    }
    free(ap); free(rp); free(fp); free(rd);
    return NULL;
}

ORC_API int32_t orc_plan_batch(const orc_ctx *c, const double *starts, const double *goals, int64_t n, int32_t threads,
                               double min_seconds, const int32_t *order, int32_t *status, int64_t *pops,
                               int64_t totals[3], double *elapsed)
{
    if (!c || n <= 0 || threads <= 0 || !starts || !goals) return -1;
    /* orc_plan allocates its Dijkstra pool, heaps and hash maps per call (3 - 20 MB, grown by realloc): with glibc's defaults every
     * one of them is an mmap / munmap pair, and a hundred threads then queue on the process's address-space lock (measured on the
     * GPU box's 128 cores before this: 11 x one core). Keep freed memory in the per-thread arenas instead: no mmap for blocks
     * below 32 MB, no trimming. Allocation strategy only -- nothing orc_plan computes depends on it. */
    mallopt(M_MMAP_THRESHOLD, 32 << 20);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 16 << 20);
    orc_batch_t b;
    b.c = c; b.starts = starts; b.goals = goals; b.order = order; b.n = n;
    b.steady = min_seconds > 0.0; b.status = status; b.pops_out = pops;
    atomic_init(&b.ticket, 0); atomic_init(&b.plans, 0); atomic_init(&b.completed, 0); atomic_init(&b.pops, 0);
    pthread_t *th = malloc(sizeof(pthread_t) * (size_t)threads);
    const double t0 = orc_now();
    b.deadline = t0 + min_seconds;
    int started = 0;
    for (int k = 0; k < threads; k++) { if (pthread_create(&th[k], NULL, orc_batch_worker, &b) != 0) break; started++; }
    for (int k = 0; k < started; k++) pthread_join(th[k], NULL);
    const double t1 = orc_now();
    free(th);
    if (totals) { totals[0] = atomic_load(&b.plans); totals[1] = atomic_load(&b.completed); totals[2] = atomic_load(&b.pops); }
    if (elapsed) *elapsed = t1 - t0;
    return started;
}

/* ------------------------------------------------------------------------------------------ */
/* Is this host's libm the one the device restates (glibc 2.35, x86-64 FMA variants)? Compares the platform's atan2 /     */
/* asin / acos / tan / pow(v, 2) with include/avp_libm.h on n pseudo-random arguments each (the Reeds-Shepp argument      */
/* shapes) and returns the number of results that differ in any bit. oracle.device_arithmetic() runs it once per process: */
/* on a host with another glibc the GPU parity tests must compare against the RESTATED mode, which is what the device     */
/* implements by specification, instead of failing although the device follows its spec.                                  */
ORC_API int64_t orc_libm_selfcheck(int64_t n, uint64_t seed)
{
    uint64_t s = seed ? seed : 0x9E3779B97F4A7C15ULL;
    int64_t bad = 0;
    #define ORC_RND() (s ^= s << 13, s ^= s >> 7, s ^= s << 17, (double)(s >> 11) * (1.0 / 9007199254740992.0))
    #define ORC_NE(a, b) (memcmp(&(double){ a }, &(double){ b }, 8) != 0 && !((a) != (a) && (b) != (b)))
    for (int64_t i = 0; i < n; i++) {
        const double u = 20.0 * ORC_RND() - 10.0, v = 20.0 * ORC_RND() - 10.0, w = 2.0 * ORC_RND() - 1.0;
        const double t = 4.0 * 3.14159265358979323846 * ORC_RND() - 2.0 * 3.14159265358979323846;
        double a, b;
        a = atan2(u, v); b = avp_atan2(u, v); bad += ORC_NE(a, b);
        a = atan2(2.0, u); b = avp_atan2(2.0, u); bad += ORC_NE(a, b);
        a = asin(w); b = avp_asin(w); bad += ORC_NE(a, b);
        a = acos(w); b = avp_acos(w); bad += ORC_NE(a, b);
        a = tan(t); b = avp_tan(t); bad += ORC_NE(a, b);
        a = pow(u, 2.0); b = avp_pow2(u); bad += ORC_NE(a, b);
    }
    #undef ORC_RND
    #undef ORC_NE
    return bad;
}
