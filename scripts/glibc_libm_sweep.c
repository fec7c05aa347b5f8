/* Compares include/avp_glibc_libm.h (host build, the same source the device compiles) with the platform libm,
 * bit for bit, on N arguments per function and distribution. Build + run (8 threads, ~1 min per 1e9):
 *   gcc -O2 -mfma -ffp-contract=off -fno-builtin -fopenmp -o /tmp/glibc_libm_sweep scripts/glibc_libm_sweep.c -lm
 *   /tmp/glibc_libm_sweep 1000000000
 * Prints one line per (function, distribution): tested, mismatches, first mismatching arguments. Exit code = any mismatch. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>
#include "../include/avp_glibc_libm.h"

static inline uint64_t splitmix(uint64_t* s) { uint64_t z = (*s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static inline double u01(uint64_t* s) { return (double)(splitmix(s) >> 11) * 0x1p-53; }
static inline double uni(uint64_t* s, double a, double b) { return a + (b - a) * u01(s); }
/* sign * 2^e * [1,2): e uniform in [elo, ehi] */
static inline double logu(uint64_t* s, int elo, int ehi) { const double m = 1.0 + u01(s); const int e = elo + (int)(splitmix(s) % (uint64_t)(ehi - elo + 1)); const double v = ldexp(m, e); return (splitmix(s) & 1) ? -v : v; }
static inline double anybits(uint64_t* s) { return avpg_from_bits(splitmix(s)); }
static inline int same(double a, double b) { return avpg_bits(a) == avpg_bits(b) || (a != a && b != b); }
static double tan_both(double x) { double r; avpg_tan_try(x, &r); return r; }

typedef struct { const char* name; int nargs; int dist; } job_t;
static long run(const char* fn, const char* dname, int dist, long n)
{
    long bad = 0; double fx = 0, fy = 0, fa = 0, fb = 0; int have = 0;
#pragma omp parallel reduction(+ : bad)
    {
        uint64_t s = 0x1234567ull * (uint64_t)(omp_get_thread_num() + 1) + (uint64_t)dist * 977 + (uint64_t)fn[1] * 131 + (uint64_t)fn[2];
#pragma omp for schedule(static)
        for (long i = 0; i < n; ++i) {
            double x = 0, y = 0, a, b;
            if (fn[0] == 'a' && fn[1] == 't') {          /* atan2(y, x) */
                switch (dist) {
                case 0: y = uni(&s, -10, 10); x = uni(&s, -10, 10); break;
                case 1: y = logu(&s, -40, 40); x = logu(&s, -40, 40); break;
                case 2: y = logu(&s, -1074, 1023); x = logu(&s, -1074, 1023); break;
                case 3: y = 2.0; x = uni(&s, -8, 8); break;                                   /* rs_curve.py:176 */
                case 4: y = uni(&s, 0, 8); x = -2.0; break;                                   /* rs_curve.py:414 */
                case 5: { const double t = uni(&s, -3.2, 3.2), r = uni(&s, 0.01, 30); y = r * sin(t); x = r * cos(t); break; }
                case 6: y = anybits(&s); x = anybits(&s); break;
                default: { const double u = uni(&s, 0.0, 1.0); x = logu(&s, -3, 3); y = x * (u < 0.5 ? 1.0 + 1e-13 * uni(&s, -1, 1) : 0.0625 * (1 + 1e-12 * uni(&s, -1, 1))); break; }
                }
                a = avpg_atan2(y, x); b = atan2(y, x); if (!same(avpg_atan2_ref(y, x), b)) a = NAN;
            } else if (fn[0] == 'a') {                    /* asin / acos */
                switch (dist) {
                case 0: x = uni(&s, -1, 1); break;
                case 1: x = logu(&s, -60, -1); break;
                case 2: { const double d = ldexp(u01(&s), -(int)(splitmix(&s) % 50)); x = (splitmix(&s) & 1) ? 1.0 - d : -1.0 + d; break; }
                case 3: x = uni(&s, 0.96, 1.0) * ((splitmix(&s) & 1) ? 1 : -1); break;
                default: x = anybits(&s); break;
                }
                if (fn[1] == 's') { a = avpg_asin(x); b = asin(x); } else { a = avpg_acos(x); b = acos(x); }
            } else if (fn[0] == 't') {
                switch (dist) {
                case 0: x = uni(&s, -6.3, 6.3); break;
                case 1: x = uni(&s, -25.5, 25.5); break;
                case 2: x = logu(&s, -40, 26); break;
                case 3: { const double k = (double)(int)uni(&s, -16, 16); x = k * 0x1.921fb54442d18p+0 + ldexp(uni(&s, -1, 1), -(int)(splitmix(&s) % 40)); break; }
                case 4: x = uni(&s, -1.05e8, 1.05e8); break;
                case 5: x = logu(&s, 26, 1023); break;                                      /* Payne-Hanek range (__branred) */
                default: x = anybits(&s); break;
                }
                a = tan_both(x); b = tan(x);
            } else {                                      /* pow(x, 2) */
                switch (dist) {
                case 0: x = uni(&s, -100, 100); break;
                case 1: x = logu(&s, -30, 30); break;
                case 2: x = logu(&s, -1074, 1023); break;
                case 3: x = 1.0 + ldexp(uni(&s, -1, 1), -(int)(splitmix(&s) % 60)); break;
                default: x = anybits(&s); break;
                }
                volatile double two = 2.0;
                a = avpg_pow2(x); b = pow(x, two);
            }
            if (!same(a, b)) {
                ++bad;
#pragma omp critical
                if (!have) { have = 1; fx = x; fy = y; fa = a; fb = b; }
            }
        }
    }
    printf("%-6s %-12s tested %ld mismatches %ld", fn, dname, n, bad);
    if (bad) printf("   first: x=%a y=%a ours=%a libm=%a", fx, fy, fa, fb);
    printf("\n");
    fflush(stdout);
    return bad;
}

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 10000000;
    const char* only = argc > 2 ? argv[2] : "";
    long bad = 0;
    static const char* an[] = { "box10", "log40", "logfull", "2_over_u", "r_over_m2", "polar", "anybits", "edges" };
    static const char* sn[] = { "uniform", "small", "near1", "sqrt_range", "anybits" };
    static const char* tn[] = { "pm2pi", "pm25", "log", "near_kpi2", "pm1e8", "huge", "anybits" };
    static const char* pn[] = { "pm100", "log30", "logfull", "near1", "anybits" };
    if (!*only || !strcmp(only, "atan2")) for (int d = 0; d < 8; ++d) bad += run("atan2", an[d], d, n);
    if (!*only || !strcmp(only, "asin")) for (int d = 0; d < 5; ++d) bad += run("asin", sn[d], d, n);
    if (!*only || !strcmp(only, "acos")) for (int d = 0; d < 5; ++d) bad += run("acos", sn[d], d, n);
    if (!*only || !strcmp(only, "tan")) for (int d = 0; d < 7; ++d) bad += run("tan", tn[d], d, n);
    if (!*only || !strcmp(only, "pow2")) for (int d = 0; d < 5; ++d) bad += run("pow2", pn[d], d, n);
    printf("total mismatches %ld\n", bad);
    return bad != 0;
}
