// Host build of avp_math.h for CPU-side unit tests (bit-compare against this host's libm).
#include "avp_math.h"
#include "../../include/avp_libm.h"
extern "C" {
__attribute__((visibility("default"))) void avp_host_linspace0(double stop, int num, double* out)
{ for (int q = 0; q < num; q++) out[q] = avp_linspace0(stop, num, q); }
// index searches of the collision set-up: the four reference routines and the unified one the kernels use
__attribute__((visibility("default"))) void avp_host_node_search(const double* A, int n, double a0, double pitch, const double* v, long m,
                                                                 int* first_ge, int* last_le, int* first_gt, int* last_lt, int* uni_lo, int* uni_hi)
{
    for (long i = 0; i < m; i++) {
        first_ge[i] = avp_first_ge(A, n, a0, pitch, v[i]); last_le[i] = avp_last_le(A, n, a0, pitch, v[i]);
        first_gt[i] = avp_first_gt(A, n, a0, pitch, v[i]); last_lt[i] = avp_last_lt(A, n, a0, pitch, v[i]);
        uni_lo[i] = avp_node_search(A, n, a0, pitch, v[i], false); uni_hi[i] = avp_node_search(A, n, a0, pitch, v[i], true);
    }
}
__attribute__((visibility("default"))) void avp_host_sincos(const double* x, long n, double* s, double* c)
{ for (long i = 0; i < n; i++) { s[i] = avp_sin(x[i]); c[i] = avp_cos(x[i]); } }
// the fused evaluation used by the kernels: must give the two values above bit for bit
__attribute__((visibility("default"))) void avp_host_sincos_fused(const double* x, long n, double* s, double* c)
{ for (long i = 0; i < n; i++) avp_sincos(x[i], s[i], c[i]); }
__attribute__((visibility("default"))) void avp_host_misc(const double* a, const double* b, long n, double* hyp, double* mod, double* p2p, double* M)
{ for (long i = 0; i < n; i++) { hyp[i] = avp_hypot(a[i], b[i]); mod[i] = avp_pymod(a[i], b[i]); p2p[i] = avp_pi_2_pi(a[i]); M[i] = avp_M(a[i]); } }
// the restated glibc libm as the host compiles it (kind as in avp_libm_batch, include/avp.h)
__attribute__((visibility("default"))) void avp_host_libm(int kind, const double* a, const double* b, long n, double* out)
{
    for (long i = 0; i < n; i++) {
        switch (kind) {
        case 0: out[i] = avp_atan2(a[i], b[i]); break;
        case 1: out[i] = avp_asin(a[i]); break;
        case 2: out[i] = avp_acos(a[i]); break;
        case 3: out[i] = avp_tan(a[i]); break;
        default: out[i] = avp_pow2(a[i]); break;
        }
    }
}
}
