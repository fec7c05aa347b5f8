// Host build of avp_math.h for CPU-side unit tests (bit-compare against this host's libm).
#include "avp_math.h"
extern "C" {
__attribute__((visibility("default"))) void avp_host_sincos(const double* x, long n, double* s, double* c)
{ for (long i = 0; i < n; i++) { s[i] = avp_sin(x[i]); c[i] = avp_cos(x[i]); } }
// the fused evaluation used by the kernels: must give the two values above bit for bit
__attribute__((visibility("default"))) void avp_host_sincos_fused(const double* x, long n, double* s, double* c)
{ for (long i = 0; i < n; i++) avp_sincos(x[i], s[i], c[i]); }
__attribute__((visibility("default"))) void avp_host_misc(const double* a, const double* b, long n, double* hyp, double* mod, double* p2p, double* M)
{ for (long i = 0; i < n; i++) { hyp[i] = avp_hypot(a[i], b[i]); mod[i] = avp_pymod(a[i], b[i]); p2p[i] = avp_pi_2_pi(a[i]); M[i] = avp_M(a[i]); } }
}
