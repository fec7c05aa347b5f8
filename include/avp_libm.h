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
 * Every finite and non-finite argument is covered (tan's Payne-Hanek range included): no other libm is called.
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

AVP_LIBM_FN double avp_atan2(double y, double x) { return avpg_atan2(y, x); }
AVP_LIBM_FN double avp_asin(double x) { return avpg_asin(x); }
AVP_LIBM_FN double avp_acos(double x) { return avpg_acos(x); }
/* libm pow(v, 2.0) -- CPython's v ** 2 -- which is not v*v */
AVP_LIBM_FN double avp_pow2(double v) { return avpg_pow2(v); }

AVP_LIBM_FN double avp_tan(double x)
{
    double r;
    avpg_tan_try(x, &r);
    return r;
}

#endif /* AVP_LIBM_H */
