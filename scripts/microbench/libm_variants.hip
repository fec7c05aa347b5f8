// Cost of the scalar libm calls of the Reeds-Shepp words under load: every lane runs a chain of dependent calls on
// lane-varying arguments, WAVES waves per CU on every CU; reports wave-nanoseconds per call (= time x waves / calls).
// Variants of where glibc's uatan.tbl rows live (compile-time, -DVAR=k):
//   (0: the rounds-1-3 double-double atan2 -- removed with its header in round 5; its figures are in profiles/r04_libm_microbench.txt)
//   1 glibc, table in global memory, 7-word rows
//   2 glibc, table in LDS, 7-word rows         3 glibc, global, rows padded to 8 words (64 B, dwordx4 loads)
//   4 glibc, LDS, rows padded to 8 words
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DVAR=1 -I include scripts/microbench/libm_variants.hip -o /tmp/lv1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#ifndef VAR
#define VAR 1
#endif
#if VAR == 0
#error "VAR 0 (the rounds-1-3 libm) was removed in round 5"
#else
#if VAR == 2
__shared__ uint64_t CIJ_LDS[241 * 7];
#define AVPG_CIJ_ROW(i) (CIJ_LDS + 7 * (i))
#elif VAR == 3
struct alignas(64) Row8 { uint64_t w[8]; };
__device__ Row8 CIJ_PAD[241];
#define AVPG_CIJ_ROW(i) (CIJ_PAD[(i)].w)
#elif VAR == 4
struct alignas(64) Row8 { uint64_t w[8]; };
__shared__ Row8 CIJ_LDS8[241];
#define AVPG_CIJ_ROW(i) (CIJ_LDS8[(i)].w)
#endif
#include "avp_glibc_libm.h"
#define F_ATAN2 avpg_atan2
#define F_ASIN avpg_asin
#define F_ACOS avpg_acos
__device__ static inline double tan_g(double x) { double r; avpg_tan_try(x, &r); return r; }
#define F_TAN tan_g
#define F_POW2 avpg_pow2
#endif

#define N 64
__global__ void fill_pad()
{
#if VAR == 3
    for (int i = threadIdx.x; i < 241 * 7; i += blockDim.x) CIJ_PAD[i / 7].w[i % 7] = AVP_G_CIJ[i];
#endif
}
template <int WHICH>
__global__ void probe(double seed, double* sink)
{
#if VAR == 2
    for (int i = threadIdx.x; i < 241 * 7; i += blockDim.x) CIJ_LDS[i] = AVP_G_CIJ[i];
#elif VAR == 4
    for (int i = threadIdx.x; i < 241 * 7; i += blockDim.x) CIJ_LDS8[i / 7].w[i % 7] = AVP_G_CIJ[i];
#endif
    __syncthreads();
    const int lane = threadIdx.x + blockIdx.x * 7;
    const double a = -3.1 + 0.0977 * (lane & 63) + seed;            // lane-varying angle in (-pi, pi)
    const double sa = sin(a), ca = cos(a);
    double x = 1e-3 * (threadIdx.x >> 6);
    for (int i = 0; i < N; i++) {
        if (WHICH == 0) x = F_ATAN2(sa * (1.0 + 1e-9 * x), ca + 1e-7 * i);
        if (WHICH == 1) x = F_ASIN(0.97 * sa + 1e-12 * x) + F_ACOS(0.9 * ca - 1e-12 * x);
        if (WHICH == 2) x = F_TAN(a * 0.5 + 1e-9 * x);
        if (WHICH == 3) x = F_POW2(a + 1e-9 * x);
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <int WHICH>
static void run(const char* name, int waves, double* sink)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256, threads = 64 * waves;
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<WHICH>, dim3(blocks), dim3(threads), 0, 0, 1e-4 * rep, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    // each CU runs `waves` waves of N calls: wave-ns per call = time / N  (all CUs in parallel, waves share the CU)
    printf("VAR %d %-10s %2d waves/CU: %8.1f ns per chain step, %7.1f wave-ns per call (CU throughput: %6.1f ns per wave-call)\n",
           VAR, name, waves, best * 1e6 / N, best * 1e6 / N, best * 1e6 / N / waves);
}

int main()
{
    double* sink; hipMalloc(&sink, 256 * 1024 * sizeof(double));
    hipLaunchKernelGGL(fill_pad, dim3(1), dim3(256), 0, 0);
    for (int waves : { 1, 4, 8, 16 }) {
        run<0>("atan2", waves, sink);
        run<1>("asin+acos", waves, sink);
        run<2>("tan", waves, sink);
        run<3>("pow2", waves, sink);
    }
    return 0;
}
