// Single-wave latency probes for the scalar maths of the hot path (not part of libavp_hip.so):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I include -I automatedvaletparking_amd/csrc \
//         scripts/microbench/latency.hip -o /tmp/latency && /tmp/latency
// Each probe runs a dependent chain of N calls in one wave and reports cycles per call (s_memtime ticks).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "avp_device.h"
#include "avp_rs_kernels.h"

#define N 256
template <typename F>
__device__ long long chain(double& x, F f)
{
    const long long t0 = clock64();
    for (int i = 0; i < N; i++) x = f(x);
    return clock64() - t0;
}

__global__ void probe(double seed, long long* out, double* sink)
{
    avp_lds_tables_fill<true>();
    rs_lds_tables_fill();
    double x = seed + 1e-3 * threadIdx.x;
    int k = 0;
    out[k++] = chain(x, [](double v) { return v * 1.0000001 + 1e-9; });                       // mul+add
    out[k++] = chain(x, [](double v) { return __builtin_fma(v, 1.0000001, 1e-9); });           // fma
    out[k++] = chain(x, [](double v) { return 1.0 / (v + 1.5); });                             // div
    out[k++] = chain(x, [](double v) { return sqrt(v + 1.5); });                               // sqrt
    out[k++] = chain(x, [](double v) { return avp_sin(v + 0.7); });                            // sin
    out[k++] = chain(x, [](double v) { return avp_sin(v + 0.7) + avp_cos(v + 0.7); });         // sin+cos same arg
    out[k++] = chain(x, [](double v) { return avp_atan2(v + 0.3, 1.1); });                     // atan2 (dd core)
    out[k++] = chain(x, [](double v) { return avp_asin(0.5 * avp_sin(v)); });                  // asin(sin)
    out[k++] = chain(x, [](double v) { return avp_tan(0.3 + 0.1 * avp_sin(v)); });             // tan(sin)
    out[k++] = chain(x, [](double v) { return avp_pymod_2pi(v + 7.0); });                      // mod 2pi
    out[k++] = chain(x, [](double v) { return avp_hypot(v, 1.3); });                           // CPython hypot
    out[k++] = chain(x, [](double v) { return floor((v + 3.0) / 0.1003); });                   // index maths
    sink[threadIdx.x] = x;
}

int main()
{
    long long* d; double* s; long long h[16];
    hipMalloc(&d, sizeof(h)); hipMalloc(&s, 64 * sizeof(double));
    const char* names[] = { "mul+add", "fma", "div", "sqrt", "sin", "sin+cos", "atan2", "asin(sin)", "tan(sin)", "pymod_2pi", "hypot", "floor(div)" };
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, 0.25, d, s);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    }
    for (int i = 0; i < 12; i++) printf("%-12s %8.1f cycles/call\n", names[i], (double)h[i] / N);
    return 0;
}
