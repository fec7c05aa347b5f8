// Wave-call latency of the scalar maths with LANE-VARYING arguments (angles spread over [-pi, pi], as in the planner:
// a wave pays every branch any of its lanes takes). Not part of libavp_hip.so.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I include -I automatedvaletparking_amd/csrc \
//         scripts/microbench/latency2.hip -o scripts/microbench/latency2 && scripts/microbench/latency2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "avp_device.h"
#include "avp_rs_kernels.h"

#define N 128
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
    const double a = -3.1 + 0.097 * threadIdx.x + seed;      // lane-varying angle in (-pi, pi)
    double x = a;
    int k = 0;
    out[k++] = chain(x, [&](double v) { double s, c; avp_sincos(a + 1e-9 * v, s, c); return s + c; });                    // sincos, spread args
    out[k++] = chain(x, [&](double v) { double s, c; avp_sincos(0.3 + 1e-9 * v, s, c); return s + c; });                 // sincos, uniform small arg
    out[k++] = chain(x, [&](double v) { return avp_atan2(sin(a) + 1e-9 * v, cos(a)); });                                  // atan2 spread quadrants (+ device sin/cos)
    out[k++] = chain(x, [&](double v) { return sin(a + 1e-9 * v) + cos(a); });                                            // device libm sin+cos alone
    out[k++] = chain(x, [&](double v) { return avp_pi_2_pi(a + 1e-9 * v + 2.0); });                                       // pi_2_pi
    out[k++] = chain(x, [&](double v) { return avp_M(a * 2.0 + 1e-9 * v); });                                             // M (python %)
    out[k++] = chain(x, [&](double v) { return avp_acos(0.9 * sin(a) + 1e-12 * v); });                                    // acos (+ device sin)
    out[k++] = chain(x, [&](double v) { return avp_hypot(a + 1e-9 * v, 1.3); });                                          // hypot
    out[k++] = chain(x, [&](double v) { double px, py, pyaw; rs_interpolate(0.4 + 1e-9 * v, 1 + (threadIdx.x & 1), 0.25, 1.0, 2.0, a, px, py, pyaw); return px + py + pyaw; });   // rs_interpolate (arc)
    out[k++] = chain(x, [&](double v) { const RsFrame f = rs_frame(1.0 + 1e-9 * v, 2.0, a, 6.0, -3.0, 0.5 * a, 0.25); return f.x0 + f.yb; });                                   // rs_frame
    for (int wd = 0; wd < 46; wd += 1) {
        const RsWord W = RS_WORDS[wd];
        if (wd > 0 && RS_WORDS[wd - 1].solver == W.solver) continue;      // one word per solver
        const RsFrame f = rs_frame(1.0, 2.0, a, 6.0 + 0.1 * threadIdx.x, -3.0, 0.5 * a, 0.25);
        long long t0 = clock64();
        double acc = 0;
        for (int i = 0; i < 16; i++) { double l[5]; RsFrame g = f; g.x0 += 1e-9 * acc; rs_word(wd, g, l); acc += l[0] + l[1]; }
        out[k++] = (clock64() - t0) * (N / 16);
        x += acc;
    }
    sink[threadIdx.x] = x;
}

int main()
{
    long long* d; double* s; long long h[32];
    hipMalloc(&d, sizeof(h)); hipMalloc(&s, 64 * sizeof(double));
    const char* names[] = { "sincos spread", "sincos uniform", "atan2 spread(+sin,cos)", "device sin+cos", "pi_2_pi", "M", "acos(+sin)", "hypot", "rs_interpolate arc", "rs_frame",
                            "word SLS", "word LSL", "word LSR", "word LRL", "word LRLRn", "word LRLRp", "word LRSL", "word LRSR", "word LRSLR" };
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, 0.0, d, s);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    }
    for (int i = 0; i < 19; i++) printf("%-24s %8.1f cycles/call\n", names[i], (double)h[i] / N);
    return 0;
}
