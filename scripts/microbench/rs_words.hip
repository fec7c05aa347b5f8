// Single-wave cost of every Reeds-Shepp word solver (cycles per call, 64 lanes with different queries):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I include -I automatedvaletparking_amd/csrc \
//         scripts/microbench/rs_words.hip -o scripts/microbench/rs_words
// Feeds the LPT weights of pl_rs_build_schedule (avp_plan_kernels.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "avp_device.h"
#include "avp_rs_kernels.h"

__global__ void probe(long long* out, int* okc, double* sink)
{
    avp_lds_tables_fill<true>();
    rs_lds_tables_fill();
    const int lane = threadIdx.x;
    // queries like the planner's: goal 2..14 m away, any heading
    const double ang = 0.37 * lane, dist = 2.0 + 0.19 * lane;
    const RsFrame f = rs_frame(0.0, 0.0, 0.3 + 0.05 * lane, dist * cos(ang), dist * sin(ang), -1.0 + 0.11 * lane, 1.0 / 3.9765932159382564);
    double acc = 0.0;
    for (int w = 0; w < 46; w++) {
        double l[5] = { 0, 0, 0, 0, 0 };
        int ok = 0;
        const long long t0 = clock64();
        for (int rep = 0; rep < 8; rep++) { ok += rs_word(w, f, l) ? 1 : 0; acc += l[0] + l[1] + l[2] + l[3] + l[4]; }
        const long long t1 = clock64();
        if (lane == 0) out[w] = (t1 - t0) / 8;
        atomicAdd(&okc[w], ok / 8);
    }
    sink[lane] = acc;
}

int main()
{
    long long* d; int* okc; double* s; long long h[46]; int hk[46];
    hipMalloc(&d, sizeof(h)); hipMalloc(&okc, sizeof(hk)); hipMalloc(&s, 64 * sizeof(double));
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(okc, 0, sizeof(hk));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, okc, s);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        hipMemcpy(hk, okc, sizeof(hk), hipMemcpyDeviceToHost);
    }
    for (int w = 0; w < 46; w++) printf("word %2d  %8lld cycles/call  ok lanes %2d\n", w, h[w], hk[w]);
    return 0;
}
