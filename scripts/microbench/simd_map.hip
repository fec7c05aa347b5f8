// Which waves of a 512-thread workgroup share a SIMD? Two waves (a, b) run a dependent fp64 chain, the others
// idle at the barrier; the pair's time doubles when both sit on the same SIMD.
//   hipcc -O3 --offload-arch=gfx950 scripts/microbench/simd_map.hip -o scripts/microbench/simd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void probe(int a, int b, long long* out, double* sink)
{
    const int wave = threadIdx.x >> 6;
    double x = 1.0 + 1e-3 * threadIdx.x;
    __syncthreads();
    const long long t0 = clock64();
    if (wave == a || wave == b || (b < 0 && wave < -b))
        for (int i = 0; i < 20000; i++) x = __builtin_fma(x, 1.0000001, 1e-9);
    const long long t1 = clock64();
    __syncthreads();
    if (threadIdx.x == a * 64) out[0] = t1 - t0;
    sink[threadIdx.x] = x;
}
int main()
{
    long long* d; double* s; long long h;
    hipMalloc(&d, 8); hipMalloc(&s, 512 * 8);
    for (int b = 0; b < 8; b++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, 0, b, d, s);
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("waves (0,%d): %lld cycles\n", b, h);
    }
    for (int n = 1; n <= 8; n++) {                 // waves 0 .. n-1 busy
        hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, 0, -n, d, s);
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("waves 0..%d busy: %lld cycles\n", n - 1, h);
    }
    return 0;
}
