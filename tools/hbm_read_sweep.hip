// Read-only HBM sweep of the attention scan's byte count (788.5 MB): the ceiling the scan is priced against in DESIGN.md.
// hipcc --offload-arch=gfx950 -O3 -o hbm_read_sweep tools/hbm_read_sweep.hip && ./hbm_read_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// read-only streaming: each workgroup sums a contiguous chunk with 16-B loads, U loads in flight per thread
template <int U>
__global__ __launch_bounds__(256) void rd(const float4 *x, long long n4_per_wg, float *out) {
    const float4 *p = x + (long long)blockIdx.x * n4_per_wg;
    float4 a = make_float4(0, 0, 0, 0);
    for (long long i = threadIdx.x; i < n4_per_wg; i += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + 256 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) out[0] = a.x;
}
int main() {
    const long long bytes = 788529152LL;            // the scan's bytes per launch
    const long long n4 = bytes / 16;
    float4 *x; float *o;
    CK(hipMalloc(&x, bytes + (1 << 20))); CK(hipMalloc(&o, 4));
    CK(hipMemset(x, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wgs : {2048, 4096, 8192, 16384, 65536}) {
        const long long per = (n4 / wgs) / (256 * 4) * (256 * 4);
        for (int U : {2, 4, 8}) {
            auto launch = [&]() {
                if (U == 2) hipLaunchKernelGGL(rd<2>, dim3(wgs), dim3(256), 0, 0, x, per, o);
                if (U == 4) hipLaunchKernelGGL(rd<4>, dim3(wgs), dim3(256), 0, 0, x, per, o);
                if (U == 8) hipLaunchKernelGGL(rd<8>, dim3(wgs), dim3(256), 0, 0, x, per, o);
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / 20;
            printf("wgs %6d  U %d  %8.1f us  %7.1f GB/s\n", wgs, U, us, (double)per * wgs * 16 / us / 1e3);
        }
    }
    return 0;
}
