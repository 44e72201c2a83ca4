// Does a read-only buffer stay in the XCD L2s / the Infinity Cache from one launch to the next?  A grid of 2048 workgroups,
// workgroup b always sweeping the same contiguous slice (so the same XCD under round-robin placement), launched back to back;
// per-launch time and rate by buffer size.   hipcc --offload-arch=gfx950 -O3 -o tools/_lab/cache_sweep tools/cache_sweep_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void sweep(const float4 *p, long long n4_per_wg, float *sink) {
    const float4 *q = p + (long long)blockIdx.x * n4_per_wg;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long i = threadIdx.x; i < n4_per_wg; i += 256 * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i + u * 256 < n4_per_wg) ? q[i + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    if (a.x + a.y + a.z + a.w == 1.2345f) sink[0] = a.x;
}

int main() {
    setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
    const size_t maxb = 1536u << 20;
    float4 *buf; float *sink;
    CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, maxb));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int mbs[] = {2, 4, 8, 16, 24, 32, 48, 64, 128, 200, 400, 1024};
    for (int grid : {256, 2048}) {
        for (int mb : mbs) {
            const long long n4 = ((long long)mb << 20) / 16 / grid;
            const int reps = 200;
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(sweep, dim3(grid), dim3(256), 0, st, buf, n4, sink);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(sweep, dim3(grid), dim3(256), 0, st, buf, n4, sink);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / reps;
            printf("grid %4d  %5d MB: %8.2f us per launch  %7.2f TB/s\n", grid, mb, us, (double)mb * 1048576.0 / us / 1e6);
        }
    }
    return 0;
}
