// Lab: fp32-accurate GEMM on the f16 matrix cores by operand splitting.
//   x = hi + lo * 2^-11  with hi = f16(x), lo = f16((x - hi) * 2^11)   (>= 22 significant bits for |x| >= 2^-14)
//   C = sum hi_a*hi_b  +  2^-11 * sum (hi_a*lo_b + lo_a*hi_b)          (lo*lo dropped: <= 2^-22 relative)
// C[M,N] = A[M,K] * W[N,K]^T; 128x128 tile, 4 waves (2x2) of 64x64, BK = 32 halfs, LDS-DMA, 2 buffers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <random>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
using I0 = std::integral_constant<int, 0>;
using I1 = std::integral_constant<int, 1>;

__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int &tm, int &tn) {
    const int nt = tiles_m * tiles_n, bid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7, j = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    tm = logical / tiles_n;
    tn = logical % tiles_n;
}

__global__ void split_kernel(const float *x, _Float16 *hi, _Float16 *lo, long long n) {
    long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 v = *reinterpret_cast<const float4 *>(x + i);
    const float a[4] = {v.x, v.y, v.z, v.w};
    _Float16 h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (_Float16)a[e];
        l[e] = (_Float16)((a[e] - (float)h[e]) * 2048.f);
    }
    *reinterpret_cast<uint2 *>(hi + i) = *reinterpret_cast<uint2 *>(h);
    *reinterpret_cast<uint2 *>(lo + i) = *reinterpret_cast<uint2 *>(l);
}

// LDS per stage: planes A_hi, A_lo, B_hi, B_lo, each [128 rows][4 slots of 16 B]; slot s of row r holds
// k-quad (8 halfs) q = s ^ ((r>>2)&3)
template <int NBUF, int TI, int TJ>
__global__ __launch_bounds__(256, 2) void gemm_h3(const _Float16 *Ah, const _Float16 *Al, const _Float16 *Wh,
                                                  const _Float16 *Wl, float *C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PL = 128 * 64;                    // bytes per plane tile
    constexpr int ST = 4 * PL;                      // bytes per stage
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = TI == 2 ? w >> 1 : w, wn = TI == 2 ? w & 1 : 0;
    int tm, tn;
    tile_coords(M / 128, N / 128, tm, tn);
    const int row0 = tm * 128, col0 = tn * 128;
    f32x16 acc0[TI][TJ], acc1[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    // staging: per plane, wave w issues instructions ii = 2w, 2w+1; instruction ii covers rows 16*ii + (lane>>2)
    const _Float16 *src[8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * (2 * w + i) + (lane >> 2);
        const int q = (lane & 3) ^ ((r >> 2) & 3);
        src[0 + i] = Ah + (long long)(row0 + r) * K + q * 8;
        src[2 + i] = Al + (long long)(row0 + r) * K + q * 8;
        src[4 + i] = Wh + (long long)(col0 + r) * K + q * 8;
        src[6 + i] = Wl + (long long)(col0 + r) * K + q * 8;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + w * 2048);
    auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off"
                             :: "s"(lds0 + buf * ST + p * PL + i * 1024), "v"(src[2 * p + i]) : "memory");
                src[2 * p + i] += 32;
            }
    };
    const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 2) & 3;
    h8 a1[2][TI], a2[2][TI], b1[2][TJ], b2[2][TJ];      // [set][tile]
    auto lfrag = [&](int buf, auto ksc, auto setc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value, S = decltype(setc)::value;
        const int slot = ((2 * ks + fh) ^ fsw) * 16;
        const char *base = smem + buf * ST;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int ra = (wm * 32 * TI + i * 32 + fr) * 64 + slot;
            a1[S][i] = *reinterpret_cast<const h8 *>(base + ra);
            a2[S][i] = *reinterpret_cast<const h8 *>(base + PL + ra);
        }
#pragma unroll
        for (int i = 0; i < TJ; ++i) {
            const int rb = (wn * 32 * TJ + i * 32 + fr) * 64 + slot;
            b1[S][i] = *reinterpret_cast<const h8 *>(base + 2 * PL + rb);
            b2[S][i] = *reinterpret_cast<const h8 *>(base + 3 * PL + rb);
        }
    };
    auto mma = [&](auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[S][i], b1[S][j], acc0[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[S][i], b2[S][j], acc1[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[S][i], b1[S][j], acc1[i][j], 0, 0, 0);
            }
    };
    const int n = K / 32;
    if constexpr (NBUF == 2) {
        stage(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        lfrag(0, I0{}, I0{});
        for (int c = 0; c < n; ++c) {
            const int cur = c & 1, nxt = cur ^ 1;
            const bool has1 = c + 1 < n;
            if (has1) stage(nxt);
            lfrag(cur, I1{}, I1{});
            mma(I0{});
            if (has1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); lfrag(nxt, I0{}, I0{}); }
            mma(I1{});
        }
    } else {
        // 3 buffers: chunk c in LDS[c%3]; DMA of chunk c+2 issued at the top of chunk c
        stage(0);
        if (n > 1) stage(1);
        if (n > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        lfrag(0, I0{}, I0{});
        int cur = 0;
        for (int c = 0; c < n; ++c) {
            const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;
            const bool has1 = c + 1 < n, has2 = c + 2 < n;
            if (has2) stage(nn);
            lfrag(cur, I1{}, I1{});
            mma(I0{});
            if (has1) {
                if (has2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                lfrag(nxt, I0{}, I0{});
            }
            mma(I1{});
            cur = nxt;
        }
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = col0 + wn * 32 * TJ + j * 32 + (lane & 31);
                C[(long long)row * N + col] = acc0[i][j][r] + acc1[i][j][r] * (1.f / 2048.f);
            }
}

// variant X: 256x128 tile, 8 waves (each 32x128), 3 buffers of 48 KB, two chunks in flight, one workgroup per CU
__global__ __launch_bounds__(512) void gemm_h3x(const _Float16 *Ah, const _Float16 *Al, const _Float16 *Wh,
                                                const _Float16 *Wl, float *C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PA = 256 * 64, PB = 128 * 64, ST = 2 * PA + 2 * PB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int tm, tn;
    tile_coords(M / 256, N / 128, tm, tn);
    const int row0 = tm * 256, col0 = tn * 128;
    f32x16 acc0[4], acc1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[j][r] = 0.f; acc1[j][r] = 0.f; }
    const _Float16 *src[6];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * (2 * w + i) + (lane >> 2);
        const int q = (lane & 3) ^ ((r >> 2) & 3);
        src[0 + i] = Ah + (long long)(row0 + r) * K + q * 8;
        src[2 + i] = Al + (long long)(row0 + r) * K + q * 8;
    }
    {
        const int r = 16 * w + (lane >> 2);
        const int q = (lane & 3) ^ ((r >> 2) & 3);
        src[4] = Wh + (long long)(col0 + r) * K + q * 8;
        src[5] = Wl + (long long)(col0 + r) * K + q * 8;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    const unsigned wv = __builtin_amdgcn_readfirstlane((unsigned)w);
    auto dma1 = [&](unsigned dst, const _Float16 *&p) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(p) : "memory");
        p += 32;
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        const unsigned b = lds0 + buf * ST;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            dma1(b + (2 * wv + i) * 1024, src[0 + i]);
            dma1(b + PA + (2 * wv + i) * 1024, src[2 + i]);
        }
        dma1(b + 2 * PA + wv * 1024, src[4]);
        dma1(b + 2 * PA + PB + wv * 1024, src[5]);
    };
    const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 2) & 3;
    h8 a1[2], a2[2], b1[2][4], b2[2][4];
    auto lfrag = [&](int buf, auto ksc, auto setc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value, S = decltype(setc)::value;
        const int slot = ((2 * ks + fh) ^ fsw) * 16;
        const char *base = smem + buf * ST;
        const int ra = (w * 32 + fr) * 64 + slot;
        a1[S] = *reinterpret_cast<const h8 *>(base + ra);
        a2[S] = *reinterpret_cast<const h8 *>(base + PA + ra);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rb = (j * 32 + fr) * 64 + slot;
            b1[S][j] = *reinterpret_cast<const h8 *>(base + 2 * PA + rb);
            b2[S][j] = *reinterpret_cast<const h8 *>(base + 2 * PA + PB + rb);
        }
    };
    auto mma = [&](auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[S], b1[S][j], acc0[j], 0, 0, 0);
            acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[S], b2[S][j], acc1[j], 0, 0, 0);
            acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[S], b1[S][j], acc1[j], 0, 0, 0);
        }
    };
    const int n = K / 32;
    auto wait_for = [&](int in_flight) __attribute__((always_inline)) {
        if (in_flight >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    stage(0);
    if (n > 1) stage(1);
    wait_for(n > 1 ? 1 : 0);
    __syncthreads();
    lfrag(0, I0{}, I0{});
    int cur = 0;
    for (int c = 0; c < n; ++c) {
        const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;
        if (c + 2 < n) stage(nn);
        lfrag(cur, I1{}, I1{});
        mma(I0{});
        if (c + 1 < n) {
            wait_for(c + 2 < n ? 1 : 0);
            __syncthreads();
            lfrag(nxt, I0{}, I0{});
        }
        mma(I1{});
        cur = nxt;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = col0 + j * 32 + (lane & 31);
            C[(long long)row * N + col] = acc0[j][r] + acc1[j][r] * (1.f / 2048.f);
        }
}

// variant 7: 128x128, 4 waves (32x128), 16-deep sub-chunks of 16 KB, four buffers, three sub-chunks in flight, 2 WG/CU
__global__ __launch_bounds__(256, 2) void gemm_h3s(const _Float16 *Ah, const _Float16 *Al, const _Float16 *Wh,
                                                   const _Float16 *Wl, float *C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PL = 128 * 32, ST = 4 * PL;      // 4 KB planes, 16 KB per buffer
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int tm, tn;
    tile_coords(M / 128, N / 128, tm, tn);
    const int row0 = tm * 128, col0 = tn * 128;
    f32x16 acc0[4], acc1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[j][r] = 0.f; acc1[j][r] = 0.f; }
    // piece w of each plane: rows 32w + (lane>>1), slot lane&1 holds k-octet (lane&1) ^ ((row>>3)&1)
    const _Float16 *src[4];
    {
        const int r = 32 * w + (lane >> 1);
        const int q = (lane & 1) ^ ((r >> 3) & 1);
        src[0] = Ah + (long long)(row0 + r) * K + q * 8;
        src[1] = Al + (long long)(row0 + r) * K + q * 8;
        src[2] = Wh + (long long)(col0 + r) * K + q * 8;
        src[3] = Wl + (long long)(col0 + r) * K + q * 8;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + w * 1024);
    auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off"
                         :: "s"(lds0 + buf * ST + p * PL), "v"(src[p]) : "memory");
            src[p] += 16;
        }
    };
    const int fr = lane & 31, fh = lane >> 5;
    h8 a1[2], a2[2], b1[2][4], b2[2][4];
    auto lfrag = [&](int buf, auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
        const int slot = (fh ^ ((fr >> 3) & 1)) * 16;
        const char *base = smem + buf * ST;
        const int ra = (w * 32 + fr) * 32 + slot;
        a1[S] = *reinterpret_cast<const h8 *>(base + ra);
        a2[S] = *reinterpret_cast<const h8 *>(base + PL + ra);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rb = (j * 32 + fr) * 32 + slot;
            b1[S][j] = *reinterpret_cast<const h8 *>(base + 2 * PL + rb);
            b2[S][j] = *reinterpret_cast<const h8 *>(base + 3 * PL + rb);
        }
    };
    auto mma = [&](auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[S], b1[S][j], acc0[j], 0, 0, 0);
            acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[S], b2[S][j], acc1[j], 0, 0, 0);
            acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[S], b1[S][j], acc1[j], 0, 0, 0);
        }
    };
    const int n = K / 16;                           // sub-chunks (n >= 4 and even assumed)
    auto wait_for = [&](int in_flight) __attribute__((always_inline)) {
        if (in_flight >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (in_flight == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    stage(0); stage(1); stage(2);
    wait_for(2);
    __syncthreads();
    lfrag(0, I0{});
    for (int c = 0; c < n; c += 2) {
        // sub-chunk c (set 0)
        if (c + 3 < n) stage((c + 3) & 3);
        {
            const int last = n - 1 < c + 3 ? n - 1 : c + 3;
            wait_for(last - (c + 1));
            __syncthreads();
            lfrag((c + 1) & 3, I1{});
        }
        mma(I0{});
        // sub-chunk c+1 (set 1)
        if (c + 4 < n) stage((c + 4) & 3);
        if (c + 2 < n) {
            const int last = n - 1 < c + 4 ? n - 1 : c + 4;
            wait_for(last - (c + 2));
            __syncthreads();
            lfrag((c + 2) & 3, I0{});
        }
        mma(I1{});
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = col0 + j * 32 + (lane & 31);
            C[(long long)row * N + col] = acc0[j][r] + acc1[j][r] * (1.f / 2048.f);
        }
}

// plain fp32 reference GEMM (one thread per output, fmaf chain) for the accuracy comparison
__global__ void gemm_f32_ref(const float *A, const float *W, float *C, int M, int N, int K, const int *rows, int nrows) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, ri = blockIdx.y;
    if (c >= N || ri >= nrows) return;
    const float *a = A + (long long)rows[ri] * K, *b = W + (long long)c * K;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(a[k], b[k], s);
    C[(long long)ri * N + c] = s;
}

static void run_shape(int M, int N, int K, float wscale, int dist) {
    std::mt19937 rng(1234);
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::normal_distribution<float> g(0.f, 1.f);
    for (auto &x : hA) x = dist == 0 ? u(rng) : g(rng) * 3.f;
    for (auto &x : hW) x = g(rng) * wscale;
    float *dA, *dW, *dC, *dR;
    _Float16 *Ah, *Al, *Wh, *Wl;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&Ah, hA.size() * 2)); CK(hipMalloc(&Al, hA.size() * 2));
    CK(hipMalloc(&Wh, hW.size() * 2)); CK(hipMalloc(&Wl, hW.size() * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto split = [&](const float *x, _Float16 *h, _Float16 *l, size_t n) {
        split_kernel<<<(unsigned)((n / 4 + 255) / 256), 256>>>(x, h, l, (long long)n);
    };
    split(dA, Ah, Al, hA.size()); split(dW, Wh, Wl, hW.size());
    CK(hipDeviceSynchronize());
    const int grid = (M / 128) * (N / 128);
    CK(hipFuncSetAttribute((const void *)gemm_h3<3, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768));
    CK(hipFuncSetAttribute((const void *)gemm_h3<2, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768));
    CK(hipFuncSetAttribute((const void *)gemm_h3<3, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768));
    CK(hipFuncSetAttribute((const void *)gemm_h3<2, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768));
    CK(hipFuncSetAttribute((const void *)gemm_h3x, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
    CK(hipFuncSetAttribute((const void *)gemm_h3s, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int nb = 4; nb <= 7; ++nb) {
        auto launch = [&]() {
            if (nb == 7) { gemm_h3s<<<grid, 256, 65536>>>(Ah, Al, Wh, Wl, dC, M, N, K); return; }
            if (nb == 6) { gemm_h3x<<<(M / 256) * (N / 128), 512, 3 * 49152>>>(Ah, Al, Wh, Wl, dC, M, N, K); return; }
            if (nb == 2) gemm_h3<2, 2, 2><<<grid, 256, 2 * 32768>>>(Ah, Al, Wh, Wl, dC, M, N, K);
            else if (nb == 3) gemm_h3<3, 2, 2><<<grid, 256, 3 * 32768>>>(Ah, Al, Wh, Wl, dC, M, N, K);
            else if (nb == 4) gemm_h3<2, 1, 4><<<grid, 256, 2 * 32768>>>(Ah, Al, Wh, Wl, dC, M, N, K);
            else gemm_h3<3, 1, 4><<<grid, 256, 3 * 32768>>>(Ah, Al, Wh, Wl, dC, M, N, K);
        };
        for (int i = 0; i < 20; ++i) launch();
        CK(hipDeviceSynchronize());
        const int it = 50;
        CK(hipEventRecord(e0));
        for (int i = 0; i < it; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / it;
        printf("M=%d N=%d K=%d  h3 v%d (2,3: 2x2 waves nbuf 2,3; 4,5: 4x1 waves; 6: 256x128 8 waves 3 buf; 7: 128x128 16-deep x4 buf): %.1f us  %.1f TF(fp32-equivalent)  %.1f TF(f16 executed)\n", M, N, K, nb, us,
               2.0 * M * N * K / us * 1e-6, 6.0 * M * N * K / us * 1e-6);
    }
    // split cost
    {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) { split(dA, Ah, Al, hA.size()); split(dW, Wh, Wl, hW.size()); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("    split A+W: %.1f us\n", ms * 1000.0 / 20);
    }
    // accuracy on 8 rows against fp64, next to an fp32 fmaf chain
    const int nr = 8;
    std::vector<int> rows(nr);
    for (int i = 0; i < nr; ++i) rows[i] = (int)((long long)i * 523 % M);
    int *drows; CK(hipMalloc(&drows, nr * 4)); CK(hipMemcpy(drows, rows.data(), nr * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dR, (size_t)nr * N * 4));
    gemm_f32_ref<<<dim3((N + 255) / 256, nr), 256>>>(dA, dW, dR, M, N, K, drows, nr);
    std::vector<float> hC((size_t)M * N), hR((size_t)nr * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hR.data(), dR, hR.size() * 4, hipMemcpyDeviceToHost));
    double e_h3 = 0, e_f32 = 0, s_h3 = 0, s_f32 = 0, mag = 0;
    for (int i = 0; i < nr; ++i)
        for (int c = 0; c < N; ++c) {
            double ref = 0, absum = 0;
            const float *a = &hA[(size_t)rows[i] * K], *b = &hW[(size_t)c * K];
            for (int k = 0; k < K; ++k) { ref += (double)a[k] * b[k]; absum += std::fabs((double)a[k] * b[k]); }
            const double d1 = std::fabs(hC[(size_t)rows[i] * N + c] - ref), d2 = std::fabs(hR[(size_t)i * N + c] - ref);
            e_h3 = std::max(e_h3, d1); e_f32 = std::max(e_f32, d2);
            s_h3 += d1 * d1; s_f32 += d2 * d2; mag = std::max(mag, std::fabs(ref));
        }
    printf("    vs fp64: h3 max %.3e rms %.3e | f32 fmaf max %.3e rms %.3e | max|C| %.3f\n", e_h3,
           std::sqrt(s_h3 / (nr * (double)N)), e_f32, std::sqrt(s_f32 / (nr * (double)N)), mag);
    hipFree(dA); hipFree(dW); hipFree(dC); hipFree(dR); hipFree(Ah); hipFree(Al); hipFree(Wh); hipFree(Wl); hipFree(drows);
}

int main() {
    run_shape(4096, 9984, 512, 0.05f, 0);      // classifier
    run_shape(4096, 2048, 1536, 0.05f, 0);     // lang-LSTM
    run_shape(4096, 2048, 2048, 0.05f, 0);     // att-LSTM (no table)
    run_shape(4096, 512, 512, 0.05f, 0);       // h projections
    run_shape(4096, 2048, 1536, 0.002f, 1);    // small weights, wide activations
    return 0;
}
