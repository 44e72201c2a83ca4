// Lab harness (stand-alone: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o h3_gemm_lab tools/h3_gemm_lab.hip) for the
// split-f16 GEMM kernels of csrc/gemm_f32.hip: times tile / staging variants on the decoder's shapes and checks them
// against fp64.  Round-2 variants: v6 = 256x128 tile, 8 waves, DMA pieces of 16 rows x 64 B (the round-1 kernel);
// v8 = the same with whole-line pieces of 8 rows x 128 B (adopted: -9..-14 %); v10 = 4 compute waves (64x128, ONE
// accumulator set, cross-term operands scaled in registers) + 4 loader waves (-3..-8 % more; not adopted);
// v12 = 256x256 tile, single accumulator (no gain: 2.4 rounds of one workgroup per CU).  Environment: LAB_SHAPE=0|1
// (classifier / lang-LSTM shape only), LAB_ONLY=<variant>, LAB_IT=<timed launches>, LAB_ZERO=1 (all-zero operands:
// the same binaries run 35-45 % faster - the random-data ceiling of these kernels is the chip's power management,
// MI355X_MICROARCH.md "DVFS give-back").  tools/pmc_lab.sh collects SQ / TCC / TCP counters for one variant.
//
// fp32-accurate GEMM on the f16 matrix cores by operand splitting.
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
    // interleaved: element i = (row, k) with K % 32 == 0 -> block (i >> 5) of 64 halfs: [hi 32 | lo 32]
    const long long o = (i >> 5) * 64 + (i & 31);
    *reinterpret_cast<uint2 *>(hi + o) = *reinterpret_cast<uint2 *>(h);
    *reinterpret_cast<uint2 *>(hi + o + 32) = *reinterpret_cast<uint2 *>(l);
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
        src[0 + i] = Ah + (long long)(row0 + r) * 2 * K + q * 8;
        src[2 + i] = Ah + (long long)(row0 + r) * 2 * K + q * 8 + 32;
    }
    {
        const int r = 16 * w + (lane >> 2);
        const int q = (lane & 3) ^ ((r >> 2) & 3);
        src[4] = Wh + (long long)(col0 + r) * 2 * K + q * 8;
        src[5] = Wh + (long long)(col0 + r) * 2 * K + q * 8 + 32;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    const unsigned wv = __builtin_amdgcn_readfirstlane((unsigned)w);
    auto dma1 = [&](unsigned dst, const _Float16 *&p) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(p) : "memory");
        p += 64;
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


// variant F: the same 256x128 tile, but every DMA piece is 8 rows x 128 B = whole lines of the interleaved plane
// layout (hi and lo of a 32-k block together); LDS image row = 128 B, position p of row r holds chunk p ^ ((r>>1)&7)
__global__ __launch_bounds__(512) void gemm_h3xf(const _Float16 *Ah, const _Float16 *Al, const _Float16 *Wh,
                                                 const _Float16 *Wl, float *C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PA = 256 * 128, PB = 128 * 128, ST = PA + PB;      // 48 KB per buffer
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int tm, tn;
    tile_coords(M / 256, N / 128, tm, tn);
    const int row0 = tm * 256, col0 = tn * 128;
    f32x16 acc0[4], acc1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[j][r] = 0.f; acc1[j][r] = 0.f; }
    // wave w: A pieces 4w..4w+3 (rows 32w + 8i + (lane>>3)), W pieces 2w, 2w+1 (rows 16w + 8i + (lane>>3))
    const char *src[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 32 * w + 8 * i + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        src[i] = reinterpret_cast<const char *>(Ah + (long long)(row0 + r) * 2 * K) + c * 16;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * w + 8 * i + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        src[4 + i] = reinterpret_cast<const char *>(Wh + (long long)(col0 + r) * 2 * K) + c * 16;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    const unsigned wv = __builtin_amdgcn_readfirstlane((unsigned)w);
    auto dma1 = [&](unsigned dst, const char *&p) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(p) : "memory");
        p += 128;
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        const unsigned b = lds0 + buf * ST;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma1(b + (4 * wv + i) * 1024, src[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) dma1(b + PA + (2 * wv + i) * 1024, src[4 + i]);
    };
    const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
    h8 a1[2], a2[2], b1[2][4], b2[2][4];
    auto lfrag = [&](int buf, auto ksc, auto setc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value, S = decltype(setc)::value;
        const int ph = ((2 * ks + fh) ^ fsw) * 16, pl = ((4 + 2 * ks + fh) ^ fsw) * 16;
        const char *base = smem + buf * ST;
        const int ra = (w * 32 + fr) * 128;
        a1[S] = *reinterpret_cast<const h8 *>(base + ra + ph);
        a2[S] = *reinterpret_cast<const h8 *>(base + ra + pl);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rb = PA + (j * 32 + fr) * 128;
            b1[S][j] = *reinterpret_cast<const h8 *>(base + rb + ph);
            b2[S][j] = *reinterpret_cast<const h8 *>(base + rb + pl);
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


// variant P: 256x128 tile, 4 COMPUTE waves (64x128 each, ONE accumulator set: the cross terms are brought to the main
// term's scale in registers, hi' = hi * 2^-5, lo' = lo * 2^-6, so hi*hi + hi'*lo' + lo'*hi' shares an accumulator) and
// 4 LOADER waves that only issue the LDS-DMA pieces (full lines) - compute waves never stall on DMA issue.
__global__ __launch_bounds__(512) void gemm_h3p(const _Float16 *Ah, const _Float16 *Al, const _Float16 *Wh,
                                                const _Float16 *Wl, float *C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PA = 256 * 128, PB = 128 * 128, ST = PA + PB;      // 48 KB per buffer, 3 buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    tile_coords(M / 256, N / 128, tm, tn);
    const int row0 = tm * 256, col0 = tn * 128;
    const int n = K / 32;
    if (w >= 4) {
        // ---- loader wave l: A pieces 8l .. 8l+7 (rows 64l + 8i + (lane>>3)), W pieces 4l .. 4l+3 (rows 32l + 8i + (lane>>3))
        const int l = w - 4;
        const char *srcA[8], *srcW[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 64 * l + 8 * i + (lane >> 3);
            srcA[i] = reinterpret_cast<const char *>(Ah + (long long)(row0 + r) * 2 * K) + ((lane & 7) ^ ((r >> 1) & 7)) * 16;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 32 * l + 8 * i + (lane >> 3);
            srcW[i] = reinterpret_cast<const char *>(Wh + (long long)(col0 + r) * 2 * K) + ((lane & 7) ^ ((r >> 1) & 7)) * 16;
        }
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
        auto stage = [&](int buf) __attribute__((always_inline)) {
            const unsigned b = lds0 + buf * ST;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(b + (8 * l + i) * 1024), "v"(srcA[i]) : "memory");
                srcA[i] += 128;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(b + PA + (4 * l + i) * 1024), "v"(srcW[i]) : "memory");
                srcW[i] += 128;
            }
        };
        stage(0);
        if (n > 1) stage(1);
        if (n > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // chunk 0 landed
        for (int c = 0; c < n; ++c) {
            if (c + 2 < n) stage((c + 2) % 3);             // its buffer held chunk c-1: consumers left it before the last barrier
            if (c + 1 < n) {
                if (c + 2 < n) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();                               // chunk c+1 landed; consumers are done with chunk c
        }
        return;
    }
    // ---- compute wave w: rows 64w .. 64w+63, all 128 columns
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
    const h8 s5 = {(_Float16)0.03125f, (_Float16)0.03125f, (_Float16)0.03125f, (_Float16)0.03125f,
                   (_Float16)0.03125f, (_Float16)0.03125f, (_Float16)0.03125f, (_Float16)0.03125f};
    const h8 s6 = s5 * (_Float16)0.5f;
    __syncthreads();                                       // chunk 0 landed
    for (int c = 0; c < n; ++c) {
        const char *base = smem + (c % 3) * ST;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ph = ((2 * ks + fh) ^ fsw) * 16, pl = ((4 + 2 * ks + fh) ^ fsw) * 16;
            h8 a1[2], a1s[2], a2s[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = (w * 64 + i * 32 + fr) * 128;
                a1[i] = *reinterpret_cast<const h8 *>(base + ra + ph);
                const h8 lo = *reinterpret_cast<const h8 *>(base + ra + pl);
                a1s[i] = a1[i] * s5;
                a2s[i] = lo * s6;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb = PA + (j * 32 + fr) * 128;
                const h8 b1 = *reinterpret_cast<const h8 *>(base + rb + ph);
                const h8 b2 = *reinterpret_cast<const h8 *>(base + rb + pl);
                const h8 b1s = b1 * s5, b2s = b2 * s6;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[i], b1, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1s[i], b2s, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2s[i], b1s, acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + w * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = col0 + j * 32 + (lane & 31);
                C[(long long)row * N + col] = acc[i][j][r];
            }
}


// variant Q: 256x256 tile, 8 waves (4 in M x 2 in N, 64x128 each), ONE accumulator set (in-register 2^-5 / 2^-6 scaling of
// the cross-term operands), full-line pieces, 2 buffers of 64 KB; every wave issues 8 DMA pieces per chunk.
__global__ __launch_bounds__(512) void gemm_h3q(const _Float16 *Ah, const _Float16 *Al, const _Float16 *Wh,
                                                const _Float16 *Wl, float *C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PA = 256 * 128, PB = 256 * 128, ST = PA + PB;      // 64 KB per buffer
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
    int tm, tn;
    tile_coords(M / 256, N / 256, tm, tn);
    const int row0 = tm * 256, col0 = tn * 256;
    const int n = K / 32;
    // wave w: A pieces 4w..4w+3 (rows 32w + 8i + (lane>>3)), W pieces 4w..4w+3
    const char *srcA[4], *srcW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 32 * w + 8 * i + (lane >> 3);
        const int c = ((lane & 7) ^ ((r >> 1) & 7)) * 16;
        srcA[i] = reinterpret_cast<const char *>(Ah + (long long)(row0 + r) * 2 * K) + c;
        srcW[i] = reinterpret_cast<const char *>(Wh + (long long)(col0 + r) * 2 * K) + c;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    auto stage = [&](int buf) __attribute__((always_inline)) {
        const unsigned b = lds0 + buf * ST;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(b + (4 * w + i) * 1024), "v"(srcA[i]) : "memory");
            srcA[i] += 128;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(b + PA + (4 * w + i) * 1024), "v"(srcW[i]) : "memory");
            srcW[i] += 128;
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
    const h8 s5 = {(_Float16)0.03125f, (_Float16)0.03125f, (_Float16)0.03125f, (_Float16)0.03125f,
                   (_Float16)0.03125f, (_Float16)0.03125f, (_Float16)0.03125f, (_Float16)0.03125f};
    const h8 s6 = s5 * (_Float16)0.5f;
    stage(0);
    for (int c = 0; c < n; ++c) {
        if (c + 1 < n) {
            stage((c + 1) & 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                   // chunk c landed everywhere
        const char *base = smem + (c & 1) * ST;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ph = ((2 * ks + fh) ^ fsw) * 16, pl = ((4 + 2 * ks + fh) ^ fsw) * 16;
            h8 a1[2], a1s[2], a2s[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = (wm * 64 + i * 32 + fr) * 128;
                a1[i] = *reinterpret_cast<const h8 *>(base + ra + ph);
                const h8 lo = *reinterpret_cast<const h8 *>(base + ra + pl);
                a1s[i] = a1[i] * s5;
                a2s[i] = lo * s6;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb = PA + (wn * 128 + j * 32 + fr) * 128;
                const h8 b1 = *reinterpret_cast<const h8 *>(base + rb + ph);
                const h8 b2 = *reinterpret_cast<const h8 *>(base + rb + pl);
                const h8 b1s = b1 * s5, b2s = b2 * s6;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[i], b1, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1s[i], b2s, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2s[i], b1s, acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                   // everyone is done with buffer c & 1 before it is refilled
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = col0 + wn * 128 + j * 32 + (lane & 31);
                C[(long long)row * N + col] = acc[i][j][r];
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
    if (getenv("LAB_ZERO")) { for (auto &x : hA) x = 0.f; for (auto &x : hW) x = 0.f; }
    float *dA, *dW, *dC, *dR;
    _Float16 *Ah, *Al, *Wh, *Wl;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&Ah, hA.size() * 4)); CK(hipMalloc(&Al, hA.size() * 2));
    CK(hipMalloc(&Wh, hW.size() * 4)); CK(hipMalloc(&Wl, hW.size() * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto split = [&](const float *x, _Float16 *h, _Float16 *l, size_t n) {
        split_kernel<<<(unsigned)((n / 4 + 255) / 256), 256>>>(x, h, l, (long long)n);
    };
    split(dA, Ah, Al, hA.size()); split(dW, Wh, Wl, hW.size());
    CK(hipDeviceSynchronize());
    CK(hipFuncSetAttribute((const void *)gemm_h3x, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
    CK(hipFuncSetAttribute((const void *)gemm_h3xf, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
    CK(hipFuncSetAttribute((const void *)gemm_h3p, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
    CK(hipFuncSetAttribute((const void *)gemm_h3q, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536));
    const int only = getenv("LAB_ONLY") ? atoi(getenv("LAB_ONLY")) : 0;
    for (int nb = 6; nb <= 12; nb += 2) {
        if (only && nb != only) continue;
        auto launch = [&]() {
            if (nb == 6) gemm_h3x<<<(M / 256) * (N / 128), 512, 3 * 49152>>>(Ah, Al, Wh, Wl, dC, M, N, K);
            else if (nb == 8) gemm_h3xf<<<(M / 256) * (N / 128), 512, 3 * 49152>>>(Ah, Al, Wh, Wl, dC, M, N, K);
            else if (nb == 10) gemm_h3p<<<(M / 256) * (N / 128), 512, 3 * 49152>>>(Ah, Al, Wh, Wl, dC, M, N, K);
            else { if (N % 256) return; gemm_h3q<<<(M / 256) * (N / 256), 512, 2 * 65536>>>(Ah, Al, Wh, Wl, dC, M, N, K); }
        };
        const int it = getenv("LAB_IT") ? atoi(getenv("LAB_IT")) : 50;
        for (int i = 0; i < (it < 20 ? 2 : 20); ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < it; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / it;
        printf("M=%d N=%d K=%d  v%d (6: 256x128 half-line pieces; 8: 256x128 full-line pieces; 10: 4 compute + 4 loader waves, single accumulator; 12: 256x256 single accumulator): %.1f us  %.1f TF(fp32-eq)  frac-of-833 %.3f\n", M, N, K, nb, us,
               2.0 * M * N * K / us * 1e-6, 2.0 * M * N * K / us * 1e-6 / 833.3);
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
    const int shape = getenv("LAB_SHAPE") ? atoi(getenv("LAB_SHAPE")) : -1;
    if (shape == 0) { run_shape(4096, 9984, 512, 0.05f, 0); return 0; }
    if (shape == 1) { run_shape(4096, 2048, 1536, 0.05f, 0); return 0; }
    run_shape(4096, 9984, 512, 0.05f, 0);      // classifier
    run_shape(4096, 2048, 1536, 0.05f, 0);     // lang-LSTM
    run_shape(4096, 2048, 2048, 0.05f, 0);     // att-LSTM (no table)
    run_shape(4096, 512, 512, 0.05f, 0);       // h projections
    run_shape(4096, 2048, 1536, 0.002f, 1);    // small weights, wide activations
    return 0;
}
