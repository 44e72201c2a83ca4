// Round-3 lab (MFMA shape): variant F of tools/h3_gemm_lab.hip (= the production 256x128 eight-wave kernel's main loop)
// against the same tile on v_mfma_f32_16x16x32_f16.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o h3_mfma16_lab tools/h3_mfma16_lab.hip;
// LAB_R3=1 runs the three headline shapes on random operands, LAB_ZERO=1 on zeros, LAB_ONLY=8|16 one variant.
// --- header of the round-2 lab follows ---
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



// variant G (round 3): variant F's tile, staging and LDS image, contracted with v_mfma_f32_16x16x32_f16: a wave's
// 32 x 128 tile = 2 row blocks x 8 column blocks of 16 x 16, ONE MFMA per block and 32-k chunk (k = 32).
// A/B fragment of lane l: row (l & 15) of the block, k-octet (l >> 4) = 16-byte chunk (l >> 4) of the hi half of the
// 128-byte row image, chunk 4 + (l >> 4) of the lo half.  Same LDS bytes per MFMA-FLOP as variant F.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void gemm_h3g(const _Float16 *Ah, const _Float16 *Al, const _Float16 *Wh,
                                                const _Float16 *Wl, float *C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PA = 256 * 128, PB = 128 * 128, ST = PA + PB;      // 48 KB per buffer
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int tm, tn;
    tile_coords(M / 256, N / 128, tm, tn);
    const int row0 = tm * 256, col0 = tn * 128;
    f32x4 acc0[2][8], acc1[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
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
    const int fr = lane & 15, fq = lane >> 4, fsw = (fr >> 1) & 7;
    const int ph = (fq ^ fsw) * 16, pl = ((4 + fq) ^ fsw) * 16;
    h8 ah[2][2], al[2][2], bh[2][4], bl[2][4];       // A: [set][row block]; B: [half][column block of the half]
    auto lfragA = [&](int buf, auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
        const char *base = smem + buf * ST;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ra = (w * 32 + i * 16 + fr) * 128;
            ah[S][i] = *reinterpret_cast<const h8 *>(base + ra + ph);
            al[S][i] = *reinterpret_cast<const h8 *>(base + ra + pl);
        }
    };
    auto lfragB = [&](int buf, auto halfc) __attribute__((always_inline)) {
        constexpr int Hf = decltype(halfc)::value;
        const char *base = smem + buf * ST + PA;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rb = ((Hf * 4 + j) * 16 + fr) * 128;
            bh[Hf][j] = *reinterpret_cast<const h8 *>(base + rb + ph);
            bl[Hf][j] = *reinterpret_cast<const h8 *>(base + rb + pl);
        }
    };
    auto mma = [&](auto setc, auto halfc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value, Hf = decltype(halfc)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc0[i][Hf * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[S][i], bh[Hf][j], acc0[i][Hf * 4 + j], 0, 0, 0);
                acc1[i][Hf * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[S][i], bl[Hf][j], acc1[i][Hf * 4 + j], 0, 0, 0);
                acc1[i][Hf * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[S][i], bh[Hf][j], acc1[i][Hf * 4 + j], 0, 0, 0);
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
    lfragA(0, I0{});
    lfragB(0, I0{});
    int cur = 0;
    auto body = [&](int c, auto setc, auto nsetc) __attribute__((always_inline)) {
        const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;
        if (c + 2 < n) stage(nn);
        lfragB(cur, I1{});
        mma(setc, I0{});
        if (c + 1 < n) {
            wait_for(c + 2 < n ? 1 : 0);
            __syncthreads();
            lfragA(nxt, nsetc);
            lfragB(nxt, I0{});
        }
        mma(setc, I1{});
        cur = nxt;
    };
    for (int c = 0; c < n; c += 2) {
        body(c, I0{}, I1{});
        if (c + 1 < n) body(c + 1, I1{}, I0{});
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + w * 32 + i * 16 + (lane >> 4) * 4 + r;
                const int col = col0 + j * 16 + (lane & 15);
                C[(long long)row * N + col] = acc0[i][j][r] + acc1[i][j][r] * (1.f / 2048.f);
            }
}

// variant K (round 3): variant G with SQUARE wave tiles - the eight waves as 4 (M) x 2 (N) of 64 x 64 = 4 row blocks x
// 4 column blocks of 16 x 16: 16 fragment reads per 48 MFMAs instead of 20 (fewer LDS read bytes per MFMA).
// A/B fragment of lane l: row (l & 15) of the block, k-octet (l >> 4) = 16-byte chunk (l >> 4) of the hi half of the
// 128-byte row image, chunk 4 + (l >> 4) of the lo half.  Same LDS bytes per MFMA-FLOP as variant F.
__global__ __launch_bounds__(512) void gemm_h3k(const _Float16 *Ah, const _Float16 *Al, const _Float16 *Wh,
                                                const _Float16 *Wl, float *C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PA = 256 * 128, PB = 128 * 128, ST = PA + PB;      // 48 KB per buffer
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int tm, tn;
    tile_coords(M / 256, N / 128, tm, tn);
    const int row0 = tm * 256, col0 = tn * 128;
    f32x4 acc0[4][4], acc1[4][4];
    const int wm = w >> 1, wn = w & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
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
    const int fr = lane & 15, fq = lane >> 4, fsw = (fr >> 1) & 7;
    const int ph = (fq ^ fsw) * 16, pl = ((4 + fq) ^ fsw) * 16;
    h8 ah[2][4], al[2][4], bh[2][2], bl[2][2];       // A: [set][row block]; B: [half][column block of the half]
    auto lfragA = [&](int buf, auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
        const char *base = smem + buf * ST;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ra = (wm * 64 + i * 16 + fr) * 128;
            ah[S][i] = *reinterpret_cast<const h8 *>(base + ra + ph);
            al[S][i] = *reinterpret_cast<const h8 *>(base + ra + pl);
        }
    };
    auto lfragB = [&](int buf, auto halfc) __attribute__((always_inline)) {
        constexpr int Hf = decltype(halfc)::value;
        const char *base = smem + buf * ST + PA;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rb = (wn * 64 + (Hf * 2 + j) * 16 + fr) * 128;
            bh[Hf][j] = *reinterpret_cast<const h8 *>(base + rb + ph);
            bl[Hf][j] = *reinterpret_cast<const h8 *>(base + rb + pl);
        }
    };
    auto mma = [&](auto setc, auto halfc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value, Hf = decltype(halfc)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc0[i][Hf * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[S][i], bh[Hf][j], acc0[i][Hf * 2 + j], 0, 0, 0);
                acc1[i][Hf * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[S][i], bl[Hf][j], acc1[i][Hf * 2 + j], 0, 0, 0);
                acc1[i][Hf * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[S][i], bh[Hf][j], acc1[i][Hf * 2 + j], 0, 0, 0);
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
    lfragA(0, I0{});
    lfragB(0, I0{});
    int cur = 0;
    auto body = [&](int c, auto setc, auto nsetc) __attribute__((always_inline)) {
        const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;
        if (c + 2 < n) stage(nn);
        lfragB(cur, I1{});
        mma(setc, I0{});
        if (c + 1 < n) {
            wait_for(c + 2 < n ? 1 : 0);
            __syncthreads();
            lfragA(nxt, nsetc);
            lfragB(nxt, I0{});
        }
        mma(setc, I1{});
        cur = nxt;
    };
    for (int c = 0; c < n; c += 2) {
        body(c, I0{}, I1{});
        if (c + 1 < n) body(c + 1, I1{}, I0{});
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                const int col = col0 + wn * 64 + j * 16 + (lane & 15);
                C[(long long)row * N + col] = acc0[i][j][r] + acc1[i][j][r] * (1.f / 2048.f);
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
    CK(hipFuncSetAttribute((const void *)gemm_h3xf, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
    CK(hipFuncSetAttribute((const void *)gemm_h3g, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
    CK(hipFuncSetAttribute((const void *)gemm_h3k, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152));
    const int only = getenv("LAB_ONLY") ? atoi(getenv("LAB_ONLY")) : 0;
    for (int nb = 8; nb <= 24; nb += 8) {
        if (only && nb != only) continue;
        auto launch = [&]() {
            if (nb == 8) gemm_h3xf<<<(M / 256) * (N / 128), 512, 3 * 49152>>>(Ah, Al, Wh, Wl, dC, M, N, K);
            else if (nb == 16) gemm_h3g<<<(M / 256) * (N / 128), 512, 3 * 49152>>>(Ah, Al, Wh, Wl, dC, M, N, K);
            else gemm_h3k<<<(M / 256) * (N / 128), 512, 3 * 49152>>>(Ah, Al, Wh, Wl, dC, M, N, K);
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
        {   // every variant against the first one's output (same products, other summation order)
            static std::vector<float> first;
            std::vector<float> now((size_t)M * N);
            CK(hipMemcpy(now.data(), dC, now.size() * 4, hipMemcpyDeviceToHost));
            if (nb == 8 || first.size() != now.size()) first = now;
            double md = 0;
            for (size_t i = 0; i < now.size(); ++i) md = std::max(md, (double)std::fabs(now[i] - first[i]));
            printf("    max |C - C(v8)| = %.3e\n", md);
        }
        printf("M=%d N=%d K=%d  v%d (8: 256x128 8 waves, 32x32x16 MFMA = production; 16: the same on 16x16x32 MFMA; 24: 16x16x32 with square 64x64 wave tiles): %.1f us  %.1f TF(fp32-eq)  frac-of-833 %.3f\n", M, N, K, nb, us,
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
    // (round 3) headline shapes on RANDOM operands: classifier, lang-LSTM, prologue att_embed
    if (getenv("LAB_R3")) {
        run_shape(4096, 9984, 512, 0.05f, 0);
        run_shape(4096, 2048, 1536, 0.05f, 0);
        run_shape(4096, 2048, 1024, 0.05f, 0);
        run_shape(147456, 512, 2048, 0.05f, 0);
        return 0;
    }
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
