// Shared host/device helpers for libinsenticap_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/insenticap_hip.h"

#define ISC_WAVE 64

static inline int isc_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

// Stream gate (isc_set_stream_gate, step.hip): the device int the forward launches enqueued on `stream` look at first -
// 0 there and the launch returns at once, every thread, before any barrier.  nullptr: no gate (the normal case).
const int *isc_stream_gate_(void *stream);
// ... in a kernel whose descriptor L carries the pointer as L.gate (uniform: a scalar load and a scalar branch)
#define ISC_GATE_RETURN(L)                                          \
    do {                                                            \
        if ((L).gate != nullptr && *(L).gate == 0) return;          \
    } while (0)

#define ISC_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// Activations on the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each): 5-6 VALU
// instructions instead of libm's ~25 (sigmoid) / ~40 (tanhf).  The attention scan evaluates 24k tanh
// per caption and step - with tanhf the "HBM-bound" kernel was in fact VALU-limited.  Absolute error
// <= ~2e-7 (tanh near 0 is accurate to ~1.2e-7 absolute, not relative), far inside the 1e-4 log-prob
// bound; +-inf saturate correctly and NaN propagates.
__device__ __forceinline__ float isc_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float isc_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// ReLU that lets NaN through, as torch.relu does (v_max_f32 / fmaxf return the non-NaN operand: a NaN pre-activation would
// come out as a clean 0 and an operand outside the split-f16 domain would go unnoticed - see isc_status).
__device__ __forceinline__ float isc_relu(float x) { return x < 0.f ? 0.f : x; }

// Sticky numerics status: two 32-bit words in host memory the caller registered (isc_set_status_words), visible to the
// device; a kernel that meets a non-finite value stores 1 into its word - a plain store, raced benignly, executed only on
// the error path - and the host reads the words as ordinary memory once it has waited for the work (no device call).
// One pointer variable per translation unit (no relocatable device code): ISC_STATUS_DECL in each .hip that flags.
#define ISC_STATUS_WORD_STATS 0
#define ISC_STATUS_WORD_LINEAR 1
#define ISC_STATUS_DECL(NAME)                                                                          \
    __device__ unsigned int *g_isc_status_##NAME;                                                      \
    int isc_set_status_##NAME##_(unsigned int *p) {                                                    \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_isc_status_##NAME), &p, sizeof(p)) == hipSuccess ? 0 : 1; \
    }                                                                                                  \
    __device__ __forceinline__ void isc_flag_##NAME(int word) {                                        \
        unsigned int *sp = g_isc_status_##NAME;                                                        \
        if (sp) __hip_atomic_store(sp + word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);        \
    }

// Diagnostic build (tools/rows_stamp_lab.hip / tools/select_stamp_lab.hip define ROWS_STAMP 1 and include a source file):
// lane 0 of every wave stores the 100 MHz wall clock at a few points into a buffer of its own ([workgroup][16 waves][8
// slots]; slot 7: shader cycles between RSTAMP_CLK0 and RSTAMP_CLK1).  The default build compiles none of it.
#ifndef ROWS_STAMP
#define ROWS_STAMP 0
#endif
#if ROWS_STAMP
// (one pointer per translation unit, set through that unit's isc_*_set_stamp; a kernel stamps into region RSTAMP_KID of the
// buffer - 1024 workgroups x 16 waves x 8 slots each - so that the kernels of one decode step do not overwrite each other)
static __device__ long long *g_rows_stamp;
#define RSTAMP_REGION 131072
#define RSTAMP_K(KID, SLOT)                                                                                  \
    do {                                                                                                     \
        if ((threadIdx.x & 63) == 0 && g_rows_stamp && blockIdx.x < 1024)                                     \
            g_rows_stamp[(long long)(KID) * RSTAMP_REGION + ((long long)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (SLOT)] = wall_clock64(); \
    } while (0)
#define RSTAMP(SLOT) RSTAMP_K(RSTAMP_KID, SLOT)
#define RSTAMP2(W, SLOT)                                                                                     \
    do {                                                                                                     \
        if (threadIdx.x == 0 && g_rows_stamp) g_rows_stamp[(long long)RSTAMP_KID * RSTAMP_REGION + ((long long)blockIdx.x * 16 + (W)) * 8 + (SLOT)] = wall_clock64(); \
    } while (0)
#define RSTAMP_CLK0() const long long rstamp_c0 = clock64()
#define RSTAMP_CLK1()                                                                                        \
    do {                                                                                                     \
        if ((threadIdx.x & 63) == 0 && g_rows_stamp && blockIdx.x < 1024)                                     \
            g_rows_stamp[(long long)RSTAMP_KID * RSTAMP_REGION + ((long long)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + 7] = clock64() - rstamp_c0;   \
    } while (0)
#define RSTAMP_SETTER(NAME)                                                                                  \
    extern "C" int NAME(long long *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_rows_stamp), &p, sizeof(p)) == hipSuccess ? 0 : 1; }
#else
#define RSTAMP(SLOT) do {} while (0)
#define RSTAMP2(W, SLOT) do {} while (0)
#define RSTAMP_CLK0() do {} while (0)
#define RSTAMP_CLK1() do {} while (0)
#define RSTAMP_K(KID, SLOT) do {} while (0)
#endif

// rows.hip: isc_attn_scan_gate_fwd's launches of up to isc_set_rows_scan_max rows on the one-workgroup-per-row kernel.
// Returns 1 when it took the launch (*rc = its status), 0 when the shape is not its own.
int rows_scan_gate_try(const isc_scan_gate_args *a, int rows, hipStream_t st, int *rc);

// f16 planes of four consecutive outputs d .. d+3 of row `row` of a [rows, D] tensor (split-f16 GEMM operands:
// hi = f16(x), lo = f16((x - hi) * 2048); interleaved layout of gemm_f32.hip: per row and 32-wide block 32 hi then
// 32 lo values, lo pointer = hi pointer + 32)
__device__ __forceinline__ void store_planes4(_Float16 *hi, _Float16 *lo, long long row, int d, int D, const float4 &x) {
    const long long o = row * 2 * D + (d >> 5) * 64 + (d & 31);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const float v[4] = {x.x, x.y, x.z, x.w};
    h4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = (_Float16)v[e];
        b[e] = (_Float16)((v[e] - (float)a[e]) * 2048.f);
    }
    *reinterpret_cast<h4 *>(hi + o) = a;
    *reinterpret_cast<h4 *>(lo + o) = b;
}

// Kernel arguments are read by scalar loads where they are first used; every first touch of a 64-byte line of the
// kernarg segment is a scalar-cache miss (a round trip to L2 or beyond), and hipcc places those loads lazily, one
// dependent wait after the other - five lines cost five round trips in front of the first weight load.  This touches
// one dword of each of the first NL lines in ONE batch and waits once: every later argument read hits the scalar cache.
template <int NL>
__device__ __forceinline__ void rows_kernarg_warm() {
    static_assert(NL >= 1 && NL <= 28, "lines");
    const unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    // Every load targets the SAME scratch SGPR (its value is never used; scalar loads may return in any order, which is
    // all a write-after-write on a dead register can show) - one statement, so the register stays reserved until the wait.
    // NOTHING past the struct is touched: a kernel that uses no hidden argument has a kernarg segment of exactly
    // sizeof(struct) bytes, and the segment may end where the mapping ends (a read 64 ... 192 bytes past it faulted in
    // a B = 1024 training graph).  The statement has four-line groups; the slots past line NL - 1 repeat that line.
    int d;
#define KA_O(i) "n"(64 * ((i) < NL ? (i) : NL - 1))
#define KA_G(a, b, c, e) "s_load_dword %0, %1, %" #a "\n\ts_load_dword %0, %1, %" #b "\n\ts_load_dword %0, %1, %" #c \
                         "\n\ts_load_dword %0, %1, %" #e "\n\t"
#define KA_W "s_waitcnt lgkmcnt(0)"
    constexpr int G = (NL + 3) / 4;
    if constexpr (G == 1)
        asm volatile(KA_G(2, 3, 4, 5) KA_W : "=&s"(d) : "s"(kp), KA_O(0), KA_O(1), KA_O(2), KA_O(3) : "memory");
    else if constexpr (G == 2)
        asm volatile(KA_G(2, 3, 4, 5) KA_G(6, 7, 8, 9) KA_W : "=&s"(d)
                     : "s"(kp), KA_O(0), KA_O(1), KA_O(2), KA_O(3), KA_O(4), KA_O(5), KA_O(6), KA_O(7) : "memory");
    else if constexpr (G == 3)
        asm volatile(KA_G(2, 3, 4, 5) KA_G(6, 7, 8, 9) KA_G(10, 11, 12, 13) KA_W : "=&s"(d)
                     : "s"(kp), KA_O(0), KA_O(1), KA_O(2), KA_O(3), KA_O(4), KA_O(5), KA_O(6), KA_O(7), KA_O(8), KA_O(9),
                       KA_O(10), KA_O(11) : "memory");
    else
        asm volatile(KA_G(2, 3, 4, 5) KA_G(6, 7, 8, 9) KA_G(10, 11, 12, 13) KA_G(14, 15, 16, 17) KA_G(18, 19, 20, 21)
                     KA_G(22, 23, 24, 25) KA_G(26, 27, 28, 29) KA_W : "=&s"(d)
                     : "s"(kp), KA_O(0), KA_O(1), KA_O(2), KA_O(3), KA_O(4), KA_O(5), KA_O(6), KA_O(7), KA_O(8), KA_O(9),
                       KA_O(10), KA_O(11), KA_O(12), KA_O(13), KA_O(14), KA_O(15), KA_O(16), KA_O(17), KA_O(18), KA_O(19),
                       KA_O(20), KA_O(21), KA_O(22), KA_O(23), KA_O(24), KA_O(25), KA_O(26), KA_O(27) : "memory");
#undef KA_W
#undef KA_G
#undef KA_O
}
#define ROWS_KERNARG_LINES(T) ((int)((sizeof(T) + 63) / 64))

// DPP lane exchanges inside a 16-lane row (VALU rate, no LDS crossbar): xor 1, xor 2, mirror inside each
// 8-lane half, mirror inside the row.  Applied in this order with a commutative combine they leave every
// lane of a row holding the row's reduction.
template <int CTRL>
__device__ __forceinline__ float isc_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int isc_dpp(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
#define ISC_DPP_XOR1 0xB1         // quad_perm [1,0,3,2]
#define ISC_DPP_XOR2 0x4E         // quad_perm [2,3,0,1]
#define ISC_DPP_HALF_MIRROR 0x141 // row_half_mirror
#define ISC_DPP_MIRROR 0x140      // row_mirror
// lane l <-> l ^ 16 inside each 32-lane half (ds_swizzle bit mode: and 0x1f, or 0, xor 0x10)
__device__ __forceinline__ float isc_swz16(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
}
__device__ __forceinline__ int isc_swz16(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x401F); }
// all-reduce over each 32-lane half of the wavefront
__device__ __forceinline__ float half_sum(float v) {
    v += isc_dpp<ISC_DPP_XOR1>(v);
    v += isc_dpp<ISC_DPP_XOR2>(v);
    v += isc_dpp<ISC_DPP_HALF_MIRROR>(v);
    v += isc_dpp<ISC_DPP_MIRROR>(v);
    return v + isc_swz16(v);
}
// (value, index) arg-max over each 32-lane half; ties resolve to the smaller index
__device__ __forceinline__ void half_argmax(float &v, int &i) {
#define ISC_ARGMAX_STEP(OV, OI)                                        \
    {                                                                  \
        const float ov = (OV);                                         \
        const int oi = (OI);                                           \
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }         \
    }
    ISC_ARGMAX_STEP(isc_dpp<ISC_DPP_XOR1>(v), isc_dpp<ISC_DPP_XOR1>(i))
    ISC_ARGMAX_STEP(isc_dpp<ISC_DPP_XOR2>(v), isc_dpp<ISC_DPP_XOR2>(i))
    ISC_ARGMAX_STEP(isc_dpp<ISC_DPP_HALF_MIRROR>(v), isc_dpp<ISC_DPP_HALF_MIRROR>(i))
    ISC_ARGMAX_STEP(isc_dpp<ISC_DPP_MIRROR>(v), isc_dpp<ISC_DPP_MIRROR>(i))
    ISC_ARGMAX_STEP(isc_swz16(v), isc_swz16(i))
#undef ISC_ARGMAX_STEP
}

// A row's tile statistics folded by one wave (the body of pointwise.hip's fold_row_stats; rows.hip folds with it as well):
// gmax = the row's maximum, gidx = its vocabulary index (smallest index on ties), S = sum exp(x - gmax); all 64 lanes get
// the result.  Returns true when the row is not finite (the caller flags the numerics status).
__device__ __forceinline__ bool fold_row_stats_impl(const float *pmax, const float *psum, const int *pidx,
                                                    int n_tile, int lane, float &gmax, int &gidx, float &S) {
    // every tile statistic of the row is requested before the first exchange (two strided passes with a reduction
    // between them were two dependent memory round trips per decode step on the roll-out's critical path)
    float mx = -INFINITY;
    int ix = 0x7fffffff;
    for (int i = lane; i < n_tile; i += 64) {
        const float v = pmax[i];
        const int id = pidx ? pidx[i] : i;
        if (v > mx || (v == mx && id < ix)) { mx = v; ix = id; }
    }
    // lane exchanges inside 32-lane halves by DPP, one cross-half swap (same order: value, then the smaller index)
    half_argmax(mx, ix);
    {
        const float ov = __shfl_xor(mx, 32, 64);
        const int oi = __shfl_xor(ix, 32, 64);
        if (ov > mx || (ov == mx && oi < ix)) { mx = ov; ix = oi; }
    }
    float s = 0.f;
    for (int i = lane; i < n_tile; i += 64) s += psum[i] * expf(pmax[i] - mx);
    s = half_sum(s);
    S = s + __shfl_xor(s, 32, 64);
    gmax = mx;
    // a row whose maxima are all NaN never satisfied `v > mx`: its index is still the sentinel, and a consumer would
    // gather an embedding row 2^31 rows past the table.  Such a row decodes <PAD> (id 0) and is flagged.
    gidx = ix == 0x7fffffff ? 0 : ix;
    return !(fabsf(mx) <= 3.0e38f && S <= 3.0e38f);          // NaN fails both
}


// all-reduce over each 16-lane row of the wavefront (the lanes that share (lane >> 4): one output row of the 16x16 MFMA's
// C/D layout): four DPP steps, no LDS crossbar
__device__ __forceinline__ float row16_sum(float v) {
    v += isc_dpp<ISC_DPP_XOR1>(v);
    v += isc_dpp<ISC_DPP_XOR2>(v);
    v += isc_dpp<ISC_DPP_HALF_MIRROR>(v);
    return v + isc_dpp<ISC_DPP_MIRROR>(v);
}
// (value, index) arg-max over each 16-lane row; ties resolve to the smaller index
__device__ __forceinline__ void row16_argmax(float &v, int &i) {
#define ISC_ARGMAX_STEP(OV, OI)                                        \
    {                                                                  \
        const float ov = (OV);                                         \
        const int oi = (OI);                                           \
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }         \
    }
    ISC_ARGMAX_STEP(isc_dpp<ISC_DPP_XOR1>(v), isc_dpp<ISC_DPP_XOR1>(i))
    ISC_ARGMAX_STEP(isc_dpp<ISC_DPP_XOR2>(v), isc_dpp<ISC_DPP_XOR2>(i))
    ISC_ARGMAX_STEP(isc_dpp<ISC_DPP_HALF_MIRROR>(v), isc_dpp<ISC_DPP_HALF_MIRROR>(i))
    ISC_ARGMAX_STEP(isc_dpp<ISC_DPP_MIRROR>(v), isc_dpp<ISC_DPP_MIRROR>(i))
#undef ISC_ARGMAX_STEP
}

// 64-lane butterfly reductions (wavefront = 64 on CDNA4).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// (value, index) arg-max; ties resolve to the smaller index.
__device__ __forceinline__ void wave_argmax(float &v, int &i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(v, o, 64);
        int oi = __shfl_xor(i, o, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}
