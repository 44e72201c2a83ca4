// Shared host/device helpers for libinsenticap_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/insenticap_hip.h"

#define ISC_WAVE 64

static inline int isc_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

#define ISC_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

__device__ __forceinline__ float isc_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

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
