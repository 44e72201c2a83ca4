// Additive-attention scan and gate fusion (captioner.py:23-35, 50-62, 96-118) for gfx950.
//
// One 256-thread workgroup (4 wavefronts) per (batch row, scan problem).  The scan is the
// HBM-bound part of the decode step: per row it streams P[b] (R x A projected features) and
// V[b] (R x D features) exactly once - 2*R*512*4 B = 147,456 B for the 36-region content
// scan - with 16-byte-per-lane coalesced loads (a wavefront reads one contiguous 1 KiB of a
// region row per instruction), never writing the [B,R,A] tanh temporaries that the
// reference's six separate torch ops re-read and re-write.
//   phase 1  wave w scores regions w, w+4, ...:  e_r = w . tanh(P[b,r,:] + q[b,:] (+ q2[b,:]))
//            (three regions per wave-iteration; DPP/swizzle butterfly reductions)
//   phase 2  softmax over the R scores in LDS
//   phase 3  out[b,:] = sum_r alpha_r V[b,r,:]  (threads own a float4 of D; region groups are
//            combined through LDS in a fixed order => deterministic)
// Measured at B=4096: 5.0-5.2 TB/s of algorithmic bytes, 5.6 TB/s of actual DRAM traffic (PMC: 1.05x
// over-fetch + the 2 KB/row outputs) against 6.1-6.3 TB/s for a bare read-only sweep of the same
// 788 MB (tools/hbm_read_sweep.hip).  A register-resident variant (all 147 KB of a row in flight at
// once, 2 workgroups per CU) measured the same 5.0 TB/s, so the simpler streaming form stays.
#include "common.h"

struct DevScan {
    const float *P, *V, *q, *q2, *w, *w_bias;
    int R, A, D;
    float *out, *alpha_out;
    long long alpha_ld;
    _Float16 *out_hi, *out_lo;
    const int64_t *ids;        // gather mode: region r of row b = row ids[b*ids_ld + r] of the P / V tables
    long long ids_ld;
    int rid_off;               // float offset of the row-id staging area in dynamic LDS
    int rows;                  // rows of THIS problem (the grid spans the longest problem of the launch)
};
struct DevScanLaunch {
    DevScan p[2];
    int nt;                    // the per-caption P / V rows are streamed with non-temporal loads (host decides)
    const int *gate;           // optional: *gate == 0 -> the launch returns at once (isc_set_stream_gate)
};
// A launch whose per-caption rows do not fit the Infinity Cache (256 MB) next to the step's other traffic re-reads
// them from HBM every step anyway; loading them non-temporally (no allocation on the way) measured 149 -> 128 us at
// B = 4096 (788 MB per launch: 6.2 TB/s, what a bare read sweep reaches), and the step's GEMMs keep their weights in
// the cache: whole roll-outs B = 1536 6.57 -> 6.03 ms, B = 2048 7.65 -> 6.97 ms, B = 4096 12.44 -> 11.96 ms.  With only
// the first 1024 / 1536 / 2048 of 4096 captions on the default policy (hoping to keep those resident) the scan took
// 140 / 148 / 153 us.  Smaller launches keep the default - their rows DO stay resident from one step to the next
// (B = 512: 20 us default vs 24 us non-temporal; B = 1024, 151 MB: 35.4 vs 35.9 us, roll-out 4.84 vs 4.75 ms).
#define ISC_SCAN_NT_BYTES (128LL << 20)
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float4 *p, bool nt) {
    if (nt) {
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

template <int NA, bool NT>  // float4 per lane along A: A <= 256*NA; NT: per-caption rows by non-temporal loads
__global__ __launch_bounds__(256) void attn_scan_kernel(const DevScanLaunch L) {
    ISC_GATE_RETURN(L);
    rows_kernarg_warm<ROWS_KERNARG_LINES(DevScanLaunch)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevScan &S = L.p[blockIdx.y];
    const int b = blockIdx.x;
    if (b >= S.rows) return;               // (workgroup-uniform: a shorter problem of a two-problem launch)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = S.R, A = S.A, D = S.D;
    float *sc = smem;                      // [R] scores -> alphas
    float *part = smem + ((R + 3) & ~3);   // [ngrp][D] partial weighted sums
    // gather mode (sentiment words in eval mode: features are rows of two vocabulary-sized tables that stay
    // cache-resident, instead of [B,R,.] copies streamed from HBM every step): row ids of this caption in LDS
    int *rid = reinterpret_cast<int *>(smem + S.rid_off);
    const bool gather = S.ids != nullptr;
    const bool nt = NT && !gather;
    if (gather) {
        for (int r = tid; r < R; r += 256) rid[r] = (int)S.ids[(long long)b * S.ids_ld + r];
        __syncthreads();
    }

    // ---- phase 1: scores
    const int na4 = A >> 2;
    float4 qv[NA], wv[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int a4 = lane + 64 * i;
        if (a4 < na4) {
            qv[i] = reinterpret_cast<const float4 *>(S.q + (long long)b * A)[a4];
            if (S.q2) {
                const float4 t = reinterpret_cast<const float4 *>(S.q2 + (long long)b * A)[a4];
                qv[i].x += t.x; qv[i].y += t.y; qv[i].z += t.z; qv[i].w += t.w;
            }
            wv[i] = reinterpret_cast<const float4 *>(S.w)[a4];
        } else {
            qv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            wv[i] = qv[i];
        }
    }
    const float4 *Pb = reinterpret_cast<const float4 *>(gather ? S.P : S.P + (long long)b * R * A);
    const float w_bias = S.w_bias ? S.w_bias[0] : 0.f;
    // three regions per wave-iteration: their 3*NA 16-byte loads are issued back to back (6 KB in flight
    // per wave instead of 2) and the three butterfly reductions interleave
    for (int r0 = wave; r0 < R; r0 += 12) {
        float4 p[3][NA];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int r = r0 + 4 * u;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int a4 = lane + 64 * i;
                const long long pr = (r < R && gather) ? rid[r] : r;
                p[u][i] = (r < R && a4 < na4) ? ld4(&Pb[pr * na4 + a4], nt) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float acc[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            acc[u] = 0.f;
#pragma unroll
            for (int i = 0; i < NA; ++i) {            // lanes past A carry w = 0
                acc[u] += wv[i].x * isc_tanh(p[u][i].x + qv[i].x);
                acc[u] += wv[i].y * isc_tanh(p[u][i].y + qv[i].y);
                acc[u] += wv[i].z * isc_tanh(p[u][i].z + qv[i].z);
                acc[u] += wv[i].w * isc_tanh(p[u][i].w + qv[i].w);
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) acc[u] = half_sum(acc[u]);
#pragma unroll
        for (int u = 0; u < 3; ++u) acc[u] += __shfl_xor(acc[u], 32, 64);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (r0 + 4 * u < R) sc[r0 + 4 * u] = acc[u] + w_bias;
        }
    }
    __syncthreads();

    // ---- phase 2: softmax (every thread redundantly folds the R scores: LDS broadcasts)
    float mx = -INFINITY;
    for (int r = 0; r < R; ++r) mx = fmaxf(mx, sc[r]);
    float z = 0.f;
    for (int r = 0; r < R; ++r) z += __expf(sc[r] - mx);
    const float inv = 1.0f / z;
    __syncthreads();
    for (int r = tid; r < R; r += 256) {
        const float a = __expf(sc[r] - mx) * inv;
        sc[r] = a;
        if (S.alpha_out) S.alpha_out[(long long)b * S.alpha_ld + r] = a;
    }
    __syncthreads();

    // ---- phase 3: weighted sum
    const int nd4 = D >> 2;
    const float4 *Vb = reinterpret_cast<const float4 *>(gather ? S.V : S.V + (long long)b * R * D);
    if (nd4 <= 256) {
        const int ngrp = 256 / nd4;             // region groups working in parallel
        const int d4 = tid % nd4, grp = tid / nd4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grp < ngrp) {
            // six regions' loads in flight per thread (24 KB per workgroup, like phase 1): one at a time the
            // weighted sum was a chain of HBM round trips.  Rows past R re-read row R-1 with weight 0.
            for (int r0 = grp; r0 < R; r0 += 6 * ngrp) {
                float4 v[6];
                float a[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int r = r0 + u * ngrp;
                    const int rc = r < R ? r : R - 1;
                    v[u] = ld4(&Vb[(long long)(gather ? rid[rc] : rc) * nd4 + d4], nt);
                    a[u] = r < R ? sc[rc] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 6; ++u) {       // ascending region order, as before
                    o.x += a[u] * v[u].x; o.y += a[u] * v[u].y; o.z += a[u] * v[u].z; o.w += a[u] * v[u].w;
                }
            }
            reinterpret_cast<float4 *>(part)[grp * nd4 + d4] = o;
        }
        __syncthreads();
        if (tid < nd4) {
            float4 s = reinterpret_cast<float4 *>(part)[tid];
            for (int g = 1; g < ngrp; ++g) {
                const float4 v = reinterpret_cast<float4 *>(part)[g * nd4 + tid];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            reinterpret_cast<float4 *>(S.out + (long long)b * D)[tid] = s;
            if (S.out_hi) store_planes4(S.out_hi, S.out_lo, b, 4 * tid, D, s);
        }
    } else {
        for (int d4 = tid; d4 < nd4; d4 += 256) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int r = 0; r < R; ++r) {
                const float a = sc[r];
                const float4 v = Vb[(long long)(gather ? rid[r] : r) * nd4 + d4];
                o.x += a * v.x; o.y += a * v.y; o.z += a * v.z; o.w += a * v.w;
            }
            reinterpret_cast<float4 *>(S.out + (long long)b * D)[d4] = o;
            if (S.out_hi) store_planes4(S.out_hi, S.out_lo, b, 4 * d4, D, o);
        }
    }
}

extern "C" int isc_attn_scan_fwd(const isc_scan_problem *pr, int n_prob, int B, void *stream) {
    if (!pr) return ISC_E_NULL;
    if (n_prob < 1 || n_prob > 2 || B <= 0) return ISC_E_SHAPE;
    DevScanLaunch L = {};
    L.gate = isc_stream_gate_(stream);
    int maxA = 0, max_rows = 0;
    size_t lds = 0;
    for (int i = 0; i < n_prob; ++i) {
        const isc_scan_problem &q = pr[i];
        if (!q.P || !q.V || !q.q || !q.w || !q.out) return ISC_E_NULL;
        if (q.rows < 0) return ISC_E_SHAPE;
        L.p[i].rows = q.rows > 0 ? q.rows : B;
        if (L.p[i].rows > max_rows) max_rows = L.p[i].rows;
        if (q.R <= 0 || q.A <= 0 || q.D <= 0 || (q.A & 3) || (q.D & 3) || q.A > 1024) return ISC_E_SHAPE;
        if (!isc_aligned16(q.P) || !isc_aligned16(q.V) || !isc_aligned16(q.q) || !isc_aligned16(q.w) ||
            !isc_aligned16(q.out) || (q.q2 && !isc_aligned16(q.q2)))
            return ISC_E_ALIGN;
        DevScan &d = L.p[i];
        d.P = q.P; d.V = q.V; d.q = q.q; d.q2 = q.q2; d.w = q.w; d.w_bias = q.w_bias;
        d.R = q.R; d.A = q.A; d.D = q.D; d.out = q.out; d.alpha_out = q.alpha_out; d.alpha_ld = q.alpha_ld;
        if ((q.out_hi == nullptr) != (q.out_lo == nullptr)) return ISC_E_NULL;
        d.out_hi = static_cast<_Float16 *>(q.out_hi); d.out_lo = static_cast<_Float16 *>(q.out_lo);
        if (q.A > maxA) maxA = q.A;
        const int nd4 = q.D / 4;
        const size_t part = nd4 <= 256 ? (size_t)(256 / nd4) * q.D : 0;
        size_t need = (((size_t)q.R + 3) & ~(size_t)3) + part;
        d.ids = q.row_ids; d.ids_ld = q.row_ids_ld; d.rid_off = (int)need;
        if (q.row_ids) need += ((size_t)q.R + 3) & ~(size_t)3;
        if (need > lds) lds = need;
    }
    lds *= sizeof(float);
    if (lds > 60000) return ISC_E_SHAPE;
    long long streamed = 0;                                // bytes of per-caption rows (gathered tables are shared)
    for (int i = 0; i < n_prob; ++i)
        if (!pr[i].row_ids) streamed += (long long)L.p[i].rows * pr[i].R * ((long long)pr[i].A + pr[i].D) * 4;
    L.nt = streamed > ISC_SCAN_NT_BYTES;
    dim3 grid(max_rows, n_prob), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (L.nt) {
        if (maxA <= 256) hipLaunchKernelGGL((attn_scan_kernel<1, true>), grid, block, lds, st, L);
        else if (maxA <= 512) hipLaunchKernelGGL((attn_scan_kernel<2, true>), grid, block, lds, st, L);
        else hipLaunchKernelGGL((attn_scan_kernel<4, true>), grid, block, lds, st, L);
    } else {
        if (maxA <= 256) hipLaunchKernelGGL((attn_scan_kernel<1, false>), grid, block, lds, st, L);
        else if (maxA <= 512) hipLaunchKernelGGL((attn_scan_kernel<2, false>), grid, block, lds, st, L);
        else hipLaunchKernelGGL((attn_scan_kernel<4, false>), grid, block, lds, st, L);
    }
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ gated scan (round 3; few rows, inference)
// Content scan + sentiment scan + gate sum + gate mix of one decode step (captioner.py:96-118) as ONE launch: a workgroup
// per row runs both scans one after the other and keeps their outputs v and s; next to each weighted feature sum it
// accumulates, with the same attention weights, the rows of G_i = V_i W_i^T - the scan's features ALREADY carried through
// the gate's projection (cont2att / senti2att, no bias), prepared once per call (content) or per weight version
// (sentiment: a vocabulary-sized table like V and P):
//     z = zh + b_c + b_s + sum_r alpha_r G_c[b,r,:] + sum_m alpha'_m G_s[ids[b,m],:]  ==  zh + cont2att(v) + senti2att(s)
//     beta = sigmoid(w_g . tanh(z) + b_g),   f = beta v + (1 - beta) s
// so the step's gate GEMM [rows x A x (E + W)] and its gate-mix kernel disappear: three launches of the eight of a decode
// step become one.  The price is R x A more floats streamed per row and step (+50 % of the content scan's bytes), which is
// why the host takes this path only for few rows, where the scan is latency-bound and its rows stay cache-resident.
// Same fixed summation orders as attn_scan_kernel (region groups combined in ascending order): deterministic.
struct DevScanGate {
    DevScan p[2];
    const float *G[2];
    const float *zh, *b_c, *b_s, *w_g, *b_g;
    float *f, *beta;
    long long beta_ld;
    _Float16 *f_hi, *f_lo;
    int lds_half;              // floats of LDS per scan (the two scans of a row run side by side)
    const int *gate;           // optional: *gate == 0 -> the launch returns at once (isc_set_stream_gate)
};

// 512 threads per row: threads 0..255 run the content scan, threads 256..511 the sentiment scan AT THE SAME TIME (same
// code, own LDS area, the same sequence of workgroup barriers), so a row costs the longer of the two dependent chains,
// not their sum; both keep six regions' loads in flight per thread in the weighted sums (two tensors each).
template <int NA>
__global__ __launch_bounds__(512) void attn_scan_gate_kernel(const DevScanGate L) {
    ISC_GATE_RETURN(L);
    rows_kernarg_warm<ROWS_KERNARG_LINES(DevScanGate)>();
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int b = blockIdx.x;
    const int half = threadIdx.x >> 8;                   // 0: content scan, 1: sentiment scan (wave-uniform)
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const DevScan &S = L.p[half];
    const int A = S.A, D = S.D;                          // equal for both scans; A == D (checked by the host)
    const int na4 = A >> 2, nd4 = D >> 2;
    const int ngrp = 256 / nd4;                          // region groups of the weighted sums (nd4 <= 256)
    const int d4 = tid % nd4, grp = tid / nd4;
    const int R = S.R;
    float *smem = smem_all + half * L.lds_half;          // this scan's area
    float *sc = smem;                                    // [R] scores -> alphas
    float *part = smem + ((R + 3) & ~3);                 // [2][ngrp][D] partial weighted sums (V, then G)
    int *rid = reinterpret_cast<int *>(smem + S.rid_off);
    const bool gather = S.ids != nullptr;
    if (gather) {
        for (int r = tid; r < R; r += 256) rid[r] = (int)S.ids[(long long)b * S.ids_ld + r];
    }
    __syncthreads();
    // ---- scores: wave w of the half scores regions w, w+4, ...; three regions per wave-iteration
    float4 qv[NA], wv[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int a4 = lane + 64 * i;
        if (a4 < na4) {
            qv[i] = reinterpret_cast<const float4 *>(S.q + (long long)b * A)[a4];
            if (S.q2) {
                const float4 t = reinterpret_cast<const float4 *>(S.q2 + (long long)b * A)[a4];
                qv[i].x += t.x; qv[i].y += t.y; qv[i].z += t.z; qv[i].w += t.w;
            }
            wv[i] = reinterpret_cast<const float4 *>(S.w)[a4];
        } else {
            qv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            wv[i] = qv[i];
        }
    }
    const float4 *Pb = reinterpret_cast<const float4 *>(gather ? S.P : S.P + (long long)b * R * A);
    const float w_bias = S.w_bias ? S.w_bias[0] : 0.f;
    for (int r0 = wave; r0 < R; r0 += 12) {
        float4 pp[3][NA];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int r = r0 + 4 * u;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int a4 = lane + 64 * i;
                const long long pr = (r < R && gather) ? rid[r] : r;
                pp[u][i] = (r < R && a4 < na4) ? Pb[pr * na4 + a4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float acc[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            acc[u] = 0.f;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                acc[u] += wv[i].x * isc_tanh(pp[u][i].x + qv[i].x);
                acc[u] += wv[i].y * isc_tanh(pp[u][i].y + qv[i].y);
                acc[u] += wv[i].z * isc_tanh(pp[u][i].z + qv[i].z);
                acc[u] += wv[i].w * isc_tanh(pp[u][i].w + qv[i].w);
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) acc[u] = half_sum(acc[u]);
#pragma unroll
        for (int u = 0; u < 3; ++u) acc[u] += __shfl_xor(acc[u], 32, 64);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (r0 + 4 * u < R) sc[r0 + 4 * u] = acc[u] + w_bias;
        }
    }
    __syncthreads();
    // ---- softmax
    float mx = -INFINITY;
    for (int r = 0; r < R; ++r) mx = fmaxf(mx, sc[r]);
    float zsum = 0.f;
    for (int r = 0; r < R; ++r) zsum += __expf(sc[r] - mx);
    const float inv = 1.0f / zsum;
    __syncthreads();
    for (int r = tid; r < R; r += 256) {
        const float al = __expf(sc[r] - mx) * inv;
        sc[r] = al;
        if (S.alpha_out) S.alpha_out[(long long)b * S.alpha_ld + r] = al;
    }
    __syncthreads();
    // ---- weighted sums of V and of G (same weights, same ascending order): six regions in flight per thread
    const float4 *Vb = reinterpret_cast<const float4 *>(gather ? S.V : S.V + (long long)b * R * D);
    const float4 *Gb = reinterpret_cast<const float4 *>(gather ? L.G[half] : L.G[half] + (long long)b * R * A);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f), g = o;
    if (grp < ngrp) {
        for (int r0 = grp; r0 < R; r0 += 6 * ngrp) {
            float4 v[6], gg[6];
            float al[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int r = r0 + u * ngrp;
                const int rc = r < R ? r : R - 1;
                const long long row = gather ? rid[rc] : rc;
                v[u] = Vb[row * nd4 + d4];
                gg[u] = Gb[row * nd4 + d4];
                al[u] = r < R ? sc[rc] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                o.x += al[u] * v[u].x; o.y += al[u] * v[u].y; o.z += al[u] * v[u].z; o.w += al[u] * v[u].w;
                g.x += al[u] * gg[u].x; g.y += al[u] * gg[u].y; g.z += al[u] * gg[u].z; g.w += al[u] * gg[u].w;
            }
        }
        reinterpret_cast<float4 *>(part)[grp * nd4 + d4] = o;
        reinterpret_cast<float4 *>(part)[(ngrp + grp) * nd4 + d4] = g;
    }
    __syncthreads();
    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f), sg = sv;
    if (tid < nd4) {
        sv = reinterpret_cast<float4 *>(part)[tid];
        sg = reinterpret_cast<float4 *>(part)[ngrp * nd4 + tid];
        for (int k = 1; k < ngrp; ++k) {
            const float4 x = reinterpret_cast<float4 *>(part)[k * nd4 + tid];
            const float4 y = reinterpret_cast<float4 *>(part)[(ngrp + k) * nd4 + tid];
            sv.x += x.x; sv.y += x.y; sv.z += x.z; sv.w += x.w;
            sg.x += y.x; sg.y += y.y; sg.z += y.z; sg.w += y.w;
        }
        if (S.out) reinterpret_cast<float4 *>(S.out + (long long)b * D)[tid] = sv;
        if (S.out && S.out_hi) store_planes4(S.out_hi, S.out_lo, b, 4 * tid, D, sv);
    }
    // ---- gate: the sentiment half hands (s, senti2att(s)) to the content half through LDS
    __syncthreads();                                     // both halves are done with their `part` areas
    float4 *xs = reinterpret_cast<float4 *>(smem_all);   // [2][nd4]
    float *red = smem_all + 8 * nd4;                     // [4] wave partials of w_g . tanh(z)
    if (half == 1 && tid < nd4) { xs[tid] = sv; xs[nd4 + tid] = sg; }
    __syncthreads();
    float dot = 0.f;
    float4 sw = sv;
    if (half == 0 && tid < nd4) {
        sw = xs[tid];
        const float4 gs = xs[nd4 + tid];
        const float4 zh = reinterpret_cast<const float4 *>(L.zh + (long long)b * A)[tid];
        const float4 bc = reinterpret_cast<const float4 *>(L.b_c)[tid], bs = reinterpret_cast<const float4 *>(L.b_s)[tid];
        const float4 wg = reinterpret_cast<const float4 *>(L.w_g)[tid];
        const float zx = (zh.x + (sg.x + bc.x)) + (gs.x + bs.x);
        const float zy = (zh.y + (sg.y + bc.y)) + (gs.y + bs.y);
        const float zz = (zh.z + (sg.z + bc.z)) + (gs.z + bs.z);
        const float zw = (zh.w + (sg.w + bc.w)) + (gs.w + bs.w);
        dot = wg.x * isc_tanh(zx) + wg.y * isc_tanh(zy) + wg.z * isc_tanh(zz) + wg.w * isc_tanh(zw);
    }
    dot = wave_sum(dot);
    if (half == 0 && lane == 0) red[wave] = dot;
    __syncthreads();
    if (half == 0) {
        const float u = ((red[0] + red[1]) + (red[2] + red[3])) + (L.b_g ? L.b_g[0] : 0.f);
        const float beta = isc_sigmoid(u);
        if (tid == 0 && L.beta) L.beta[(long long)b * L.beta_ld] = beta;
        if (tid < nd4) {
            float4 f;
            f.x = beta * sv.x + (1.0f - beta) * sw.x; f.y = beta * sv.y + (1.0f - beta) * sw.y;
            f.z = beta * sv.z + (1.0f - beta) * sw.z; f.w = beta * sv.w + (1.0f - beta) * sw.w;
            reinterpret_cast<float4 *>(L.f + (long long)b * D)[tid] = f;
            if (L.f_hi) store_planes4(L.f_hi, L.f_lo, b, 4 * tid, D, f);
        }
    }
}

extern "C" int isc_attn_scan_gate_fwd(const isc_scan_gate_args *a, int B, void *stream) {
    if (!a) return ISC_E_NULL;
    if (B <= 0) return ISC_E_SHAPE;
    if (!a->zh || !a->b_gc || !a->b_gs || !a->w_gate || !a->f || !a->G[0] || !a->G[1]) return ISC_E_NULL;
    if ((a->f_hi == nullptr) != (a->f_lo == nullptr)) return ISC_E_NULL;
    {   // few hundred rows, <= 36 regions / 12 words: one 1024-thread workgroup per row (rows.hip)
        int rc = ISC_OK;
        if (rows_scan_gate_try(a, B, (hipStream_t)stream, &rc)) return rc;
    }
    DevScanGate L = {};
    L.gate = isc_stream_gate_(stream);
    size_t lds = 0;
    const int A = a->scan[0].A, D = a->scan[0].D;
    if (A <= 0 || A != D || (A & 3) || A > 1024) return ISC_E_SHAPE;       // (a thread holds four columns: A / 4 <= 256)
    for (int i = 0; i < 2; ++i) {
        const isc_scan_problem &q = a->scan[i];
        if (!q.P || !q.V || !q.q || !q.w) return ISC_E_NULL;
        if (q.R <= 0 || q.A != A || q.D != D) return ISC_E_SHAPE;
        if (!isc_aligned16(q.P) || !isc_aligned16(q.V) || !isc_aligned16(q.q) || !isc_aligned16(q.w) ||
            !isc_aligned16(a->G[i]) || (q.q2 && !isc_aligned16(q.q2)) || (q.out && !isc_aligned16(q.out)))
            return ISC_E_ALIGN;
        if ((q.out_hi == nullptr) != (q.out_lo == nullptr)) return ISC_E_NULL;
        DevScan &d = L.p[i];
        d.P = q.P; d.V = q.V; d.q = q.q; d.q2 = q.q2; d.w = q.w; d.w_bias = q.w_bias;
        d.R = q.R; d.A = q.A; d.D = q.D; d.out = q.out; d.alpha_out = q.alpha_out; d.alpha_ld = q.alpha_ld;
        d.out_hi = static_cast<_Float16 *>(q.out_hi); d.out_lo = static_cast<_Float16 *>(q.out_lo);
        const int nd4 = q.D / 4;
        size_t need = (((size_t)q.R + 3) & ~(size_t)3) + 2 * (size_t)(256 / nd4) * q.D;
        d.ids = q.row_ids; d.ids_ld = q.row_ids_ld; d.rid_off = (int)need;
        if (q.row_ids) need += ((size_t)q.R + 3) & ~(size_t)3;
        if (need > lds) lds = need;
        L.G[i] = a->G[i];
    }
    if (!isc_aligned16(a->zh) || !isc_aligned16(a->b_gc) || !isc_aligned16(a->b_gs) || !isc_aligned16(a->w_gate) ||
        !isc_aligned16(a->f))
        return ISC_E_ALIGN;
    if (lds < (size_t)(2 * A + 16)) lds = 2 * A + 16;        // the hand-over area of the gate lives at the start
    lds = (lds + 3) & ~(size_t)3;
    L.lds_half = (int)lds;
    lds *= 2 * sizeof(float);
    if (lds > 60000) return ISC_E_SHAPE;
    L.zh = a->zh; L.b_c = a->b_gc; L.b_s = a->b_gs; L.w_g = a->w_gate; L.b_g = a->b_gate;
    L.f = a->f; L.beta = a->beta; L.beta_ld = a->beta_ld;
    L.f_hi = static_cast<_Float16 *>(a->f_hi); L.f_lo = static_cast<_Float16 *>(a->f_lo);
    dim3 grid(B), block(512);
    hipStream_t st = (hipStream_t)stream;
    if (A <= 256) hipLaunchKernelGGL((attn_scan_gate_kernel<1>), grid, block, lds, st, L);
    else if (A <= 512) hipLaunchKernelGGL((attn_scan_gate_kernel<2>), grid, block, lds, st, L);
    else hipLaunchKernelGGL((attn_scan_gate_kernel<4>), grid, block, lds, st, L);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

__device__ __forceinline__ bool isc_aligned16_dev(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// beta = sigmoid(w . tanh(z) + w_bias); out = beta*v + (1-beta)*s.  One wavefront per row.
__global__ __launch_bounds__(256) void gate_mix_kernel(const float *z, const float *w, const float *w_bias,
                                                       const float *v, const float *s, int B, int A,
                                                       int D, float *out, float *beta_out,
                                                       long long beta_ld, _Float16 *out_hi, _Float16 *out_lo) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float acc = 0.f;
    if ((A & 3) == 0 && isc_aligned16_dev(z) && isc_aligned16_dev(w)) {     // 16-byte loads: two per lane at A = 512
        const float4 *z4 = reinterpret_cast<const float4 *>(z + (long long)b * A);
        const float4 *w4 = reinterpret_cast<const float4 *>(w);
        for (int a = lane; a < (A >> 2); a += 64) {
            const float4 zz = z4[a], ww = w4[a];
            // same products, summed in the element order of the scalar loop within the lane's four
            acc += ww.x * isc_tanh(zz.x);
            acc += ww.y * isc_tanh(zz.y);
            acc += ww.z * isc_tanh(zz.z);
            acc += ww.w * isc_tanh(zz.w);
        }
    } else {
        for (int a = lane; a < A; a += 64) acc += w[a] * isc_tanh(z[(long long)b * A + a]);
    }
    acc = wave_sum(acc);
    const float beta = isc_sigmoid(acc + (w_bias ? w_bias[0] : 0.f));
    if (lane == 0 && beta_out) beta_out[(long long)b * beta_ld] = beta;
    if ((D & 3) == 0) {      // rows are 16-byte aligned (D % 4 == 0, device-allocated bases): four outputs per lane and pass
        const float4 *v4 = reinterpret_cast<const float4 *>(v + (long long)b * D);
        const float4 *s4 = reinterpret_cast<const float4 *>(s + (long long)b * D);
        float4 *o4 = reinterpret_cast<float4 *>(out + (long long)b * D);
        for (int d = lane; d < (D >> 2); d += 64) {
            const float4 a = v4[d], c = s4[d];
            float4 f;
            f.x = beta * a.x + (1.0f - beta) * c.x; f.y = beta * a.y + (1.0f - beta) * c.y;
            f.z = beta * a.z + (1.0f - beta) * c.z; f.w = beta * a.w + (1.0f - beta) * c.w;
            o4[d] = f;
            if (out_hi) store_planes4(out_hi, out_lo, b, 4 * d, D, f);
        }
        return;
    }
    for (int d = lane; d < D; d += 64) {
        const long long o = (long long)b * D + d;
        const float f = beta * v[o] + (1.0f - beta) * s[o];
        out[o] = f;
        if (out_hi) {
            const _Float16 fh = (_Float16)f;
            const long long po = (long long)b * 2 * D + (d >> 5) * 64 + (d & 31);
            out_hi[po] = fh;
            out_lo[po] = (_Float16)((f - (float)fh) * 2048.f);
        }
    }
}

extern "C" int isc_gate_mix_fwd(const float *z, const float *w, const float *w_bias, const float *v,
                                const float *s, int B, int A, int D, float *out, float *beta_out,
                                int64_t beta_ld, void *out_hi, void *out_lo, void *stream) {
    if (!z || !w || !v || !s || !out) return ISC_E_NULL;
    if ((out_hi == nullptr) != (out_lo == nullptr)) return ISC_E_NULL;
    if ((D & 3) == 0 && (!isc_aligned16(v) || !isc_aligned16(s) || !isc_aligned16(out))) return ISC_E_ALIGN;
    if (B <= 0 || A <= 0 || D <= 0) return ISC_E_SHAPE;
    hipLaunchKernelGGL(gate_mix_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, w,
                       w_bias, v, s, B, A, D, out, beta_out, (long long)beta_ld, static_cast<_Float16 *>(out_hi),
                       static_cast<_Float16 *>(out_lo));
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}
