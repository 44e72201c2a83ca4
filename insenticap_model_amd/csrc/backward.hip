// Backward (BPTT) kernels of the decode step - what torch autograd derives for the reference's
// stock ops (train_xe.py:189-192, decoder.py:161-167) - plus the fused clamp+Adam update
// (train_xe.py:19-23 / captioner.py:422-423).  All contractions run on the MFMA GEMM
// (isc_gemm_bwd); this file holds the HBM-bound pointwise / scan parts.
#include "common.h"

// ------------------------------------------------------------------ log-softmax backward
// dlogits[m,:] = dlogp[m,:] - exp(logp[m,:]) * sum_v dlogp[m,v];  columns V..ld_out-1 are zeroed
// (the padded row doubles as the k-minor A operand of dH = dlogits * W_cls).
__global__ __launch_bounds__(256) void logsoftmax_bwd_kernel(const float *dlogp, const float *logp,
                                                             long long ld_in, int V, float *dlogits,
                                                             long long ld_out, int M, int remap_T) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // optional row remap [B,T] -> [T,B]: the time-major copy feeds the per-step BPTT slices
    const int mo = remap_T > 0 ? (m % remap_T) * (M / remap_T) + m / remap_T : m;
    const float *g = dlogp + (long long)m * ld_in;
    const float *lp = logp + (long long)m * ld_in;
    float s = 0.f;
    for (int i = tid; i < V; i += 256) s += g[i];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    float *o = dlogits + (long long)mo * ld_out;
    for (int i = tid; i < ld_out; i += 256) o[i] = (i < V) ? g[i] - expf(lp[i]) * tot : 0.f;
}

extern "C" int isc_logsoftmax_bwd(const float *dlogp, const float *logp, int64_t ld_in, int M, int V,
                                  float *dlogits, int64_t ld_out, int remap_T, void *stream) {
    if (!dlogp || !logp || !dlogits) return ISC_E_NULL;
    if (M <= 0 || V <= 0 || ld_out < V || remap_T < 0 || (remap_T > 0 && M % remap_T)) return ISC_E_SHAPE;
    hipLaunchKernelGGL(logsoftmax_bwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, dlogp, logp,
                       (long long)ld_in, V, dlogits, (long long)ld_out, M, remap_T);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// Sparse hand-over of d log-prob (round 3).  The criteria of this path touch ONE column per (caption, step) row:
// XECriterion the target token (captioner.py:427-440), the REINFORCE term the drawn token (self_critical/utils.py:
// 169-177 applied to captioner.py:336's gather).  Their gradient w.r.t. the [B,T,V] log-probs is therefore `coef[m]` at
// column `ids[m]` of row m and zero elsewhere; as a dense tensor it cost a zero fill, a scatter and a full read (2.5 GB
// of traffic at B = 1024).  Here it arrives as up to ISC_SPARSE_MAX (ids, coef) pairs per row, next to an OPTIONAL
// dense part (a caller-defined loss on the log-probs), and
//   d logits[m, v] = scale * ( dense[m, v] + sum_j coef_j[m] [v == ids_j[m]] - exp(logp[m, v]) * tot[m] ),
//   tot[m] = sum_v dense[m, v] + sum_j coef_j[m]
// `scale` (device scalar, may be NULL = 1) is the power-of-two gradient scale of isc_grad_scale.
struct SparseDlogp {
    const int64_t *ids[ISC_SPARSE_MAX];
    const float *coef[ISC_SPARSE_MAX];
    int n;
};

__global__ __launch_bounds__(256) void logsoftmax_bwd_sparse_kernel(const float *dense, const float *logp,
                                                                    long long ld_in, int V, SparseDlogp sp,
                                                                    const float *scale, float *dlogits,
                                                                    long long ld_out, int M, int remap_T,
                                                                    int out_step_rows) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (out_step_rows: rows per step of the time-major output - M / T, or more when it holds a sibling unroll's rows too)
    const long long mo = remap_T > 0 ? (long long)(m % remap_T) * out_step_rows + m / remap_T : m;
    const float *lp = logp + (long long)m * ld_in;
    const float sc = scale ? scale[0] : 1.f;
    float tot = 0.f;
    if (dense) {
        const float *g = dense + (long long)m * ld_in;
        float s = 0.f;
        for (int i = tid; i < V; i += 256) s += g[i];
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        tot = (red[0] + red[1]) + (red[2] + red[3]);
    }
    long long id[ISC_SPARSE_MAX];
    float cf[ISC_SPARSE_MAX];
#pragma unroll
    for (int j = 0; j < ISC_SPARSE_MAX; ++j) {
        id[j] = -1; cf[j] = 0.f;
        if (j < sp.n) { id[j] = sp.ids[j][m]; cf[j] = sp.coef[j][m]; tot += cf[j]; }
    }
    float *o = dlogits + mo * ld_out;
    const float *g = dense ? dense + (long long)m * ld_in : nullptr;
    for (int i = tid; i < ld_out; i += 256) {
        float v = 0.f;
        if (i < V) {
            v = g ? g[i] : 0.f;
#pragma unroll
            for (int j = 0; j < ISC_SPARSE_MAX; ++j)
                if ((long long)i == id[j]) v += cf[j];
            v = (v - expf(lp[i]) * tot) * sc;
        }
        o[i] = v;
    }
}

extern "C" int isc_logsoftmax_bwd_sparse(const float *dlogp_dense, const float *logp, int64_t ld_in, int M, int V,
                                         const int64_t *const *ids_host, const float *const *coef_host, int n_sparse,
                                         const float *scale, float *dlogits, int64_t ld_out, int remap_T,
                                         int out_step_rows, void *stream) {
    if (!logp || !dlogits) return ISC_E_NULL;
    if (M <= 0 || V <= 0 || ld_out < V || remap_T < 0 || (remap_T > 0 && M % remap_T)) return ISC_E_SHAPE;
    if (remap_T > 0 && out_step_rows == 0) out_step_rows = M / remap_T;
    if (out_step_rows < 0 || (remap_T > 0 && out_step_rows < M / remap_T)) return ISC_E_SHAPE;
    if (n_sparse < 0 || n_sparse > ISC_SPARSE_MAX || (!dlogp_dense && n_sparse == 0)) return ISC_E_SHAPE;
    SparseDlogp sp = {};
    sp.n = n_sparse;
    for (int j = 0; j < n_sparse; ++j) {
        if (!ids_host || !coef_host || !ids_host[j] || !coef_host[j]) return ISC_E_NULL;
        sp.ids[j] = ids_host[j]; sp.coef[j] = coef_host[j];
    }
    hipLaunchKernelGGL(logsoftmax_bwd_sparse_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, dlogp_dense, logp,
                       (long long)ld_in, V, sp, scale, dlogits, (long long)ld_out, M, remap_T, out_step_rows);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// The same gradient from the RAW logits and their tile statistics (isc_gather_logp_raw's counterpart: the training
// iteration that never wrote the [B,T,V] log-probs): exp(logp) = exp((x - max) - log(sum)), the expression the log-probs
// would have been stored with - the same d logits bits.  Rows in [B,T] order for ids / coef; logits row (b,t) at
// b*ld_b + t*ld_t, statistics row t*step_rows + b, output row t*out_step_rows + b (time-major).
template <bool VEC4>
__global__ __launch_bounds__(256) void logsoftmax_bwd_raw_kernel(const float *raw, long long ld_b, long long ld_t, int B, int T,
                                                                 int V, const float *pmax, const float *psum, int n_tile,
                                                                 int step_rows, SparseDlogp sp, const float *scale,
                                                                 float *dlogits, long long ld_out, int out_step_rows) {
    __shared__ float sh[2];
    const int mi = blockIdx.x, tid = threadIdx.x;
    const int b = mi / T, t = mi - b * T;
    const long long ms = (long long)t * step_rows + b;
    float tot = 0.f;
    long long id[ISC_SPARSE_MAX];
    float cf[ISC_SPARSE_MAX];
    bool any = false;
#pragma unroll
    for (int j = 0; j < ISC_SPARSE_MAX; ++j) {
        id[j] = -1; cf[j] = 0.f;
        if (j < sp.n) { id[j] = sp.ids[j][mi]; cf[j] = sp.coef[j][mi]; tot += cf[j]; any = any || cf[j] != 0.f; }
    }
    float *o = dlogits + ((long long)t * out_step_rows + b) * ld_out;
    if (!any) {
        // every coefficient of this position is zero (a position behind its caption's end: XECriterion's mask,
        // captioner.py:431-436): its d logits row is zero whatever the logits hold - written without reading them or
        // their statistics (block-uniform: the coefficients are per row)
        if (VEC4) {
            for (int i = tid * 4; i < ld_out; i += 1024) *reinterpret_cast<float4 *>(o + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (int i = tid; i < ld_out; i += 256) o[i] = 0.f;
        }
        return;
    }
    if (tid < 64) {
        float gmax, S;
        int gi;
        fold_row_stats_impl(pmax + ms * n_tile, psum + ms * n_tile, nullptr, n_tile, tid, gmax, gi, S);
        if (tid == 0) { sh[0] = gmax; sh[1] = logf(S); }
    }
    __syncthreads();
    const float gmax = sh[0], logS = sh[1];
    const float sc = scale ? scale[0] : 1.f;
    const float *y = raw + (long long)b * ld_b + (long long)t * ld_t;
    auto one = [&](int i, float yi) __attribute__((always_inline)) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < ISC_SPARSE_MAX; ++j)
            if ((long long)i == id[j]) v += cf[j];
        return (v - expf((yi - gmax) - logS) * tot) * sc;
    };
    if (VEC4) {                          // rows start on 16-byte boundaries, V % 4 == 0: four columns per lane and access
        for (int i = tid * 4; i < ld_out; i += 1024) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < V) {
                const float4 yv = *reinterpret_cast<const float4 *>(y + i);
                v.x = one(i, yv.x); v.y = one(i + 1, yv.y); v.z = one(i + 2, yv.z); v.w = one(i + 3, yv.w);
            }
            *reinterpret_cast<float4 *>(o + i) = v;
        }
    } else {
        for (int i = tid; i < ld_out; i += 256) o[i] = i < V ? one(i, y[i]) : 0.f;
    }
}

extern "C" int isc_logsoftmax_bwd_raw(const float *raw, int64_t ld_b, int64_t ld_t, int B, int T, int V, const float *part_max,
                                      const float *part_sum, int step_rows, const int64_t *const *ids_host,
                                      const float *const *coef_host, int n_sparse, const float *scale, float *dlogits,
                                      int64_t ld_out, int out_step_rows, void *stream) {
    if (!raw || !part_max || !part_sum || !dlogits) return ISC_E_NULL;
    if (B <= 0 || T <= 0 || V <= 0 || ld_out < V || (long long)B * T > 2147483647LL) return ISC_E_SHAPE;
    if (n_sparse < 1 || n_sparse > ISC_SPARSE_MAX) return ISC_E_SHAPE;
    if (step_rows == 0) step_rows = B;
    if (out_step_rows == 0) out_step_rows = B;
    if (step_rows < B || out_step_rows < B) return ISC_E_SHAPE;
    SparseDlogp sp = {};
    sp.n = n_sparse;
    for (int j = 0; j < n_sparse; ++j) {
        if (!ids_host || !coef_host || !ids_host[j] || !coef_host[j]) return ISC_E_NULL;
        sp.ids[j] = ids_host[j]; sp.coef[j] = coef_host[j];
    }
    const bool vec4 = (V & 3) == 0 && (ld_out & 3) == 0 && (ld_b & 3) == 0 && (ld_t & 3) == 0 && isc_aligned16(raw) &&
                      isc_aligned16(dlogits);
    if (vec4)
        hipLaunchKernelGGL(logsoftmax_bwd_raw_kernel<true>, dim3((unsigned)(B * T)), dim3(256), 0, (hipStream_t)stream, raw,
                           (long long)ld_b, (long long)ld_t, B, T, V, part_max, part_sum, (V + 127) / 128, step_rows, sp,
                           scale, dlogits, (long long)ld_out, out_step_rows);
    else
        hipLaunchKernelGGL(logsoftmax_bwd_raw_kernel<false>, dim3((unsigned)(B * T)), dim3(256), 0, (hipStream_t)stream, raw,
                           (long long)ld_b, (long long)ld_t, B, T, V, part_max, part_sum, (V + 127) / 128, step_rows, sp,
                           scale, dlogits, (long long)ld_out, out_step_rows);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// Power-of-two gradient scale ("loss scaling" for the split-f16 backward contractions).  The backward pass is linear
// in the gradients that enter it, and those are small - a token-mean loss hands in |d log-prob| <= 1/N_tokens, 5e-5 at
// B = 1024 - i.e. below the f16 normal range (2^-14), where the hi plane of x = hi + lo 2^-11 holds subnormals and an
// element keeps ~22 bits relative to 2^-14, not to itself (measured round 3: gradient errors 5-10x those of the exact
// fp32 engine at B = 1024).  out2 = { S, 1/S } with S = 2^k chosen so that the largest entering |gradient| becomes
// 2^-4..2^-3: the sweep runs on S x gradients (exact in fp32), the parameter gradients are multiplied by 1/S at the
// end (exact), and every element within 2^-10 of the largest splits at full precision, with 2^19 of headroom below
// the f16 maximum.  All zero / non-finite input: S = 1.
struct ScaleSrc {
    const float *p[ISC_SCALE_SRC_MAX];
    long long n[ISC_SCALE_SRC_MAX];
    int count;
};

// Many workgroups sweep the sources (the [B,512] feature gradients are half a million values at B = 1024: one workgroup
// took 150 us), fold their maxima with an integer atomicMax on the bit pattern (order-preserving for non-negative
// floats; deterministic - a maximum does not depend on the order), and the LAST one to finish turns the maximum into
// { S, 1/S } and resets the two state words for the next call.  out4 = { S, 1/S, max bits, arrival counter }; the caller
// provides the last two zeroed once (they are left zeroed).
__global__ __launch_bounds__(256) void grad_scale_kernel(const ScaleSrc src, float *out4) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long gtid = (long long)blockIdx.x * 256 + tid, gstride = (long long)gridDim.x * 256;
    float mx = 0.f;
    for (int k = 0; k < src.count; ++k)
        for (long long i = gtid; i < src.n[k]; i += gstride) mx = fmaxf(mx, fabsf(src.p[k][i]));     // (NaN is skipped)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (tid != 0) return;
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    unsigned int *state = reinterpret_cast<unsigned int *>(out4 + 2);
    atomicMax(&state[0], __float_as_uint(mx));
    __threadfence();
    if (atomicAdd(&state[1], 1u) != gridDim.x - 1) return;
    mx = __uint_as_float(atomicMax(&state[0], 0u));          // every arrival's maximum is in (device-scope atomics)
    float S = 1.f;
    if (mx > 0.f && mx < 3.0e38f) {
        int e;
        frexpf(mx, &e);                      // mx = f * 2^e, f in [0.5, 1)
        int k = -3 - e;                      // S * mx = f * 2^-3  in [2^-4, 2^-3)
        k = k > 60 ? 60 : (k < -60 ? -60 : k);
        S = ldexpf(1.f, k);
    }
    out4[0] = S;
    out4[1] = 1.f / S;
    state[0] = 0u;
    state[1] = 0u;
}

extern "C" int isc_grad_scale(const float *const *src_host, const int64_t *numel_host, int n_src, float *out4,
                              void *stream) {
    if (!src_host || !numel_host || !out4) return ISC_E_NULL;
    if (n_src < 0 || n_src > ISC_SCALE_SRC_MAX) return ISC_E_SHAPE;
    ScaleSrc s = {};
    long long total = 0;
    for (int k = 0; k < n_src; ++k) {
        if (numel_host[k] < 0 || (numel_host[k] > 0 && !src_host[k])) return ISC_E_NULL;
        if (numel_host[k] == 0) continue;
        s.p[s.count] = src_host[k]; s.n[s.count] = numel_host[k]; ++s.count;
        total += numel_host[k];
    }
    long long blocks = (total + 4095) / 4096;                 // ~16 values per thread
    if (blocks < 1) blocks = 1;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(grad_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, s, out4);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ RewardCriterion (self_critical/utils.py:169-177)
// loss = -sum(logp * mask * reward) / sum(mask) over [B,T]; single workgroup, fixed reduction order => bitwise
// reproducible.  out2 = { sum(-logp*mask*reward), sum(mask) }.
__global__ __launch_bounds__(256) void reward_loss_kernel(const float *logp, const float *mask, const float *reward,
                                                          long long n, float *out2) {
    __shared__ float ss[4], sn[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f, c = 0.f;
    for (long long i = tid; i < n; i += 256) {
        const float mk = mask[i];
        s += (-logp[i] * mk) * reward[i];        // the reference's order: (-logp * mask) * reward
        c += mk;
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if (lane == 0) { ss[wave] = s; sn[wave] = c; }
    __syncthreads();
    if (tid == 0) {
        out2[0] = (ss[0] + ss[1]) + (ss[2] + ss[3]);
        out2[1] = (sn[0] + sn[1]) + (sn[2] + sn[3]);
    }
}

extern "C" int isc_reward_loss_fwd(const float *seq_logprobs, const float *seq_masks, const float *reward, int B, int T,
                                   float *out2, void *stream) {
    if (!seq_logprobs || !seq_masks || !reward || !out2) return ISC_E_NULL;
    if (B <= 0 || T <= 0) return ISC_E_SHAPE;
    hipLaunchKernelGGL(reward_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, seq_logprobs, seq_masks, reward,
                       (long long)B * T, out2);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// d seq_logprobs[b,t] = -gout * mask[b,t] * reward[b,t] / sum(mask)   (mask and reward are constants of the criterion)
__global__ __launch_bounds__(256) void reward_loss_bwd_kernel(const float *mask, const float *reward, long long n,
                                                              const float *gout, const float *sum_count,
                                                              float *dlogp) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    dlogp[i] = -(gout[0] / sum_count[1]) * mask[i] * reward[i];
}

extern "C" int isc_reward_loss_bwd(const float *seq_masks, const float *reward, int B, int T, const float *gout,
                                   const float *sum_count, float *d_seq_logprobs, void *stream) {
    if (!seq_masks || !reward || !gout || !sum_count || !d_seq_logprobs) return ISC_E_NULL;
    if (B <= 0 || T <= 0) return ISC_E_SHAPE;
    const long long n = (long long)B * T;
    hipLaunchKernelGGL(reward_loss_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       seq_masks, reward, n, gout, sum_count, d_seq_logprobs);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ LSTM cell backward (pointwise)
// gates = activated (i,f,g,o) saved by the forward kernel.
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float *dh, const float *dh2, const float *dc_next,
                                                       const float *gates, const float *c_prev,
                                                       const float *c, int M, int H, float *dgates,
                                                       float *dc_prev, float *dgates_sum) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)M * H) return;
    const int m = (int)(idx / H), u = (int)(idx % H);
    const float *g = gates + (long long)m * 4 * H + u;
    const float gi = g[0], gf = g[H], gg = g[2 * H], go = g[3 * H];
    float d_h = dh[idx];
    if (dh2) d_h += dh2[idx];
    const float tc = isc_tanh(c[idx]);
    float d_c = d_h * go * (1.f - tc * tc);
    if (dc_next) d_c += dc_next[idx];
    const float di = d_c * gg * gi * (1.f - gi);
    const float df = d_c * c_prev[idx] * gf * (1.f - gf);
    const float dg = d_c * gi * (1.f - gg * gg);
    const float d_o = d_h * tc * go * (1.f - go);
    float *o = dgates + (long long)m * 4 * H + u;
    o[0] = di; o[H] = df; o[2 * H] = dg; o[3 * H] = d_o;
    dc_prev[idx] = d_c * gf;
    if (dgates_sum) {
        float *a = dgates_sum + (long long)m * 4 * H + u;
        a[0] += di; a[H] += df; a[2 * H] += dg; a[3 * H] += d_o;
    }
}

extern "C" int isc_lstm_bwd(const float *dh, const float *dh2, const float *dc_next, const float *gates,
                            const float *c_prev, const float *c, int M, int H, float *dgates,
                            float *dc_prev, float *dgates_sum, void *stream) {
    if (!dh || !gates || !c_prev || !c || !dgates || !dc_prev) return ISC_E_NULL;
    if (M <= 0 || H <= 0) return ISC_E_SHAPE;
    const long long n = (long long)M * H;
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dh, dh2, dc_next, gates, c_prev, c, M, H, dgates, dc_prev,
                       dgates_sum);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ attention scan backward
// d e * w * (1 - tanh^2) with a fixed instruction sequence (one fused 1 - t^2, two multiplications, nothing fused into
// what follows): the per-step kernel and isc_attn_dp_from_de must produce the same bits from the same inputs.
__device__ __forceinline__ float isc_dtanh_term(float der, float w, float t) {
#pragma clang fp contract(off)
    const float omt = __builtin_fmaf(-t, t, 1.0f);
    float r = (der * w) * omt;
    asm volatile("" : "+v"(r));
    return r;
}
struct DevScanBwd {
    const float *P, *V, *q, *q2, *w, *alpha, *dout;
    long long alpha_ld;
    int R, A, D, accumulate;
    float *dP, *dV, *dq, *dw_rows, *de_out;
    int rows;                 // rows of THIS problem (the grid spans the longest problem of the launch)
};
struct DevScanBwdLaunch {
    DevScanBwd p[2];
    int nt;
};
typedef float f4vb __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4b(const float4 *p, bool nt) {
    if (nt) {
        const f4vb v = __builtin_nontemporal_load(reinterpret_cast<const f4vb *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

// One workgroup per (row, problem).  LDS: da[R] (d alpha -> d e), red[ngrp][A] x2 for dq / dw.
// Launched with 1024 threads (16 wavefronts): at B=128 there are only 128 rows, so the parallelism has
// to come from inside the row.  All row traffic is 16 bytes per lane: a thread owns 4 consecutive columns
// (A/4 resp. D/4 threads span a row, 1024 / (A/4) region groups work in parallel).
__global__ __launch_bounds__(1024) void attn_scan_bwd_kernel(const DevScanBwdLaunch L) {
    rows_kernarg_warm<ROWS_KERNARG_LINES(DevScanBwdLaunch)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DevScanBwd &S = L.p[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (b >= S.rows) return;                  // (workgroup-uniform: a shorter problem of a two-problem launch)
    const int NT = blockDim.x, NW = NT >> 6;
    const int R = S.R, A = S.A, D = S.D;
    const int A4 = A >> 2, D4 = D >> 2;
    float *de = smem;                         // [R]
    float *red = smem + ((R + 3) & ~3);       // [2][ngrp][A]
    const float4 *Vb = reinterpret_cast<const float4 *>(S.V + (long long)b * R * D);
    const float4 *Pb = reinterpret_cast<const float4 *>(S.P + (long long)b * R * A);
    const float4 *dout4 = reinterpret_cast<const float4 *>(S.dout + (long long)b * D);
    const float *alpha = S.alpha + (long long)b * S.alpha_ld;

    // d alpha_r = dout . V[r]
    for (int r = wave; r < R; r += NW) {
        float acc = 0.f;
        for (int d = lane; d < D4; d += 64) {
            const float4 g = dout4[d], v = ld4b(&Vb[(long long)r * D4 + d], L.nt);
            acc += g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w;
        }
        acc = wave_sum(acc);
        if (lane == 0) de[r] = acc;
    }
    __syncthreads();
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += alpha[r] * de[r];
    __syncthreads();
    for (int r = tid; r < R; r += NT) {                                  // d e_r (softmax backward)
        const float d = alpha[r] * (de[r] - s);
        de[r] = d;
        if (S.de_out) S.de_out[(long long)b * R + r] = d;                // kept per step: isc_attn_dp_from_de
    }
    __syncthreads();

    // dV[r,:] (+)= alpha_r * dout   (S.dV == nullptr: the caller sums alpha_t x dout_t over the steps in one pass
    // afterwards - isc_attn_dv_from_alpha - instead of re-reading and re-writing dV at every step)
    float4 *dVb = reinterpret_cast<float4 *>(S.dV + (long long)b * R * D);
    if (S.dV) {
        const int dgrp = NT / D4;                         // row groups working in parallel
        const int d = tid % D4, g0 = tid / D4;
        if (g0 < dgrp) {
            const float4 g = dout4[d];
            for (int r = g0; r < R; r += dgrp) {
                const long long o = (long long)r * D4 + d;
                const float al = alpha[r];
                float4 v = make_float4(al * g.x, al * g.y, al * g.z, al * g.w);
                if (S.accumulate) { const float4 c = dVb[o]; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
                dVb[o] = v;
            }
        }
    }
    // dP[r,a] (+)= de_r * w[a] * (1 - tanh^2);  dq[a] = sum_r (...);  dw_rows[b,a] += sum_r de_r * tanh
    float4 *dPb = reinterpret_cast<float4 *>(S.dP + (long long)b * R * A);
    const int ngrp = NT / A4;                             // region groups working in parallel
    const int a4 = tid % A4, grp = tid / A4;
    if (grp < ngrp) {
        float4 qa = reinterpret_cast<const float4 *>(S.q + (long long)b * A)[a4];
        if (S.q2) {
            const float4 t = reinterpret_cast<const float4 *>(S.q2 + (long long)b * A)[a4];
            qa.x += t.x; qa.y += t.y; qa.z += t.z; qa.w += t.w;
        }
        const float4 wa = reinterpret_cast<const float4 *>(S.w)[a4];
        float4 dq = make_float4(0.f, 0.f, 0.f, 0.f), dw = dq;
        for (int r = grp; r < R; r += ngrp) {
            const long long o = (long long)r * A4 + a4;
            const float4 pv = ld4b(&Pb[o], L.nt);
            const float der = de[r];
            const float tx = isc_tanh(pv.x + qa.x), ty = isc_tanh(pv.y + qa.y);
            const float tz = isc_tanh(pv.z + qa.z), tw = isc_tanh(pv.w + qa.w);
            float4 gr = make_float4(isc_dtanh_term(der, wa.x, tx), isc_dtanh_term(der, wa.y, ty),
                                    isc_dtanh_term(der, wa.z, tz), isc_dtanh_term(der, wa.w, tw));
            dq.x += gr.x; dq.y += gr.y; dq.z += gr.z; dq.w += gr.w;
            dw.x += der * tx; dw.y += der * ty; dw.z += der * tz; dw.w += der * tw;
            if (S.dP) {      // (nullptr: dP is formed once after the sweep from the per-step d e - isc_attn_dp_from_de)
                if (S.accumulate) { const float4 c = dPb[o]; gr.x += c.x; gr.y += c.y; gr.z += c.z; gr.w += c.w; }
                dPb[o] = gr;
            }
        }
        reinterpret_cast<float4 *>(red + grp * A)[a4] = dq;
        reinterpret_cast<float4 *>(red + (ngrp + grp) * A)[a4] = dw;
    }
    __syncthreads();
    for (int a = tid; a < A; a += NT) {
        float dq = 0.f, dw = 0.f;
        for (int g = 0; g < ngrp; ++g) { dq += red[g * A + a]; dw += red[(ngrp + g) * A + a]; }
        S.dq[(long long)b * A + a] = dq;
        float *o = S.dw_rows + (long long)b * A + a;
        *o = S.accumulate ? *o + dw : dw;
    }
}

extern "C" int isc_attn_scan_bwd(const isc_scan_bwd_problem *pr, int n_prob, int B, void *stream) {
    if (!pr) return ISC_E_NULL;
    if (n_prob < 1 || n_prob > 2 || B <= 0) return ISC_E_SHAPE;
    DevScanBwdLaunch L = {};
    {   // P / V rows by non-temporal loads once they exceed what stays in the Infinity Cache next to the accumulated
        // dP / dV (attention.hip, isc_attn_scan_fwd): XE iteration at B = 1024 19.9 -> 19.6 ms, B = 512 unchanged
        long long streamed = 0;
        for (int i = 0; i < n_prob; ++i)
            streamed += (long long)(pr[i].rows > 0 ? pr[i].rows : B) * pr[i].R * ((long long)pr[i].A + pr[i].D) * 4;
        L.nt = streamed > (128LL << 20);
    }
    size_t lds = 0;
    int max_rows = 0;
    for (int i = 0; i < n_prob; ++i) {
        const isc_scan_bwd_problem &q = pr[i];
        if (q.rows < 0) return ISC_E_SHAPE;
        L.p[i].rows = q.rows > 0 ? q.rows : B;
        if (L.p[i].rows > max_rows) max_rows = L.p[i].rows;
        if (!q.P || !q.V || !q.q || !q.w || !q.alpha || !q.dout || !q.dq || !q.dw_rows)
            return ISC_E_NULL;                     // (dV / dP may be null: isc_attn_dv_from_alpha / isc_attn_dp_from_de)
        if (!q.dP && !q.de_out) return ISC_E_NULL;
        if (q.R <= 0 || q.A <= 0 || q.D <= 0 || q.A > 1024) return ISC_E_SHAPE;
        if ((q.A & 3) || (q.D & 3) || (1024 % (q.A / 4)) != 0 || (1024 % (q.D / 4)) != 0) return ISC_E_SHAPE;
        if (!isc_aligned16(q.P) || !isc_aligned16(q.V) || (q.dP && !isc_aligned16(q.dP)) || (q.dV && !isc_aligned16(q.dV)) ||
            !isc_aligned16(q.q) || !isc_aligned16(q.w) || !isc_aligned16(q.dout) || (q.q2 && !isc_aligned16(q.q2)))
            return ISC_E_ALIGN;
        DevScanBwd &d = L.p[i];
        d.P = q.P; d.V = q.V; d.q = q.q; d.q2 = q.q2; d.w = q.w; d.alpha = q.alpha; d.dout = q.dout;
        d.alpha_ld = q.alpha_ld; d.R = q.R; d.A = q.A; d.D = q.D; d.accumulate = q.accumulate;
        d.dP = q.dP; d.dV = q.dV; d.dq = q.dq; d.dw_rows = q.dw_rows; d.de_out = q.de_out;
        const int ngrp = 1024 / (q.A / 4);
        const size_t need = (((size_t)q.R + 3) & ~(size_t)3) + (size_t)2 * ngrp * q.A;
        if (need > lds) lds = need;
    }
    lds *= sizeof(float);
    if (lds > 60000) return ISC_E_SHAPE;
    hipLaunchKernelGGL(attn_scan_bwd_kernel, dim3(max_rows, n_prob), dim3(1024), lds, (hipStream_t)stream, L);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// dV[b,r,:] = sum over the steps, in the order of the backward sweep (t = T-1 ... 0), of alpha[b,t,r] * dout[t,b,:]:
// the accumulation that attn_scan_bwd_kernel otherwise does as a read-modify-write of the whole [B,R,D] tensor at every
// step (a third of its traffic), done once from the T small per-step gradients.  Same products, same order of additions.
// Any R, any D % 4 == 0: the grid is (B, region chunks, column blocks of 256 float4); a chunk holds as many regions as
// fit the [T][Rc] LDS image (the 14x14 = 196-region grid of the reference's encoder at T = 20: 4 chunks).  An output
// element's additions do not depend on the chunking.
#define ISC_DV_TMAX 24
#define ISC_POST_LDS_BYTES 60000
__global__ __launch_bounds__(256) void attn_dv_from_alpha_kernel(const float *alpha, long long ld_b, long long ld_t,
                                                                 const float *dout, int B, int T, int R, int D, int Rc,
                                                                 float *dV, int step_rows) {
#pragma clang fp contract(off)
    extern __shared__ float sa[];                  // [T][Rc]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int r_lo = blockIdx.y * Rc, nr = min(Rc, R - r_lo);
    for (int i = tid; i < T * nr; i += 256)
        sa[(i / nr) * Rc + (i % nr)] = alpha[(long long)b * ld_b + (long long)(i / nr) * ld_t + r_lo + (i % nr)];
    __syncthreads();
    const int D4 = D >> 2, c_lo = blockIdx.z * 256, cols = min(256, D4 - c_lo);
    const int ngrp = 256 / cols, d4 = c_lo + tid % cols, grp = tid / cols;
    if (grp >= ngrp) return;
    const float4 *g4 = reinterpret_cast<const float4 *>(dout);
    float4 *o4 = reinterpret_cast<float4 *>(dV + (long long)b * R * D);
    if (T <= ISC_DV_TMAX) {
        float4 g[ISC_DV_TMAX];
#pragma unroll
        for (int t = 0; t < ISC_DV_TMAX; ++t)
            g[t] = t < T ? g4[((long long)t * step_rows + b) * D4 + d4] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = grp; r < nr; r += ngrp) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = ISC_DV_TMAX - 1; t >= 0; --t) {
                if (t < T) {
                    const float al = sa[t * Rc + r];
                    // product rounded, then added (contraction is off in this kernel): what the per-step kernel's
                    // `v = alpha * g; v += previous` compiles to - checked bit for bit by tests/test_gpu_backward.py
                    acc.x = al * g[t].x + acc.x; acc.y = al * g[t].y + acc.y;
                    acc.z = al * g[t].z + acc.z; acc.w = al * g[t].w + acc.w;
                }
            }
            o4[(long long)(r_lo + r) * D4 + d4] = acc;
        }
    } else {
        for (int r = grp; r < nr; r += ngrp) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = T - 1; t >= 0; --t) {
                const float al = sa[t * Rc + r];
                const float4 g = g4[((long long)t * step_rows + b) * D4 + d4];
                acc.x = al * g.x + acc.x; acc.y = al * g.y + acc.y; acc.z = al * g.z + acc.z; acc.w = al * g.w + acc.w;
            }
            o4[(long long)(r_lo + r) * D4 + d4] = acc;
        }
    }
}

extern "C" int isc_attn_dv_from_alpha(const float *alpha, int64_t alpha_ld_b, int64_t alpha_ld_t, const float *dout,
                                      int B, int T, int R, int D, float *dV, int dout_step_rows, void *stream) {
    if (!alpha || !dout || !dV) return ISC_E_NULL;
    if (B <= 0 || T <= 0 || R <= 0 || D <= 0 || (D & 3)) return ISC_E_SHAPE;
    if (dout_step_rows == 0) dout_step_rows = B;
    if (dout_step_rows < B) return ISC_E_SHAPE;
    if (!isc_aligned16(dout) || !isc_aligned16(dV)) return ISC_E_ALIGN;
    int Rc = ISC_POST_LDS_BYTES / (int)sizeof(float) / T;      // regions per chunk: [T][Rc] floats of LDS
    if (Rc < 1) return ISC_E_SHAPE;                             // T > 15000 steps
    if (Rc > R) Rc = R;
    const int ncol = (D / 4 + 255) / 256;
    // few rows: more region chunks until the grid covers the chip twice (an output element's additions do not depend on
    // the chunking; B = 128 with 36 regions was 128 workgroups)
    while (Rc > 4 && (long long)B * ((R + Rc - 1) / Rc) * ncol < 512) Rc = (Rc + 1) / 2;
    const int nchunk = (R + Rc - 1) / Rc;
    if (nchunk > 65535 || ncol > 65535) return ISC_E_SHAPE;
    const size_t lds = (size_t)T * Rc * sizeof(float);
    hipLaunchKernelGGL(attn_dv_from_alpha_kernel, dim3(B, nchunk, ncol), dim3(256), lds, (hipStream_t)stream, alpha,
                       (long long)alpha_ld_b, (long long)alpha_ld_t, dout, B, T, R, D, Rc, dV, dout_step_rows);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// dP[b,r,a] = sum over the steps, in the sweep's order, of d e[t,b,r] * w[a] * (1 - tanh^2(P[b,r,a] + q_t[b,a] (+ q2[b,a]))):
// the other per-step read-modify-write of attn_scan_bwd_kernel, done once from the T small per-step vectors d e_t and
// q_t.  P is read once and kept in registers over the steps; the tanh terms are recomputed (the per-step kernel needs
// them anyway for d q).  Same expression, same order of additions as the per-step accumulation.
// Grid (B, region chunks, column blocks): a thread keeps ISC_DP_RMAX regions in registers, so a chunk is
// ISC_DP_RMAX x (region groups of the workgroup) regions (A = 512: 36 - the 36-region features in one chunk, 196 in six).
#define ISC_DP_RMAX 18
__global__ __launch_bounds__(256) void attn_dp_from_de_kernel(const float *P, const float *q, const float *q2,
                                                              const float *w, const float *de, int B, int T, int R,
                                                              int A, int Rc, float *dP) {
    extern __shared__ float sde[];                 // [T][Rc]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int r_lo = blockIdx.y * Rc, nr = min(Rc, R - r_lo);
    for (int i = tid; i < T * nr; i += 256)
        sde[(i / nr) * Rc + (i % nr)] = de[((long long)(i / nr) * B + b) * R + r_lo + (i % nr)];
    __syncthreads();
    const int A4 = A >> 2, c_lo = blockIdx.z * 256, cols = min(256, A4 - c_lo);
    const int ngrp = 256 / cols, a4 = c_lo + tid % cols, grp = tid / cols;
    if (grp >= ngrp) return;
    const float4 *Pb = reinterpret_cast<const float4 *>(P + ((long long)b * R + r_lo) * A);
    float4 *dPb = reinterpret_cast<float4 *>(dP + ((long long)b * R + r_lo) * A);
    const float4 wa = reinterpret_cast<const float4 *>(w)[a4];
    float4 q2v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q2) q2v = reinterpret_cast<const float4 *>(q2 + (long long)b * A)[a4];
    float4 pv[ISC_DP_RMAX], acc[ISC_DP_RMAX];
#pragma unroll
    for (int i = 0; i < ISC_DP_RMAX; ++i) {
        const int r = grp + i * ngrp;
        pv[i] = r < nr ? Pb[(long long)r * A4 + a4] : make_float4(0.f, 0.f, 0.f, 0.f);
        acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int t = T - 1; t >= 0; --t) {
        float4 qa = reinterpret_cast<const float4 *>(q + ((long long)t * B + b) * A)[a4];
        if (q2) { qa.x += q2v.x; qa.y += q2v.y; qa.z += q2v.z; qa.w += q2v.w; }
#pragma unroll
        for (int i = 0; i < ISC_DP_RMAX; ++i) {
            const int r = grp + i * ngrp;
            if (r < nr) {
                const float der = sde[t * Rc + r];
                const float tx = isc_tanh(pv[i].x + qa.x), ty = isc_tanh(pv[i].y + qa.y);
                const float tz = isc_tanh(pv[i].z + qa.z), tw = isc_tanh(pv[i].w + qa.w);
                const float4 gr = make_float4(isc_dtanh_term(der, wa.x, tx), isc_dtanh_term(der, wa.y, ty),
                                              isc_dtanh_term(der, wa.z, tz), isc_dtanh_term(der, wa.w, tw));
                // (the per-step kernel adds the finished value to what it read back: an add, never fused)
                acc[i].x = gr.x + acc[i].x; acc[i].y = gr.y + acc[i].y;
                acc[i].z = gr.z + acc[i].z; acc[i].w = gr.w + acc[i].w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < ISC_DP_RMAX; ++i) {
        const int r = grp + i * ngrp;
        if (r < nr) dPb[(long long)r * A4 + a4] = acc[i];
    }
}

extern "C" int isc_attn_dp_from_de(const float *P, const float *q, const float *q2, const float *w, const float *de,
                                   int B, int T, int R, int A, float *dP, void *stream) {
    if (!P || !q || !w || !de || !dP) return ISC_E_NULL;
    if (B <= 0 || T <= 0 || R <= 0 || A <= 0 || (A & 3)) return ISC_E_SHAPE;
    if (!isc_aligned16(P) || !isc_aligned16(q) || !isc_aligned16(w) || !isc_aligned16(dP) || (q2 && !isc_aligned16(q2)))
        return ISC_E_ALIGN;
    const int A4 = A / 4, cols = A4 < 256 ? A4 : 256;          // (a ragged last column block only has MORE groups)
    int rmax = ISC_DP_RMAX;                                     // regions a thread keeps in registers
    const int ncol0 = (A4 + 255) / 256;
    // few rows (B = 128, 36 regions: 128 workgroups on 256 CUs, each bound by its 47 M tanh): fewer regions per thread,
    // more region chunks - every (row, region, column) is computed by one thread either way: the same values
    while (rmax > 3 && (long long)B * ((R + rmax * (256 / cols) - 1) / (rmax * (256 / cols))) * ncol0 < 512) rmax = (rmax + 1) / 2;
    int Rc = rmax * (256 / cols);                               // what a workgroup's threads hold in registers
    const int lds_cap = ISC_POST_LDS_BYTES / (int)sizeof(float) / T;
    if (lds_cap < 1) return ISC_E_SHAPE;
    if (Rc > lds_cap) Rc = lds_cap;
    if (Rc > R) Rc = R;
    const int nchunk = (R + Rc - 1) / Rc, ncol = (A4 + 255) / 256;
    if (nchunk > 65535 || ncol > 65535) return ISC_E_SHAPE;
    const size_t lds = (size_t)T * Rc * sizeof(float);
    hipLaunchKernelGGL(attn_dp_from_de_kernel, dim3(B, nchunk, ncol), dim3(256), lds, (hipStream_t)stream, P, q, q2, w,
                       de, B, T, R, A, Rc, dP);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ gate mix backward
// feat = beta v + (1-beta) s, beta = sigmoid(u), u = w . tanh(z) + b
__global__ __launch_bounds__(256) void gate_mix_bwd_kernel(const float *z, const float *w, const float *v,
                                                           const float *s, const float *beta,
                                                           long long beta_ld, const float *dfeat, int B,
                                                           int A, int D, float *dv, float *ds, float *dz,
                                                           float *dw_rows, float *db_rows, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float bt = beta[(long long)b * beta_ld];
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) {
        const long long o = (long long)b * D + d;
        const float g = dfeat[o];
        dot += g * (v[o] - s[o]);
        dv[o] = bt * g;
        ds[o] = (1.f - bt) * g;
    }
    dot = wave_sum(dot);
    const float du = dot * bt * (1.f - bt);
    for (int a = lane; a < A; a += 64) {
        const long long o = (long long)b * A + a;
        const float t = isc_tanh(z[o]);
        dz[o] = du * w[a] * (1.f - t * t);
        dw_rows[o] = accumulate ? dw_rows[o] + du * t : du * t;
    }
    if (lane == 0) db_rows[b] = accumulate ? db_rows[b] + du : du;
}

extern "C" int isc_gate_mix_bwd(const float *z, const float *w, const float *v, const float *s,
                                const float *beta, int64_t beta_ld, const float *dfeat, int B, int A,
                                int D, float *dv, float *ds, float *dz, float *dw_rows, float *db_rows,
                                int accumulate, void *stream) {
    if (!z || !w || !v || !s || !beta || !dfeat || !dv || !ds || !dz || !dw_rows || !db_rows) return ISC_E_NULL;
    if (B <= 0 || A <= 0 || D <= 0) return ISC_E_SHAPE;
    hipLaunchKernelGGL(gate_mix_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, w, v, s,
                       beta, (long long)beta_ld, dfeat, B, A, D, dv, ds, dz, dw_rows, db_rows, accumulate);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ embedding backward (deterministic scatter-add)
// demb[ids[r], :] += scale * dout[r / rows_per_grad, :] * (emb[ids[r], :] > 0) [* mask[r,:]*mask_scale]
// No floating-point atomics: one workgroup per input position r; it works only if r is the FIRST position
// holding its id, then sums every position with that id in ascending order and adds the total to the row it
// alone owns in this launch.  The id array (<= 80 KB) stays in L2, so the "am I first" test over the
// earlier positions and the match scan over the later ones cost ~N/256 cached reads per thread.
struct EmbIds {
    const int64_t *ids;
    long long ids_stride, pad_id, skip_id;
    int pad_first;
    __device__ __forceinline__ long long at(int r) const {
        if (pad_first) {            // senti-word layout: row = b*(n+1)+m, m==0 is the <PAD> prefix
            const int Mw = pad_first, b = r / Mw, m = r % Mw;
            return (m == 0) ? pad_id : ids[(long long)b * (Mw - 1) + (m - 1)];
        }
        return ids[(long long)r * ids_stride];
    }
};

#define ISC_EMB_ROUND 2048   // positions compacted per round (LDS list)
template <bool MASK>
__global__ __launch_bounds__(256) void embed_relu_bwd_kernel(const float *emb, int W, EmbIds I, int n_rows,
                                                             int rows_per_grad, const float *dout,
                                                             float scale, const uint8_t *mask,
                                                             float mask_scale, float *demb) {
    __shared__ int list[ISC_EMB_ROUND];
    __shared__ int cnt[256];
    __shared__ int flag;
    const int tid = threadIdx.x, r = blockIdx.x;
    const long long id = I.at(r);
    if (tid == 0) flag = 0;
    __syncthreads();
    int seen = 0;
    for (int q = tid; q < r; q += 256) seen |= (I.at(q) == id);
    if (seen) flag = 1;                       // benign race: every writer stores 1
    __syncthreads();
    if (flag || id == I.skip_id) return;     // an earlier position owns this id / the caller discards this row
    float acc[4] = {0.f, 0.f, 0.f, 0.f};      // columns tid, tid+256, ... (W <= 1024)
    for (int base = r; base < n_rows; base += ISC_EMB_ROUND) {
        // ordered compaction of the matches in [base, base + ROUND): thread t scans a contiguous slice
        const int per = ISC_EMB_ROUND / 256, q0 = base + tid * per;
        int mine[ISC_EMB_ROUND / 256], n = 0;
#pragma unroll
        for (int k = 0; k < per; ++k) {
            const int q = q0 + k;
            const bool hit = q < n_rows && I.at(q) == id;
            mine[k] = hit ? q : -1;
            n += hit;
        }
        cnt[tid] = n;
        __syncthreads();
        int off = 0, total = 0;
        for (int t = 0; t < 256; ++t) {       // 256 LDS broadcasts; the launch is tiny next to the GEMMs
            const int c = cnt[t];
            off += (t < tid) ? c : 0;
            total += c;
        }
#pragma unroll
        for (int k = 0; k < per; ++k)
            if (mine[k] >= 0) list[off++] = mine[k];
        __syncthreads();
        // ascending position order = a fixed summation order; 8 rows' loads in flight at a time (a frequent
        // token is ONE workgroup walking hundreds of rows: without this the walk is a chain of L2 round
        // trips).  Loads are unconditional - out-of-range slots re-read a valid element with weight 0 -
        // because a predicated load is a branch, and branches serialise the 16 loads of a group again.
        int ci[4];
        float cw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + 256 * k;
            ci[k] = i < W ? i : 0;
            cw[k] = i < W ? 1.f : 0.f;
        }
        for (int m0 = 0; m0 < total; m0 += 8) {
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool live = m0 + u < total;
                const int q = list[live ? m0 + u : total - 1];
                const float wq = live ? scale : 0.f;
                const float *g = dout + (long long)(q / rows_per_grad) * W;
                if (MASK) {
                    const uint8_t *mk = mask + (long long)q * W;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[u][k] = g[ci[k]] * (wq * cw[k]) * ((float)mk[ci[k]] * mask_scale);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[u][k] = g[ci[k]] * (wq * cw[k]);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] += v[u][k];      // weight-0 slots add 0
        }
        __syncthreads();
    }
    const float *e = emb + id * W;
    float *o = demb + id * W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + 256 * k;
        if (i < W && e[i] > 0.f) o[i] += acc[k];
    }
}

extern "C" int isc_embed_relu_bwd(const float *emb, int V, int W, const int64_t *ids, int64_t ids_stride,
                                  int n_rows, int rows_per_grad, int pad_first, int64_t pad_id,
                                  const float *dout, float scale, const uint8_t *keep_mask,
                                  float mask_scale, float *demb, int64_t skip_id, void *stream) {
    if (!emb || !ids || !dout || !demb) return ISC_E_NULL;
    if (n_rows <= 0 || W <= 0 || W > 1024 || V <= 0 || rows_per_grad <= 0) return ISC_E_SHAPE;
    EmbIds I = {ids, (long long)ids_stride, (long long)pad_id, (long long)skip_id, pad_first};
    if (keep_mask)
        hipLaunchKernelGGL(embed_relu_bwd_kernel<true>, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, emb, W, I,
                           n_rows, rows_per_grad, dout, scale, keep_mask, mask_scale, demb);
    else
        hipLaunchKernelGGL(embed_relu_bwd_kernel<false>, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, emb, W, I,
                           n_rows, rows_per_grad, dout, scale, keep_mask, mask_scale, demb);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ---- the same gradient through a position index (isc_embed_relu_bwd_ws) ----
// The kernel above finds the positions of an id by scanning the id array: every workgroup reads the ids in front of
// it (am I the first?) and the owner every id behind it - O(n^2) reads in the token count (561 us for the 20 480
// fed tokens of an XE step at B = 1024).  With a workspace the positions are indexed first: (1) count[id]++
// (integer atomics: the counts are exact whatever the order), (2) one workgroup turns the counts into offsets and lists
// the ids that occur, in id order, (3) every position drops itself into its id's segment (integer atomic cursor: the
// order INSIDE a segment is arbitrary), (4) one workgroup per occurring id sorts its segment ascending in LDS and sums the
// rows in that order, eight loads in flight - the summation order of the kernel above, so the two agree bit for bit.
__global__ __launch_bounds__(256) void emb_zero_kernel(int *p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0;
}

__global__ __launch_bounds__(256) void emb_count_kernel(EmbIds I, int n_rows, int V, int *count) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const long long id = I.at(r);
    if (id == I.skip_id || id < 0 || id >= V) return;
    atomicAdd(&count[id], 1);
}

// offset[id] = exclusive prefix of count; active[0 .. n_active) = ids with count > 0, ascending.  One workgroup.
__global__ __launch_bounds__(1024) void emb_scan_kernel(const int *count, int V, int *offset, int *active,
                                                        int *n_active) {
    __shared__ int sc[1024], sa[1024];
    const int tid = threadIdx.x, per = (V + 1023) / 1024, i0 = tid * per;
    int c = 0, a = 0;
    for (int k = 0; k < per; ++k) {
        const int i = i0 + k;
        if (i < V) { c += count[i]; a += count[i] > 0; }
    }
    sc[tid] = c; sa[tid] = a;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                   // inclusive Hillis-Steele scans of both
        const int vc = tid >= o ? sc[tid - o] : 0, va = tid >= o ? sa[tid - o] : 0;
        __syncthreads();
        sc[tid] += vc; sa[tid] += va;
        __syncthreads();
    }
    int oc = sc[tid] - c, oa = sa[tid] - a;
    for (int k = 0; k < per; ++k) {
        const int i = i0 + k;
        if (i < V) {
            const int n = count[i];
            offset[i] = oc;
            if (n > 0) active[oa++] = i;
            oc += n;
        }
    }
    if (tid == 1023) *n_active = sa[1023];
}

__global__ __launch_bounds__(256) void emb_fill_kernel(EmbIds I, int n_rows, int V, const int *offset, int *cursor,
                                                       int *list) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const long long id = I.at(r);
    if (id == I.skip_id || id < 0 || id >= V) return;
    list[offset[id] + atomicAdd(&cursor[id], 1)] = r;
}

template <bool MASK>
__global__ __launch_bounds__(256) void emb_accumulate_kernel(const float *emb, int W, int n_rows, int rows_per_grad,
                                                             const float *dout, float scale, const uint8_t *mask,
                                                             float mask_scale, float *demb, const int *count,
                                                             const int *offset, const int *active,
                                                             const int *n_active, const int *seg_list) {
    __shared__ int raw[ISC_EMB_ROUND], list[ISC_EMB_ROUND];
    __shared__ int m_sh;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= *n_active) return;
    const long long id = active[blockIdx.x];
    const int s = count[id];
    const int *seg = seg_list + offset[id];
    // a segment that fits the LDS list is sorted in one go; a longer one window by window over the positions
    const int nwin = s <= ISC_EMB_ROUND ? 1 : (n_rows + ISC_EMB_ROUND - 1) / ISC_EMB_ROUND;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};      // columns tid, tid+256, ... (W <= 1024)
    int ci[4];
    float cw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + 256 * k;
        ci[k] = i < W ? i : 0;
        cw[k] = i < W ? 1.f : 0.f;
    }
    for (int win = 0; win < nwin; ++win) {
        if (tid == 0) m_sh = 0;
        __syncthreads();
        if (nwin == 1) {
            for (int i = tid; i < s; i += 256) raw[i] = seg[i];
            if (tid == 0) m_sh = s;
        } else {
            for (int i = tid; i < s; i += 256) {
                const int q = seg[i];
                if (q / ISC_EMB_ROUND == win) raw[atomicAdd(&m_sh, 1)] = q;
            }
        }
        __syncthreads();
        const int total = m_sh;
        for (int e = tid; e < total; e += 256) {          // positions are distinct: rank = how many are smaller
            const int v = raw[e];
            int rank = 0;
            for (int j = 0; j < total; ++j) rank += raw[j] < v;
            list[rank] = v;
        }
        __syncthreads();
        for (int m0 = 0; m0 < total; m0 += 8) {           // as embed_relu_bwd_kernel: ascending positions, 8 in flight
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool live = m0 + u < total;
                const int q = list[live ? m0 + u : total - 1];
                const float wq = live ? scale : 0.f;
                const float *g = dout + (long long)(q / rows_per_grad) * W;
                if (MASK) {
                    const uint8_t *mk = mask + (long long)q * W;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[u][k] = g[ci[k]] * (wq * cw[k]) * ((float)mk[ci[k]] * mask_scale);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[u][k] = g[ci[k]] * (wq * cw[k]);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] += v[u][k];      // weight-0 slots add 0
        }
        __syncthreads();
    }
    const float *e = emb + id * W;
    float *o = demb + id * W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + 256 * k;
        if (i < W && e[i] > 0.f) o[i] += acc[k];
    }
}

extern "C" int isc_embed_relu_bwd_ws(const float *emb, int V, int W, const int64_t *ids, int64_t ids_stride,
                                     int n_rows, int rows_per_grad, int pad_first, int64_t pad_id,
                                     const float *dout, float scale, const uint8_t *keep_mask,
                                     float mask_scale, float *demb, int64_t skip_id, void *workspace,
                                     int64_t workspace_bytes, void *stream) {
    if (!emb || !ids || !dout || !demb) return ISC_E_NULL;
    if (n_rows <= 0 || W <= 0 || W > 1024 || V <= 0 || rows_per_grad <= 0) return ISC_E_SHAPE;
    const int64_t need = ((int64_t)4 * V + 64 + n_rows) * (int64_t)sizeof(int);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15) || n_rows < 1024)
        return isc_embed_relu_bwd(emb, V, W, ids, ids_stride, n_rows, rows_per_grad, pad_first, pad_id, dout, scale,
                                  keep_mask, mask_scale, demb, skip_id, stream);       // few positions: the scan is cheap
    hipStream_t st = (hipStream_t)stream;
    int *count = static_cast<int *>(workspace), *cursor = count + V, *offset = cursor + V, *active = offset + V;
    int *n_active = active + V, *list = n_active + 64;
    // count and cursor start at zero - by a kernel, not hipMemsetAsync: a captured memset NODE did not always clear them on
    // replay (training graphs: counts piled up over replays, the fill kernel wrote past the list, the accumulate kernel read
    // slots nobody had written - GPU memory faults in emb_fill_kernel / emb_accumulate_kernel)
    hipLaunchKernelGGL(emb_zero_kernel, dim3((unsigned)((2 * V + 255) / 256)), dim3(256), 0, st, count, 2 * V);
    ISC_LAUNCH_CHECK();             // (a failed clear must not let count / fill / accumulate run on stale counters)
    EmbIds I = {ids, (long long)ids_stride, (long long)pad_id, (long long)skip_id, pad_first};
    const unsigned nb = (unsigned)((n_rows + 255) / 256);
    hipLaunchKernelGGL(emb_count_kernel, dim3(nb), dim3(256), 0, st, I, n_rows, V, count);
    hipLaunchKernelGGL(emb_scan_kernel, dim3(1), dim3(1024), 0, st, count, V, offset, active, n_active);
    hipLaunchKernelGGL(emb_fill_kernel, dim3(nb), dim3(256), 0, st, I, n_rows, V, offset, cursor, list);
    const unsigned na = (unsigned)(V < n_rows ? V : n_rows);       // an upper bound of the ids that occur
    if (keep_mask)
        hipLaunchKernelGGL(emb_accumulate_kernel<true>, dim3(na), dim3(256), 0, st, emb, W, n_rows, rows_per_grad, dout,
                           scale, keep_mask, mask_scale, demb, count, offset, active, n_active, list);
    else
        hipLaunchKernelGGL(emb_accumulate_kernel<false>, dim3(na), dim3(256), 0, st, emb, W, n_rows, rows_per_grad, dout,
                           scale, keep_mask, mask_scale, demb, count, offset, active, n_active, list);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ column sums (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float *x, long long ld, int M, int N, float *out,
                                                     int accumulate) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (n < N)
        for (int m = grp; m < M; m += 4) s += x[(long long)m * ld + n];
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && n < N) {
        const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        out[n] = accumulate ? out[n] + t : t;
    }
}

// Tall matrices: stage 1 sums row chunks in parallel into part[chunk][N], stage 2 (the kernel above on the
// [chunks, N] partials) finishes in chunk order - deterministic, and no workgroup walks thousands of rows.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *x, long long ld, int M, int N,
                                                             int rows_per_chunk, float *part) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int m0 = blockIdx.y * rows_per_chunk;
    int m1 = m0 + rows_per_chunk;
    if (m1 > M) m1 = M;
    float s = 0.f;
    if (n < N)
        for (int m = m0 + grp; m < m1; m += 4) s += x[(long long)m * ld + n];
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && n < N)
        part[(long long)blockIdx.y * N + n] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

extern "C" int isc_colsum(const float *x, int64_t ld, int M, int N, float *out, int accumulate,
                          float *workspace, int64_t workspace_floats, void *stream) {
    if (!x || !out) return ISC_E_NULL;
    if (M <= 0 || N <= 0) return ISC_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    int chunks = (M + 63) / 64;
    if (chunks > 64) chunks = 64;
    if (workspace && M >= 256 && (int64_t)chunks * N <= workspace_floats) {
        const int rpc = (M + chunks - 1) / chunks;
        chunks = (M + rpc - 1) / rpc;
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, chunks), dim3(256), 0, st, x, (long long)ld, M,
                           N, rpc, workspace);
        ISC_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, st, workspace, (long long)N, chunks, N,
                           out, accumulate);
    } else {
        hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, st, x, (long long)ld, M, N, out,
                           accumulate);
    }
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// Many column sums in (at most) two launches: the bias gradients of one backward sweep are ~15 independent
// reductions of 5-9 us each, most of them a launch floor.  Phase 1: every job's (column block, row chunk) partial -
// written straight to the job's outputs when the job has one chunk; phase 2 (only if some job has several chunks):
// the chunk partials in chunk order.  Same 4-way row interleave and pairwise finish as colsum_kernel; deterministic.
// A job may have up to three outputs (tied biases: b_ih / b_hh of an LSTM, the three biases summed into the gate).
struct DevColJob {
    const float *x;
    long long ld;
    int M, N, chunks, rpc, blk0, blk1;          // blk0 / blk1: first block of the job in phase 1 / phase 2
    long long part_off;
    float *out[ISC_COLSUM_MAX_OUT];
    int nout, accumulate;
};
struct DevColLaunch {
    DevColJob j[ISC_COLSUM_MAX_JOBS];
    int njobs;
};

__device__ __forceinline__ void colsum_store(const DevColJob &J, int n, float t) {
    for (int o = 0; o < J.nout; ++o) J.out[o][n] = J.accumulate ? J.out[o][n] + t : t;
}

__global__ __launch_bounds__(256) void colsum_multi_kernel(const DevColLaunch L, float *part) {
    __shared__ float red[4][64];
    int ji = 0;
    while (ji + 1 < L.njobs && (int)blockIdx.x >= L.j[ji + 1].blk0) ++ji;
    const DevColJob &J = L.j[ji];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int b = blockIdx.x - J.blk0, nb = (J.N + 63) / 64, cb = b % nb, chunk = b / nb;
    const int n = cb * 64 + lane;
    const int m0 = chunk * J.rpc;
    int m1 = m0 + J.rpc;
    if (m1 > J.M) m1 = J.M;
    float s = 0.f;
    if (n < J.N)
        for (int m = m0 + grp; m < m1; m += 4) s += J.x[(long long)m * J.ld + n];
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && n < J.N) {
        const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (J.chunks == 1) colsum_store(J, n, t);
        else part[J.part_off + (long long)chunk * J.N + n] = t;
    }
}

__global__ __launch_bounds__(64) void colsum_multi_finish_kernel(const DevColLaunch L, const float *part) {
    int ji = -1;
    for (int i = 0; i < L.njobs; ++i)
        if (L.j[i].chunks > 1 && (int)blockIdx.x >= L.j[i].blk1) ji = i;
    const DevColJob &J = L.j[ji];
    const int n = (blockIdx.x - J.blk1) * 64 + threadIdx.x;
    if (n >= J.N) return;
    float t = 0.f;
    for (int c = 0; c < J.chunks; ++c) t += part[J.part_off + (long long)c * J.N + n];
    colsum_store(J, n, t);
}

extern "C" int isc_colsum_multi(const isc_colsum_job *jobs, int n_jobs, float *workspace, int64_t workspace_floats,
                                void *stream) {
    if (!jobs) return ISC_E_NULL;
    if (n_jobs < 1 || n_jobs > ISC_COLSUM_MAX_JOBS) return ISC_E_SHAPE;
    DevColLaunch L = {};
    L.njobs = n_jobs;
    int blocks1 = 0, blocks2 = 0;
    long long part = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const isc_colsum_job &q = jobs[i];
        if (!q.x || q.n_out < 1 || q.n_out > ISC_COLSUM_MAX_OUT) return q.x ? ISC_E_SHAPE : ISC_E_NULL;
        if (q.M <= 0 || q.N <= 0) return ISC_E_SHAPE;
        DevColJob &J = L.j[i];
        J.x = q.x; J.ld = q.ld; J.M = q.M; J.N = q.N; J.nout = q.n_out; J.accumulate = q.accumulate;
        for (int o = 0; o < q.n_out; ++o) {
            if (!q.out[o]) return ISC_E_NULL;
            J.out[o] = q.out[o];
        }
        int chunks = 1;
        if (q.M >= 256) {
            chunks = (q.M + 63) / 64;
            if (chunks > 64) chunks = 64;
        }
        J.rpc = (q.M + chunks - 1) / chunks;
        J.chunks = (q.M + J.rpc - 1) / J.rpc;
        const int nb = (q.N + 63) / 64;
        J.blk0 = blocks1; blocks1 += nb * J.chunks;
        J.blk1 = blocks2; J.part_off = part;
        if (J.chunks > 1) { blocks2 += nb; part += (long long)J.chunks * q.N; }
    }
    if (part > 0 && (!workspace || part > workspace_floats)) return ISC_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(blocks1), dim3(256), 0, st, L, workspace);
    ISC_LAUNCH_CHECK();
    if (blocks2 > 0) {
        hipLaunchKernelGGL(colsum_multi_finish_kernel, dim3(blocks2), dim3(64), 0, st, L, workspace);
        ISC_LAUNCH_CHECK();
    }
    return ISC_OK;
}

// ------------------------------------------------------------------ ReLU (+dropout) backward
__global__ __launch_bounds__(256) void relu_mask_bwd_kernel(const float *dy, const float *y,
                                                            const uint8_t *mask, float scale, long long n,
                                                            float *dz) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float g = (!y || y[i] > 0.f) ? dy[i] : 0.f;
    if (mask) g *= (float)mask[i] * scale;
    dz[i] = g;
}

extern "C" int isc_relu_mask_bwd(const float *dy, const float *y, const uint8_t *keep_mask, float scale,
                                 int64_t n, float *dz, void *stream) {
    if (!dy || !dz) return ISC_E_NULL;
    if (n <= 0) return ISC_E_SHAPE;
    hipLaunchKernelGGL(relu_mask_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dy, y, keep_mask, scale, (long long)n, dz);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ XECriterion backward
// dlogp must be zero-filled by the caller; scatters -g/count at the target of every unmasked token.
__global__ __launch_bounds__(256) void xe_loss_bwd_kernel(const int64_t *target, const int *lengths, int B,
                                                          int T, int V, const float *gout,
                                                          const float *sum_count, float *dlogp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i % T;
    if (t < lengths[b]) dlogp[(long long)i * V + target[i]] = -gout[0] / sum_count[1];
}

// The same gradient in the sparse form isc_logsoftmax_bwd_sparse takes: coef[b,t] = -gout/count at unmasked tokens, 0
// elsewhere; the column is the target itself.
__global__ __launch_bounds__(256) void xe_loss_bwd_coef_kernel(const int *lengths, int B, int T, const float *gout,
                                                               const float *sum_count, float *coef) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * T) return;
    coef[i] = (i % T) < lengths[i / T] ? -gout[0] / sum_count[1] : 0.f;
}

extern "C" int isc_xe_loss_bwd_sparse(const int32_t *lengths, int B, int T, const float *gout, const float *sum_count,
                                      float *coef, void *stream) {
    if (!lengths || !gout || !sum_count || !coef) return ISC_E_NULL;
    if (B <= 0 || T <= 0) return ISC_E_SHAPE;
    hipLaunchKernelGGL(xe_loss_bwd_coef_kernel, dim3((B * T + 255) / 256), dim3(256), 0, (hipStream_t)stream, lengths,
                       B, T, gout, sum_count, coef);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

extern "C" int isc_xe_loss_bwd(const int64_t *target, const int32_t *lengths, int B, int T, int V,
                               const float *gout, const float *sum_count, float *dlogp, void *stream) {
    if (!target || !lengths || !gout || !sum_count || !dlogp) return ISC_E_NULL;
    if (B <= 0 || T <= 0 || V <= 0) return ISC_E_SHAPE;
    hipLaunchKernelGGL(xe_loss_bwd_kernel, dim3((B * T + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       target, lengths, B, T, V, gout, sum_count, dlogp);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ fused clamp + Adam (multi-tensor)
struct DevAdam {
    float *p[ISC_ADAM_MAX_TENSORS];
    float *g[ISC_ADAM_MAX_TENSORS];
    float *m[ISC_ADAM_MAX_TENSORS];
    float *v[ISC_ADAM_MAX_TENSORS];
    int blk_start[ISC_ADAM_MAX_TENSORS + 1];
    long long n[ISC_ADAM_MAX_TENSORS];
    int count;
    float lr, b1, b2, eps, clip, bc1, bc2_sqrt, wd;
    const float *hyper;      // device {lr, bc1, bc2_sqrt} (a captured launch: the step-dependent scalars cannot be arguments)
};

__global__ __launch_bounds__(256) void clamp_adam_kernel(const DevAdam Aa) {
    int ti = 0;
    while (ti + 1 < Aa.count && (int)blockIdx.x >= Aa.blk_start[ti + 1]) ++ti;
    const long long base = ((long long)blockIdx.x - Aa.blk_start[ti]) * 1024;
    float *p = Aa.p[ti], *g = Aa.g[ti], *m = Aa.m[ti], *v = Aa.v[ti];
    const long long n = Aa.n[ti];
    const float lr = Aa.hyper ? Aa.hyper[0] : Aa.lr, bc1 = Aa.hyper ? Aa.hyper[1] : Aa.bc1,
                bc2_sqrt = Aa.hyper ? Aa.hyper[2] : Aa.bc2_sqrt;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i >= n) break;
        float gr = g[i];
        if (Aa.clip > 0.f) { gr = fminf(fmaxf(gr, -Aa.clip), Aa.clip); g[i] = gr; }  // clamp_ is in place
        if (Aa.wd != 0.f) gr += Aa.wd * p[i];
        const float mm = Aa.b1 * m[i] + (1.f - Aa.b1) * gr;
        const float vv = Aa.b2 * v[i] + (1.f - Aa.b2) * gr * gr;
        m[i] = mm;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + Aa.eps;
        p[i] -= (lr / bc1) * (mm / denom);
    }
}

static int clamp_adam_launch(float *const *params, float *const *grads, float *const *exp_avg,
                             float *const *exp_avg_sq, const int64_t *numel, int n_tensors, double lr,
                             double beta1, double beta2, double eps, double weight_decay, double clip,
                             int step, const float *hyper, void *stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel) return ISC_E_NULL;
    if (n_tensors <= 0 || step <= 0) return ISC_E_SHAPE;
    for (int off = 0; off < n_tensors; off += ISC_ADAM_MAX_TENSORS) {
        DevAdam A = {};
        const int cnt = (n_tensors - off) < ISC_ADAM_MAX_TENSORS ? (n_tensors - off) : ISC_ADAM_MAX_TENSORS;
        int blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            if (!params[off + i] || !grads[off + i] || !exp_avg[off + i] || !exp_avg_sq[off + i]) return ISC_E_NULL;
            A.p[i] = params[off + i]; A.g[i] = grads[off + i];
            A.m[i] = exp_avg[off + i]; A.v[i] = exp_avg_sq[off + i];
            A.n[i] = numel[off + i];
            A.blk_start[i] = blocks;
            blocks += (int)((numel[off + i] + 1023) / 1024);
        }
        A.blk_start[cnt] = blocks;
        A.count = cnt;
        A.lr = (float)lr; A.b1 = (float)beta1; A.b2 = (float)beta2; A.eps = (float)eps;
        A.clip = (float)clip; A.wd = (float)weight_decay;
        // bias corrections in double, like torch.optim.Adam's Python-side scalars
        A.bc1 = (float)(1.0 - pow(beta1, (double)step));
        A.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
        A.hyper = hyper;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(clamp_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, A);
        ISC_LAUNCH_CHECK();
    }
    return ISC_OK;
}

extern "C" int isc_clamp_adam(float *const *params, float *const *grads, float *const *exp_avg,
                              float *const *exp_avg_sq, const int64_t *numel, int n_tensors, double lr,
                              double beta1, double beta2, double eps, double weight_decay, double clip,
                              int step, void *stream) {
    return clamp_adam_launch(params, grads, exp_avg, exp_avg_sq, numel, n_tensors, lr, beta1, beta2, eps, weight_decay,
                             clip, step, nullptr, stream);
}

extern "C" int isc_clamp_adam_hyper(float *const *params, float *const *grads, float *const *exp_avg,
                                    float *const *exp_avg_sq, const int64_t *numel, int n_tensors,
                                    const float *hyper3_dev, double beta1, double beta2, double eps,
                                    double weight_decay, double clip, void *stream) {
    if (!hyper3_dev) return ISC_E_NULL;
    return clamp_adam_launch(params, grads, exp_avg, exp_avg_sq, numel, n_tensors, 0.0, beta1, beta2, eps, weight_decay,
                             clip, 1, hyper3_dev, stream);
}
