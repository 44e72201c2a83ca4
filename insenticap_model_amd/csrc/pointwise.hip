// Embedding gathers, roll-out bookkeeping, log-softmax finalisation, beam top-k and the
// masked-NLL criterion (captioner.py:170-172, 201-202, 307-311, 329-344, 394-408, 427-440).
#include <atomic>
#include <cstddef>

#include "common.h"

// ------------------------------------------------------------------ embeddings
__global__ __launch_bounds__(256) void embed_relu_kernel(const float *emb, int W, const int64_t *ids,
                                                         long long ids_stride, const float *add,
                                                         int B, float *out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const long long id = ids[(long long)b * ids_stride];
    const float4 *src = reinterpret_cast<const float4 *>(emb + id * W);
    const float4 *ad = add ? reinterpret_cast<const float4 *>(add + (long long)b * W) : nullptr;
    float4 *dst = reinterpret_cast<float4 *>(out + (long long)b * W);
    for (int i = lane; i < (W >> 2); i += 64) {
        float4 v = src[i];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (ad) { const float4 a = ad[i]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
        dst[i] = v;
    }
}

extern "C" int isc_embed_relu_fwd(const float *emb, int V, int W, const int64_t *ids,
                                  int64_t ids_stride, const float *add, int B, float *out,
                                  void *stream) {
    if (!emb || !ids || !out) return ISC_E_NULL;
    if (B <= 0 || V <= 0 || W <= 0 || (W & 3)) return ISC_E_SHAPE;
    hipLaunchKernelGGL(embed_relu_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, emb, W,
                       ids, (long long)ids_stride, add, B, out);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

__global__ __launch_bounds__(256) void embed_relu_mean_kernel(const float *emb, int W, const int64_t *ids,
                                                              int C, int B, float *out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    if ((W & 3) == 0 && (((uintptr_t)emb | (uintptr_t)out) & 15) == 0) {
        // the ids of up to eight words first, then every word's row chunk in flight at once (the rows come from an 82 MB
        // table: one dependent round trip per word and chunk made this a 20-27 us kernel at B = 128 ... 208); the sum
        // runs over the words in ascending order, as before
        for (int i = lane * 4; i < W; i += 256) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int c0 = 0; c0 < C; c0 += 8) {
                long long id[8];
                float4 x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) id[k] = ids[(long long)b * C + (c0 + k < C ? c0 + k : C - 1)];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = *reinterpret_cast<const float4 *>(emb + id[k] * W + i);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c0 + k < C) {
                        s.x += fmaxf(x[k].x, 0.f); s.y += fmaxf(x[k].y, 0.f);
                        s.z += fmaxf(x[k].z, 0.f); s.w += fmaxf(x[k].w, 0.f);
                    }
            }
            const float n = (float)C;
            *reinterpret_cast<float4 *>(out + (long long)b * W + i) = make_float4(s.x / n, s.y / n, s.z / n, s.w / n);
        }
        return;
    }
    for (int i = lane; i < W; i += 64) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += fmaxf(emb[ids[(long long)b * C + c] * W + i], 0.f);
        out[(long long)b * W + i] = s / (float)C;
    }
}

extern "C" int isc_embed_relu_mean_fwd(const float *emb, int V, int W, const int64_t *ids, int C,
                                       int B, float *out, void *stream) {
    if (!emb || !ids || !out) return ISC_E_NULL;
    if (B <= 0 || V <= 0 || W <= 0 || C <= 0) return ISC_E_SHAPE;
    hipLaunchKernelGGL(embed_relu_mean_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       emb, W, ids, C, B, out);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

__global__ __launch_bounds__(256) void embed_senti_words_kernel(const float *emb, int W,
                                                                const int64_t *ids, int n_words,
                                                                long long pad_id, int B,
                                                                const uint8_t *mask, float scale,
                                                                float *out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);  // row = b*(n_words+1) + m
    const int Mw = n_words + 1;
    if (row >= B * Mw) return;
    const int b = row / Mw, m = row % Mw;
    const long long id = (m == 0) ? pad_id : ids[(long long)b * n_words + (m - 1)];
    for (int i = lane; i < W; i += 64) {
        float v = fmaxf(emb[id * W + i], 0.f);
        if (mask) v = v * (float)mask[(long long)row * W + i] * scale;
        out[(long long)row * W + i] = v;
    }
}

extern "C" int isc_embed_senti_words_fwd(const float *emb, int V, int W, const int64_t *ids,
                                         int n_words, int64_t pad_id, int B,
                                         const uint8_t *keep_mask, float mask_scale, float *out,
                                         void *stream) {
    if (!emb || !ids || !out) return ISC_E_NULL;
    if (B <= 0 || V <= 0 || W <= 0 || n_words <= 0) return ISC_E_SHAPE;
    const int rows = B * (n_words + 1);
    hipLaunchKernelGGL(embed_senti_words_kernel, dim3((rows + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, emb, W, ids, n_words, (long long)pad_id, B, keep_mask,
                       mask_scale, out);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ beam search: candidate merge on the device
// captioner.py:378-411 for every image of a batched search, without the per-step device->host->device round trip:
// one workgroup per image.  Candidates in the reference's insertion order (parents by rank; an ended parent carries
// itself, a live one contributes its `beam` children by rank), scores as fp64 sums of the fp32 log-probs (the
// reference adds Python floats), selection = the first `beam` of a STABLE descending sort:
//     rank(c) = #{ j : s_j > s_c } + #{ j < c : s_j == s_c }.
// State is double-buffered by the caller (in -> out).  gather[row] = source row of the new row inside
// [next-state rows ; current-state rows] (a stepped parent's new state, or a carried candidate's old one).
#define ISC_BEAM_MAX 8
__global__ __launch_bounds__(64) void beam_merge_kernel(const isc_beam_merge_args a) {
    __shared__ double cs[ISC_BEAM_MAX * ISC_BEAM_MAX];
    __shared__ long long ctok[ISC_BEAM_MAX * ISC_BEAM_MAX];
    __shared__ int cpar[ISC_BEAM_MAX * ISC_BEAM_MAX], ccar[ISC_BEAM_MAX * ISC_BEAM_MAX], coff[ISC_BEAM_MAX + 1];
    __shared__ int s_all_ended;
    const int i = blockIdx.x, tid = threadIdx.x, beam = a.beam, T = a.T, rows = a.n_img * beam, base = i * beam;
    if (a.done[i]) {                      // frozen image: state carried over unchanged, rows keep their old state
        for (int k = tid; k < beam; k += 64) {
            a.gather[base + k] = base + k + rows;
            a.last_out[base + k] = a.last_in[base + k];
            a.score_out[base + k] = a.score_in[base + k];
            a.len_out[base + k] = a.len_in[base + k];
        }
        for (int e = tid; e < beam * T; e += 64) a.words_out[(long long)base * T + e] = a.words_in[(long long)base * T + e];
        return;
    }
    const int ncand = a.t == 0 ? 1 : beam;
    if (tid == 0) {
        int off = 0, all_ended = 1;
        for (int k = 0; k < ncand; ++k) {
            coff[k] = off;
            const int ended = a.t > 0 && a.last_in[base + k] == a.eos_id;
            off += ended ? 1 : beam;
            all_ended &= ended;
        }
        coff[ncand] = off;
        s_all_ended = all_ended;
    }
    __syncthreads();
    const int n = coff[ncand];
    for (int e = tid; e < ncand * beam; e += 64) {
        const int k = e / beam, j = e % beam, row = base + k;
        const int ended = a.t > 0 && a.last_in[row] == a.eos_id;
        if (ended) {
            if (j == 0) { const int c = coff[k]; cs[c] = a.score_in[row]; ctok[c] = a.last_in[row]; cpar[c] = k; ccar[c] = 1; }
        } else {
            const int c = coff[k] + j;
            cs[c] = a.score_in[row] + (double)a.top_val[(long long)row * beam + j];
            ctok[c] = a.top_idx[(long long)row * beam + j];
            cpar[c] = k; ccar[c] = 0;
        }
    }
    __syncthreads();
    if (tid < n) {
        const double sc = cs[tid];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (cs[j] > sc) || (cs[j] == sc && j < tid);
        if (rank < beam) {
            const int par = cpar[tid], car = ccar[tid], dst = base + rank, srow = base + par;
            a.score_out[dst] = sc;
            a.last_out[dst] = ctok[tid];
            a.gather[dst] = car ? srow + rows : srow;
            const int len = a.len_in[srow];
            for (int e = 0; e < T; ++e) a.words_out[(long long)dst * T + e] = a.words_in[(long long)srow * T + e];
            if (!car && len < T) a.words_out[(long long)dst * T + len] = ctok[tid];
            a.len_out[dst] = len + (car ? 0 : 1);
        }
    }
    if (tid == 0) {
        if (s_all_ended) a.done[i] = 1;
        else atomicAdd(&a.live[a.t + 1], 1);
    }
}

extern "C" int isc_beam_merge(const isc_beam_merge_args *args, void *stream) {
    if (!args) return ISC_E_NULL;
    const isc_beam_merge_args &a = *args;
    if (!a.top_val || !a.top_idx || !a.score_in || !a.score_out || !a.last_in || !a.last_out || !a.words_in ||
        !a.words_out || !a.len_in || !a.len_out || !a.done || !a.gather || !a.live)
        return ISC_E_NULL;
    if (a.n_img <= 0 || a.beam <= 0 || a.beam > ISC_BEAM_MAX || a.T <= 0 || a.t < 0 || a.t >= a.T) return ISC_E_SHAPE;
    hipLaunchKernelGGL(beam_merge_kernel, dim3(a.n_img), dim3(64), 0, (hipStream_t)stream, a);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// New recurrent state of every beam row: out[p, r, :] = (gather[r] < rows ? next : current)[p, gather[r] % rows, :]
// for the P = 4 planes (h|c x layer) of a [P, rows, H] state - one launch instead of a concatenation + index_select.
__global__ __launch_bounds__(256) void beam_gather_kernel(const float *nxt, const float *cur, const int64_t *gather,
                                                          float *out, int rows, int H) {
    const int r = blockIdx.x, p = blockIdx.y;
    const long long g = gather[r];
    const float *src = (g < rows ? nxt : cur) + ((long long)p * rows + (g < rows ? g : g - rows)) * H;
    float *dst = out + ((long long)p * rows + r) * H;
    for (int i = threadIdx.x; i < (H >> 2); i += 256)
        reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(src)[i];
}

extern "C" int isc_beam_gather(const float *state_next, const float *state_cur, const int64_t *gather, float *out,
                               int planes, int rows, int H, void *stream) {
    if (!state_next || !state_cur || !gather || !out) return ISC_E_NULL;
    if (planes <= 0 || rows <= 0 || H <= 0 || (H & 3)) return ISC_E_SHAPE;
    if (!isc_aligned16(state_next) || !isc_aligned16(state_cur) || !isc_aligned16(out)) return ISC_E_ALIGN;
    hipLaunchKernelGGL(beam_gather_kernel, dim3(rows, planes), dim3(256), 0, (hipStream_t)stream, state_next, state_cur,
                       gather, out, rows, H);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ row statistics helper
// Folds the per-tile (max, sumexp, argmax) triples of one row: returns the global max, its
// vocabulary index (smallest index on ties) and S = sum exp(x - gmax).  All 64 lanes get
// the result.
// Numerics status (isc_status, include/insenticap_hip.h): word 0 = a row of vocabulary statistics held a non-finite
// maximum or sum.  Every decode path folds the step's tile statistics through fold_row_stats (roll-out finalize,
// scheduled sampling, the log-softmax passes, beam top-k), so whatever went non-finite upstream in the step - an operand
// beyond the split-f16 domain |x| < 65504 whose hi plane became inf, a NaN feature (ReLU lets NaN through: isc_relu) -
// is flagged within the same step at no measurable cost (one compare per row) instead of surfacing as garbage tokens.
ISC_STATUS_DECL(pw)

__device__ __forceinline__ void fold_row_stats(const float *pmax, const float *psum, const int *pidx,
                                               int n_tile, int lane, float &gmax, int &gidx, float &S) {
    if (fold_row_stats_impl(pmax, psum, pidx, n_tile, lane, gmax, gidx, S) && lane == 0)
        isc_flag_pw(ISC_STATUS_WORD_STATS);
}

// ------------------------------------------------------------------ roll-out step
struct DevRollout {
    int B, V, T, t, n_tile, W;
    const float *part_max, *part_sum;
    const int *part_idx;
    const float *logits;
    long long ld_logits;
    const int64_t *forced;
    const float *sample_u;
    long long eos_id;
    int64_t *seq;
    float *seq_logprobs, *seq_masks;
    int *unfinished, *alive;
    int64_t *raw_tokens;
    const float *emb, *xt_add;
    float *xt_next;
    int rows_per_wave;
};

// Inverse-CDF sampling from softmax over one row, in vocabulary order, by one wavefront.  Two levels: the
// vocabulary kernel already left sum exp(x - tile max) per 128-column tile, so the tile holding the target
// mass is found by a prefix over n_tile values and only its 128 columns are exponentiated (4 wave-prefix
// rounds per row instead of V/64 = 157: 300 us -> ~12 us per roll-out step at B=512).
// Element mass = exp(x[i] - shift) * scale: (gmax, 1) for raw logits, (0, S) for log-probabilities, so that
// both agree with the tile masses psum[j] * exp(pmax[j] - gmax) and with target = u * S.
// Returns the drawn id, or -1 when rounding put the target past the total mass.
__device__ __forceinline__ int sample_two_level(const float *x, float shift, float scale, const float *pm,
                                                const float *ps, int n_tile, int V, float gmax, float target,
                                                int lane) {
    auto wave_incl = [&](float v) __attribute__((always_inline)) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float n = __shfl_up(v, o, 64);
            if (lane >= o) v += n;
        }
        return v;
    };
    float run = 0.f;
    int tile = -1;
    for (int base = 0; base < n_tile && tile < 0; base += 64) {
        const int j = base + lane;
        const float w = (j < n_tile) ? ps[j] * expf(pm[j] - gmax) : 0.f;
        const float incl = wave_incl(w);
        const unsigned long long hit = __ballot((run + incl > target) && j < n_tile);
        if (hit) {
            const int l = __ffsll((long long)hit) - 1;
            tile = base + l;
            run += __shfl(incl - w, l, 64);       // mass in front of the tile
        } else {
            run += __shfl(incl, 63, 64);
        }
    }
    if (tile < 0) return -1;
    int pick = -1;
    for (int c = 0; c < 2 && pick < 0; ++c) {
        const int i = tile * 128 + c * 64 + lane;
        const float e = (i < V) ? expf(x[i] - shift) * scale : 0.f;
        const float incl = wave_incl(e);
        const unsigned long long hit = __ballot((run + incl > target) && i < V);
        if (hit) pick = tile * 128 + c * 64 + __ffsll((long long)hit) - 1;
        run += __shfl(incl, 63, 64);
    }
    if (pick < 0) {                               // the tile's own sum rounded differently: its last column
        pick = tile * 128 + 127;
        if (pick > V - 1) pick = V - 1;
    }
    return pick;
}

#define ISC_FIN_ROWS_PER_WAVE 4
// One wavefront per row, 4 rows per wave, 16 rows per workgroup; the count of still-unfinished rows
// is reduced inside the workgroup so that the `alive` counter sees one atomic per 16 rows (4096
// same-address atomics used to dominate this kernel).
__device__ __forceinline__ int rollout_finalize_row(const DevRollout &R, int b, int lane) {
    float gmax, S;
    int gidx;
    fold_row_stats(R.part_max + (long long)b * R.n_tile, R.part_sum + (long long)b * R.n_tile,
                   R.part_idx + (long long)b * R.n_tile, R.n_tile, lane, gmax, gidx, S);
    const float logS = logf(S);
    long long it;
    float lp;
    if (R.forced) {
        it = R.forced[(long long)b * R.T + R.t];
        lp = (R.logits[(long long)b * R.ld_logits + it] - gmax) - logS;
    } else if (R.sample_u) {
        const float *x = R.logits + (long long)b * R.ld_logits;
        int pick = sample_two_level(x, gmax, 1.0f, R.part_max + (long long)b * R.n_tile,
                                    R.part_sum + (long long)b * R.n_tile, R.n_tile, R.V, gmax,
                                    R.sample_u[(long long)b * R.T + R.t] * S, lane);
        if (pick < 0) pick = gidx;  // target == S after rounding
        it = pick;
        lp = (x[it] - gmax) - logS;
    } else {
        it = gidx;
        lp = -logS;  // log_softmax at the arg-max: (x_max - x_max) - log S
    }
    const int u = R.unfinished[b];
    const long long itm = u ? it : 0;  // finished rows feed <PAD> (id 0): `it * unfinished`
    const int u2 = u && (itm != R.eos_id);
    if (lane == 0) {
        const long long o = (long long)b * R.T + R.t;
        R.seq_masks[o] = (float)u;
        R.seq[o] = itm;
        R.seq_logprobs[o] = lp;
        if (R.raw_tokens) R.raw_tokens[o] = it;
        R.unfinished[b] = u2;
    }
    if (R.xt_next) {
        const float4 *src = reinterpret_cast<const float4 *>(R.emb + itm * R.W);
        const float4 *ad = R.xt_add ? reinterpret_cast<const float4 *>(R.xt_add + (long long)b * R.W) : nullptr;
        float4 *dst = reinterpret_cast<float4 *>(R.xt_next + (long long)b * R.W);
        for (int i = lane; i < (R.W >> 2); i += 64) {
            float4 v = src[i];
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            if (ad) { const float4 a = ad[i]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
            dst[i] = v;
        }
    }
    return u2;
}

__global__ __launch_bounds__(256) void rollout_finalize_kernel(const DevRollout R) {
    __shared__ int cnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the reference `break`s once no row is unfinished (captioner.py:343-344): later steps
    // leave seq / seq_logprobs / seq_masks at their zero initialisation
    if (R.alive[R.t] == 0) return;          // block-uniform
    int alive = 0;
    for (int i = 0; i < R.rows_per_wave; ++i) {
        const int b = (blockIdx.x * 4 + wave) * R.rows_per_wave + i;
        if (b < R.B) alive += rollout_finalize_row(R, b, lane);
    }
    if (lane == 0) cnt[wave] = alive;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        if (tot) atomicAdd(&R.alive[R.t + 1], tot);
    }
}

// Many rows: sixteen waves per workgroup, ONE row per wave (every row's dependent chain - two strided sweeps, a dozen
// cross-lane steps, logf, stores - runs at the same time as all the others) and still one `alive` atomic per 16 rows.
// (Round 3: the four-rows-per-wave form of rounds 1-2 walked its rows one after the other: 16.8 us at B = 4096.)
__global__ __launch_bounds__(1024) void rollout_finalize_wide_kernel(const DevRollout R) {
    __shared__ int cnt[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (R.alive[R.t] == 0) return;          // block-uniform
    const int b = blockIdx.x * 16 + wave;
    int alive = 0;
    if (b < R.B) alive = rollout_finalize_row(R, b, lane);
    if (lane == 0) cnt[wave] = alive;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += cnt[w];
        if (tot) atomicAdd(&R.alive[R.t + 1], tot);
    }
}

static std::atomic<long long> g_finalize_launches{0};
extern "C" long long isc_rollout_finalize_launches(void) { return g_finalize_launches.load(); }

extern "C" int isc_rollout_finalize(const isc_rollout_step *s, void *stream) {
    if (!s) return ISC_E_NULL;
    if (!s->part_max || !s->part_sum || !s->part_idx || !s->seq || !s->seq_logprobs || !s->seq_masks ||
        !s->unfinished || !s->alive || !s->emb)
        return ISC_E_NULL;
    if ((s->forced || s->sample_u) && !s->logits) return ISC_E_NULL;
    if (s->B <= 0 || s->T <= 0 || s->t < 0 || s->t >= s->T || (s->W & 3)) return ISC_E_SHAPE;
    DevRollout R;
    R.B = s->B; R.V = s->V; R.T = s->T; R.t = s->t; R.n_tile = s->n_tile; R.W = s->W;
    R.part_max = s->part_max; R.part_sum = s->part_sum; R.part_idx = s->part_idx;
    R.logits = s->logits; R.ld_logits = s->ld_logits; R.forced = s->forced; R.sample_u = s->sample_u;
    R.eos_id = s->eos_id; R.seq = s->seq; R.seq_logprobs = s->seq_logprobs; R.seq_masks = s->seq_masks;
    R.unfinished = s->unfinished; R.alive = s->alive; R.raw_tokens = s->raw_tokens;
    R.emb = s->emb; R.xt_add = s->xt_add; R.xt_next = s->xt_next;
    // one row per wavefront; many rows: 16 waves per workgroup, so that the `alive` counter sees one atomic per 16 rows
    R.rows_per_wave = 1;
    if (s->B > 1024)
        hipLaunchKernelGGL(rollout_finalize_wide_kernel, dim3((s->B + 15) / 16), dim3(1024), 0, (hipStream_t)stream, R);
    else
        hipLaunchKernelGGL(rollout_finalize_kernel, dim3((s->B + 3) / 4), dim3(256), 0, (hipStream_t)stream, R);
    ISC_LAUNCH_CHECK();
    ++g_finalize_launches;
    return ISC_OK;
}

// ------------------------------------------------------------------ scheduled sampling (captioner.py:219-228)
// out[b] = (u_select[b] < ss_prob) ? draw from exp(logp[b, :]) with uniform u_draw[b] : base[b].
// One launch and no host decision in place of rand / `if mask.sum() == 0` / multinomial / index_copy_:
// the reference's host test costs a device->host sync at every step.
// RAW: `logp` holds the raw logits of the row (the statistics describe them): the draw is then the roll-out's
// (exp(x - max) against tile masses psum * exp(pmax - max)) and a teacher-forced unroll with scheduled sampling can turn
// its logits into log-probs ONCE, after the last step, instead of step by step.
template <bool RAW>
__global__ __launch_bounds__(256) void sched_sample_kernel(const float *logp, long long ld, int M, int V,
                                                           const float *pmax, const float *psum, const int *pidx,
                                                           int n_tile, const float *u_select, const float *u_draw,
                                                           float ss_prob, const int64_t *base, long long base_stride,
                                                           int64_t *out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= M) return;
    long long it = base[(long long)b * base_stride];
    if (u_select[b] < ss_prob) {                  // wave-uniform
        float gmax, S;
        int gidx;
        fold_row_stats(pmax + (long long)b * n_tile, psum + (long long)b * n_tile, pidx + (long long)b * n_tile,
                       n_tile, lane, gmax, gidx, S);
        const int pick = sample_two_level(logp + (long long)b * ld, RAW ? gmax : 0.f, RAW ? 1.0f : S,
                                          pmax + (long long)b * n_tile, psum + (long long)b * n_tile, n_tile, V, gmax,
                                          u_draw[b] * S, lane);
        it = pick < 0 ? gidx : pick;
    }
    if (lane == 0) out[b] = it;
}

static int sched_sample_launch(bool raw, const float *logp, int64_t ld, int M, int V, const float *part_max,
                               const float *part_sum, const int32_t *part_idx, const float *u_select,
                               const float *u_draw, float ss_prob, const int64_t *base_ids, int64_t base_stride,
                               int64_t *out_ids, void *stream) {
    if (!logp || !part_max || !part_sum || !part_idx || !u_select || !u_draw || !base_ids || !out_ids)
        return ISC_E_NULL;
    if (M <= 0 || V <= 0) return ISC_E_SHAPE;
    if (raw)
        hipLaunchKernelGGL(sched_sample_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, logp,
                           (long long)ld, M, V, part_max, part_sum, part_idx, (V + 127) / 128, u_select, u_draw, ss_prob,
                           base_ids, (long long)base_stride, out_ids);
    else
        hipLaunchKernelGGL(sched_sample_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, logp,
                           (long long)ld, M, V, part_max, part_sum, part_idx, (V + 127) / 128, u_select, u_draw, ss_prob,
                           base_ids, (long long)base_stride, out_ids);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

extern "C" int isc_sched_sample(const float *logp, int64_t ld, int M, int V, const float *part_max,
                                const float *part_sum, const int32_t *part_idx, const float *u_select,
                                const float *u_draw, float ss_prob, const int64_t *base_ids, int64_t base_stride,
                                int64_t *out_ids, void *stream) {
    return sched_sample_launch(false, logp, ld, M, V, part_max, part_sum, part_idx, u_select, u_draw, ss_prob, base_ids,
                               base_stride, out_ids, stream);
}

extern "C" int isc_sched_sample_raw(const float *logits, int64_t ld, int M, int V, const float *part_max,
                                    const float *part_sum, const int32_t *part_idx, const float *u_select,
                                    const float *u_draw, float ss_prob, const int64_t *base_ids, int64_t base_stride,
                                    int64_t *out_ids, void *stream) {
    return sched_sample_launch(true, logits, ld, M, V, part_max, part_sum, part_idx, u_select, u_draw, ss_prob, base_ids,
                               base_stride, out_ids, stream);
}

// ------------------------------------------------------------------ several small device-to-device copies, one launch
// (the input copies in front of a graph replay: fc / att features, word ids, labels of one image are 300 KB in four
// tensors of two dtypes - torch._foreach_copy_ is one 17 us + one 4.5 us launch for them)
struct DevCopyMulti {
    unsigned char *dst[ISC_COPY_MULTI_MAX];
    const unsigned char *src[ISC_COPY_MULTI_MAX];
    long long bytes[ISC_COPY_MULTI_MAX];
    int first_block[ISC_COPY_MULTI_MAX + 1];      // blocks of 16 KB, prefix over the entries
    int n;
};
__global__ __launch_bounds__(256) void copy_multi_kernel(const DevCopyMulti a) {
    int e = 0;
#pragma unroll
    for (int i = 1; i < ISC_COPY_MULTI_MAX; ++i) e += (i < a.n && (int)blockIdx.x >= a.first_block[i]) ? 1 : 0;
    unsigned char *dst = a.dst[0];
    const unsigned char *src = a.src[0];
    long long nb = a.bytes[0];
    int fb = a.first_block[0];
#pragma unroll
    for (int i = 1; i < ISC_COPY_MULTI_MAX; ++i)
        if (e == i) { dst = a.dst[i]; src = a.src[i]; nb = a.bytes[i]; fb = a.first_block[i]; }
    const long long off = (long long)((int)blockIdx.x - fb) * 16384;
    const long long end = off + 16384 < nb ? off + 16384 : nb;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        for (long long o = off + (long long)threadIdx.x * 16; o < end; o += 256 * 16) {
            if (o + 16 <= end) {
                *reinterpret_cast<float4 *>(dst + o) = *reinterpret_cast<const float4 *>(src + o);
            } else {
                for (long long q = o; q < end; ++q) dst[q] = src[q];
            }
        }
    } else {
        for (long long o = off + threadIdx.x; o < end; o += 256) dst[o] = src[o];
    }
}

extern "C" int isc_copy_multi(void *const *dst, const void *const *src, const int64_t *bytes, int n, void *stream) {
    if (!dst || !src || !bytes) return ISC_E_NULL;
    if (n < 1 || n > ISC_COPY_MULTI_MAX) return ISC_E_SHAPE;
    DevCopyMulti a = {};
    a.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (bytes[i] < 0) return ISC_E_SHAPE;
        if (bytes[i] > 0 && (!dst[i] || !src[i])) return ISC_E_NULL;
        a.dst[i] = static_cast<unsigned char *>(dst[i]);
        a.src[i] = static_cast<const unsigned char *>(src[i]);
        a.bytes[i] = bytes[i];
        a.first_block[i] = blocks;
        if (bytes[i] > (1LL << 34)) return ISC_E_SHAPE;
        blocks += (int)((bytes[i] + 16383) / 16384);
    }
    for (int i = n; i <= ISC_COPY_MULTI_MAX; ++i) a.first_block[i] = blocks;
    if (blocks == 0) return ISC_OK;
    hipLaunchKernelGGL(copy_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ log-softmax apply
__global__ __launch_bounds__(256) void logsoftmax_apply_kernel(float *logits, long long ld, int M, int V,
                                                               const float *pmax, const float *psum,
                                                               int n_tile, float *lse_out) {
    __shared__ float sh[2];
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid < 64) {
        float gmax, S;
        int gi;
        fold_row_stats(pmax + (long long)m * n_tile, psum + (long long)m * n_tile, nullptr, n_tile, tid,
                       gmax, gi, S);
        if (tid == 0) {
            sh[0] = gmax;
            sh[1] = logf(S);
            if (lse_out) lse_out[m] = gmax + sh[1];
        }
    }
    __syncthreads();
    const float gmax = sh[0], logS = sh[1];
    float *x = logits + (long long)m * ld;
    for (int i = blockIdx.y * 256 + tid; i < V; i += gridDim.y * 256) x[i] = (x[i] - gmax) - logS;
}

extern "C" int isc_logsoftmax_apply(float *logits, int64_t ld_logits, int M, int V,
                                    const float *part_max, const float *part_sum, float *lse_out,
                                    void *stream) {
    if (!logits || !part_max || !part_sum) return ISC_E_NULL;
    if (M <= 0 || V <= 0) return ISC_E_SHAPE;
    const int n_tile = (V + 127) / 128;
    int gy = (V + 2047) / 2048;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(logsoftmax_apply_kernel, dim3(M, gy), dim3(256), 0, (hipStream_t)stream, logits,
                       (long long)ld_logits, M, V, part_max, part_sum, n_tile, lse_out);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// The same for ALL T steps of a teacher-forced unroll in one launch: logits [B,T,V] (row (b,t) at b*ld_b + t*ld_t), tile
// statistics stacked per step [T,B,n_tile].  With every fed token known up front nothing reads a step's log-probs
// before the unroll ends, so the T per-step launches (38 of the ~750 of an XE iteration at B = 128) become one.
__global__ __launch_bounds__(256) void logsoftmax_apply_steps_kernel(float *logits, long long ld_b, long long ld_t,
                                                                     int B, int V, const float *pmax,
                                                                     const float *psum, int n_tile,
                                                                     const float *src, int step_rows) {
    __shared__ float sh[2];
    const int mi = blockIdx.x;                // = t * B + b
    // row of (t, b) in the per-step stacks: step_rows rows per step (>= B: the stacks of a merged unroll hold the rows
    // of both branches per step; statistics / src then point at this branch's first row)
    const long long m = (long long)(mi / B) * step_rows + (mi % B);
    const int tid = threadIdx.x;
    if (tid < 64) {
        float gmax, S;
        int gi;
        fold_row_stats(pmax + m * n_tile, psum + m * n_tile, nullptr, n_tile, tid,
                       gmax, gi, S);
        if (tid == 0) {
            sh[0] = gmax;
            sh[1] = logf(S);
        }
    }
    __syncthreads();
    const float gmax = sh[0], logS = sh[1];
    float *x = logits + (long long)(mi % B) * ld_b + (long long)(mi / B) * ld_t;
    const float *y = src ? src + m * V : x;       // src: raw logits stacked per step, [T*step_rows, V]
    for (int i = blockIdx.y * 256 + tid; i < V; i += gridDim.y * 256) x[i] = (y[i] - gmax) - logS;
}

extern "C" int isc_logsoftmax_apply_steps(float *logits, int64_t ld_b, int64_t ld_t, int B, int T, int V,
                                          const float *part_max, const float *part_sum, const float *src,
                                          int step_rows, void *stream) {
    if (!logits || !part_max || !part_sum) return ISC_E_NULL;
    if (B <= 0 || T <= 0 || V <= 0 || (long long)B * T > 2147483647LL) return ISC_E_SHAPE;
    if (step_rows == 0) step_rows = B;
    if (step_rows < B) return ISC_E_SHAPE;
    const int n_tile = (V + 127) / 128;
    int gy = (V + 2047) / 2048;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(logsoftmax_apply_steps_kernel, dim3(B * T, gy), dim3(256), 0, (hipStream_t)stream, logits,
                       (long long)ld_b, (long long)ld_t, B, V, part_max, part_sum, n_tile, src, step_rows);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ beam top-k
// Form for beam <= 8 over the tile statistics the classifier already produced: the top-`beam` masked log-probs can
// only sit in column tiles whose maximum is among the (beam + 4) largest tile maxima - at most 4 words are masked
// (<PAD>, <SOS>, <UNK>, the repeated last word), so at least `beam` unmasked entries reach that threshold and nothing
// below it can be selected.  Those ~9 of 79 tiles are swept once, every thread keeping its best 8 in registers, then
// `beam` rounds of a block-wide arg-max over the threads' list heads.  Same values and order as the kernel below
// (descending value, ties to the smaller word id, masked entries as -inf): [5 x 10000] 78 us -> ~10 us, which was
// 40 % of a one-image search.
__global__ __launch_bounds__(256) void beam_topk8_kernel(const float *logits, long long ld,
                                                         const float *pmax, const float *psum, int n_tile,
                                                         int V, int beam, const int64_t *last_word,
                                                         long long pad_id, long long sos_id,
                                                         long long unk_id, int mask_special, int cons,
                                                         float *top_val, int64_t *top_idx, const int *gate) {
    if (gate != nullptr && *gate == 0) return;      // (uniform; isc_set_stream_gate: the search has ended)
    constexpr int K = 8;
    __shared__ float sv[2][4];
    __shared__ int si[2][4];
    __shared__ float sh[2];
    __shared__ int sel[256];               // selected tile ids (n_tile <= 256 on this path)
    __shared__ float spm[256];             // this row's tile maxima
    __shared__ int nsel;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) nsel = 0;
    if (tid >= 64 && tid - 64 < n_tile) spm[tid - 64] = pmax[(long long)row * n_tile + tid - 64];   // (waves 1-3 idle here)
    if (tid < 64) {
        if (n_tile > 192) for (int j = 192 + tid; j < n_tile; j += 64) spm[j] = pmax[(long long)row * n_tile + j];
        float gmax, S;
        int gi;
        fold_row_stats(pmax + (long long)row * n_tile, psum + (long long)row * n_tile, nullptr, n_tile,
                       tid, gmax, gi, S);
        if (tid == 0) { sh[0] = gmax; sh[1] = logf(S); }
    }
    __syncthreads();
    const float gmax = sh[0], logS = sh[1];
    // tile selection: rank of this thread's tile among the tile maxima (ties by tile id); tiles tied with the last
    // admitted maximum are admitted as well because the comparison below is on values, not ranks
    const int want = beam + 4;
    if (tid < n_tile) {
        const float mine = spm[tid];
        int larger = 0;
        for (int j = 0; j < n_tile; ++j) larger += spm[j] > mine;     // LDS broadcasts (was n_tile global loads)
        if (larger < want) sel[atomicAdd(&nsel, 1)] = tid;     // order of `sel` is irrelevant: ids break ties later
    }
    __syncthreads();
    const int ns = nsel;
    const float *x = logits + (long long)row * ld;
    const long long last = last_word[row];
    float tv[K];
    int ti[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { tv[k] = -INFINITY; ti[k] = 0x7fffffff; }
    auto offer = [&](float raw, int i) __attribute__((always_inline)) {
        bool banned = false;
        if (mask_special && (i == pad_id || i == sos_id || i == unk_id)) banned = true;
        if (cons && i == last) banned = true;
        float v = banned ? -INFINITY : (raw - gmax) - logS;
        int id = i;
#pragma unroll
        for (int k = 0; k < K; ++k) {      // insertion into the sorted list (best first)
            const bool better = v > tv[k] || (v == tv[k] && id < ti[k]);
            const float ov = tv[k];
            const int oi = ti[k];
            tv[k] = better ? v : ov;
            ti[k] = better ? id : oi;
            v = better ? ov : v;
            id = better ? oi : id;
        }
    };
    // two selected tiles per pass: thread t takes column (t & 127) of tile sel[2*pass + (t >> 7)]; the loads of four
    // passes are issued together (the ~9 selected tiles used to be five dependent-looking L2 round trips)
    for (int s0 = 0; s0 < ns; s0 += 8) {
        float xv[4];
        int xi[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int si2 = s0 + 2 * u + (tid >> 7);
            ok[u] = si2 < ns;
            xi[u] = ok[u] ? sel[si2] * 128 + (tid & 127) : 0;
            ok[u] = ok[u] && xi[u] < V;
            xv[u] = x[ok[u] ? xi[u] : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u]) offer(xv[u], xi[u]);
    }
    for (int k = 0; k < beam; ++k) {
        float mx = tv[0];
        int ix = ti[0];
        wave_argmax(mx, ix);
        if (lane == 0) { sv[k & 1][wave] = mx; si[k & 1][wave] = ix; }
        __syncthreads();
        float bm = sv[k & 1][0];
        int bi = si[k & 1][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float wv2 = sv[k & 1][w];
            const int wi = si[k & 1][w];
            if (wv2 > bm || (wv2 == bm && wi < bi)) { bm = wv2; bi = wi; }
        }
        if (ti[0] == bi) {                 // this thread's head won: pop it (word ids are unique)
#pragma unroll
            for (int j = 0; j + 1 < K; ++j) { tv[j] = tv[j + 1]; ti[j] = ti[j + 1]; }
            tv[K - 1] = -INFINITY; ti[K - 1] = 0x7fffffff;
        }
        if (tid == 0) {
            top_val[(long long)row * beam + k] = bm;
            top_idx[(long long)row * beam + k] = (unsigned)bi < (unsigned)V ? bi : 0;      // (NaN rows: sentinel -> <PAD>)
        }
    }
}

// One workgroup per live beam row; `beam` rounds of a block-wide arg-max over the masked
// log-probabilities (rows are few: images x beam).  Ties resolve to the smaller word id.
__global__ __launch_bounds__(256) void beam_topk_kernel(const float *logits, long long ld,
                                                        const float *pmax, const float *psum, int n_tile,
                                                        int V, int beam, const int64_t *last_word,
                                                        long long pad_id, long long sos_id,
                                                        long long unk_id, int mask_special, int cons,
                                                        float *top_val, int64_t *top_idx, const int *gate) {
    if (gate != nullptr && *gate == 0) return;      // (uniform; isc_set_stream_gate: the search has ended)
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float sh[2];
    __shared__ int chosen[16];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) {
        float gmax, S;
        int gi;
        fold_row_stats(pmax + (long long)row * n_tile, psum + (long long)row * n_tile, nullptr, n_tile,
                       tid, gmax, gi, S);
        if (tid == 0) { sh[0] = gmax; sh[1] = logf(S); }
    }
    __syncthreads();
    const float gmax = sh[0], logS = sh[1];
    const float *x = logits + (long long)row * ld;
    const long long last = last_word[row];
    for (int k = 0; k < beam; ++k) {
        float mx = -INFINITY;
        int ix = 0x7fffffff;
        for (int i = tid; i < V; i += 256) {
            bool banned = false;
            if (mask_special && (i == pad_id || i == sos_id || i == unk_id)) banned = true;
            if (cons && i == last) banned = true;
            for (int j = 0; j < k; ++j) banned |= (chosen[j] == i);
            const float v = banned ? -INFINITY : (x[i] - gmax) - logS;
            // banned entries still take part as -inf (torch.sort keeps them, at the tail)
            if (v > mx || (v == mx && i < ix)) { mx = v; ix = i; }
        }
        wave_argmax(mx, ix);
        if (lane == 0) { sv[wave] = mx; si[wave] = ix; }
        __syncthreads();
        if (tid == 0) {
            float bm = sv[0];
            int bi = si[0];
            for (int w = 1; w < 4; ++w)
                if (sv[w] > bm || (sv[w] == bm && si[w] < bi)) { bm = sv[w]; bi = si[w]; }
            chosen[k] = bi;
            top_val[(long long)row * beam + k] = bm;
            top_idx[(long long)row * beam + k] = (unsigned)bi < (unsigned)V ? bi : 0;      // (NaN rows: sentinel -> <PAD>)
        }
        __syncthreads();
    }
}

extern "C" int isc_beam_topk(const float *logits, int64_t ld_logits, const float *part_max,
                             const float *part_sum, int n_tile, int rows, int V, int beam,
                             const int64_t *last_word, int64_t pad_id, int64_t sos_id, int64_t unk_id,
                             int mask_special, int decoding_constraint, float *top_val,
                             int64_t *top_idx, void *stream) {
    if (!logits || !part_max || !part_sum || !last_word || !top_val || !top_idx) return ISC_E_NULL;
    if (rows <= 0 || V <= 0 || beam <= 0 || beam > 16 || beam > V) return ISC_E_SHAPE;
    const int *gate = isc_stream_gate_(stream);
    if (beam <= 8 && n_tile <= 256 && V >= (beam + 4) * 128)
        hipLaunchKernelGGL(beam_topk8_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits,
                           (long long)ld_logits, part_max, part_sum, n_tile, V, beam, last_word,
                           (long long)pad_id, (long long)sos_id, (long long)unk_id, mask_special,
                           decoding_constraint, top_val, top_idx, gate);
    else
        hipLaunchKernelGGL(beam_topk_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits,
                           (long long)ld_logits, part_max, part_sum, n_tile, V, beam, last_word,
                           (long long)pad_id, (long long)sos_id, (long long)unk_id, mask_special,
                           decoding_constraint, top_val, top_idx, gate);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ beam step tail in one launch
// isc_beam_select: what isc_beam_topk + isc_beam_merge + isc_beam_gather do in three launches, from the tile
// statistics and per-tile sorted candidate lists the few-row classifier (rows.hip) leaves.  One workgroup per image.  A
// lone wave retires an instruction every ~6-8 cycles here (and a launch that returns at once already takes 4.6 us:
// tools/select_lab.py), so the launch is as long as the LONGEST dependent instruction chain of any wave.  The chain is
// merge -> scores -> rank -> new rows; everything that is not on it runs beside it on waves of its own:
//   merge wave r (one per beam row): the row's tile candidates (value, id) [n_tile][8] - a lane owns tiles lane,
//     lane + 64, ... and holds their lists in registers - and a `beam`-round k-way merge over the lists' heads: ONE 32-bit
//     DPP maximum per round, ties by ballots and scalar bit scans (value descending, then tile ascending = word id
//     ascending: tiles are column ranges, lists are sorted), the winner's list moves up one place, round k's winner
//     stays in lane k's registers;  barrier;  parent r's candidates scored in fp64 by lanes 0..beam-1 (captioner.py:
//     378-411, beam_merge_kernel's rules: the parents' bookkeeping sits in lanes 0..7 of every wave, no staging) -> LDS;
//     barrier;  stable descending rank with EIGHT lanes per candidate (each counts an eighth of the others), ranks
//     < beam write the new rows' bookkeeping;  barrier;  new row r's word list;
//   side wave r (one per beam row): row r's normaliser from (pmax, psum) - fold_row_stats, as every decode path - and its
//     word list -> LDS, row r's planes of the recurrent state -> LDS by LDS-DMA (when the state is re-ordered here), all
//     while the merges run; winners known: new row r's planes from its parent's LDS image.
// Word ids, log-probs ((x - max) - log(sum), the expression of beam_topk8_kernel) and tie order are those of the
// three-launch path on the same logits and statistics.
#define ISC_SEL_TILES_PER_LANE 4
#if ROWS_STAMP
#define RSTAMP_KID 5
RSTAMP_SETTER(isc_pw_set_stamp)
#endif
// Wave-wide maximum / sum that never leave the VALU: four DPP steps leave every 16-lane row holding its reduction, the
// four row values then meet through row_bcast15 / row_bcast31 (maximum) or four v_readlane (sum: the association of
// half_sum + the cross-half add, so the normaliser keeps its bits).  The ds_swizzle / ds_bpermute steps these replace are
// LDS round trips (~100+ cycles each) - in a kernel that is ONE dependent chain they were a third of its time.
__device__ __forceinline__ float sel_wave_fmax(float v) {               // -> the wave's maximum, wave-uniform (NaNs are dropped)
    // v_max_f32 with the DPP operand folded in (the compiler keeps v_mov_dpp + a canonicalising v_max + v_max per step);
    // s_nop 1 = the two wait states a DPP read needs behind the VALU write of its source.  row_bcast15 / 31 write rows
    // 1, 3 / 2, 3 only (row_mask), the other rows keep their value.
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float sel_wave_sum(float v) {                // == half_sum(v) + the other half's, bit for bit
    v += isc_dpp<ISC_DPP_XOR1>(v);
    v += isc_dpp<ISC_DPP_XOR2>(v);
    v += isc_dpp<ISC_DPP_HALF_MIRROR>(v);
    v += isc_dpp<ISC_DPP_MIRROR>(v);
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
// The winner's list (list Q of lane wl, both wave-uniform) moves up one place in its lane - static register indices,
// position 0 is always the head.  Branch-free: each list shifts under its own lane mask (empty for the three lists that did
// not win) - four copies under scalar branches cost more in the register copies that rejoin them than the idle selects do.
// A search pops at most `beam` <= 8 entries and a list holds 8, so no list runs dry before the last round; positions past
// the fifth are only ever heads for beam > 5.
#define SEL_SHIFT(Q, LO, HI)                                                                \
    _Pragma("unroll") for (int s_ = LO; s_ < HI; ++s_) {                                    \
        cvv[Q][s_] = me##Q ? cvv[Q][s_ + 1] : cvv[Q][s_];                                   \
        cid[Q][s_] = me##Q ? cid[Q][s_ + 1] : cid[Q][s_];                                   \
    }

__global__ __launch_bounds__(1024) void beam_select_kernel(const isc_beam_select_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sel_smem[];
    __shared__ double cs[64];                          // the image's candidates (<= beam x beam): score,
    __shared__ long long ctk[64];                      //   token
    __shared__ float nrm[ISC_BEAM_MAX][2];             // per row: max, log(sum)
    __shared__ int w_par[ISC_BEAM_MAX], w_car[ISC_BEAM_MAX], w_len[ISC_BEAM_MAX];     // winners by rank
    __shared__ long long w_tok[ISC_BEAM_MAX];
    rows_kernarg_warm<ROWS_KERNARG_LINES(isc_beam_select_args)>();
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int beam = a.beam, T = a.T, base = i * beam, n_tile = a.n_tile;
    const bool merger = wave < beam;                  // (wave-uniform) else: a side wave
    const int wrow = merger ? wave : wave - beam;     // the beam row this wave serves
    RSTAMP(0);
    // ---- the two block-uniform switches first (their round trip runs under everything requested behind them; without
    // live_in the same load reads this launch's own `beam` > 0 out of the argument block - no branch in front of a load)
    typedef const __attribute__((address_space(1))) int *sel_gint_p;      // (global, not flat: counted in order with the rest)
    const sel_gint_p gop = a.live_in ? (sel_gint_p)(unsigned long long)a.live_in
                                     : (sel_gint_p)((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() +
                                                    offsetof(isc_beam_select_args, beam));
    const int done_v = a.done[i];
    const int go_v = *gop;
    const int row = base + wrow;
    float cvv[ISC_SEL_TILES_PER_LANE][8];
    int cid[ISC_SEL_TILES_PER_LANE][8];
    float pm[ISC_SEL_TILES_PER_LANE], ps[ISC_SEL_TILES_PER_LANE];
    long long wv[4];                                  // the row's word list, positions lane, lane + 64, ..
    long long *wl = reinterpret_cast<long long *>(sel_smem);                                              // [row][T]
    float *st_lds = reinterpret_cast<float *>(sel_smem + (((size_t)beam * T * 8 + 15) & ~(size_t)15));   // [plane][row][H]
    const int rows_all = a.n_img * beam;
    // parents' bookkeeping: lane l of EVERY wave holds parent min(l & 7, beam - 1)
    const int pk = (lane & 7) < beam ? (lane & 7) : beam - 1;
    double my_score = 0.0;
    int my_len = 0;
    if (merger) {
        const float *cvp = a.cand_val + (long long)row * n_tile * 8;        // (wave-uniform bases, 32-bit lane offsets)
        const int *cip = a.cand_idx + (long long)row * n_tile * 8;
#pragma unroll
        for (int q = 0; q < ISC_SEL_TILES_PER_LANE; ++q) {
            const unsigned tl = lane + 64 * q, tc = tl < (unsigned)n_tile ? tl : (unsigned)n_tile - 1u;
            const float4 *gv = reinterpret_cast<const float4 *>(cvp + tc * 8u);
            const int4 *gi = reinterpret_cast<const int4 *>(cip + tc * 8u);
            const float4 v0 = gv[0], v1 = gv[1];
            const int4 i0 = gi[0], i1 = gi[1];
            cvv[q][0] = v0.x; cvv[q][1] = v0.y; cvv[q][2] = v0.z; cvv[q][3] = v0.w;
            cvv[q][4] = v1.x; cvv[q][5] = v1.y; cvv[q][6] = v1.z; cvv[q][7] = v1.w;
            cid[q][0] = i0.x; cid[q][1] = i0.y; cid[q][2] = i0.z; cid[q][3] = i0.w;
            cid[q][4] = i1.x; cid[q][5] = i1.y; cid[q][6] = i1.z; cid[q][7] = i1.w;
        }
        my_score = a.score_in[base + pk];
        my_len = a.len_in[base + pk];
    }
    const long long my_last = a.last_in[base + pk];
    if (!merger) {
        const float *pmp = a.part_max + (long long)row * n_tile, *psp = a.part_sum + (long long)row * n_tile;
#pragma unroll
        for (int q = 0; q < ISC_SEL_TILES_PER_LANE; ++q) {
            const unsigned tl = lane + 64 * q, tc = tl < (unsigned)n_tile ? tl : (unsigned)n_tile - 1u;
            pm[q] = pmp[tc];
            ps[q] = psp[tc];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pos = lane + 64 * q;
            wv[q] = a.words_in[(long long)row * T + (pos < T ? pos : T - 1)];
        }
        if (a.state_out) {
            // this row's planes -> LDS image [plane][row][H], on their way while the merges run
            const unsigned lds0 = (unsigned)(size_t)st_lds;
            for (int pl = 0; pl < a.state_planes; ++pl) {
                const float *srow = a.state_in + ((long long)pl * rows_all + row) * a.H;
                const int pr = pl * beam + wrow;
                if ((a.H & 255) == 0) {
                    for (int c0 = 0; c0 < a.H; c0 += 256) {     // one 1 KB LDS-DMA per 256 floats
                        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(pr * a.H + c0) * 4u);
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(srow + c0 + lane * 4) : "memory");
                    }
                } else {                                        // small / odd widths: through registers
                    for (int c = lane * 4; c < a.H; c += 256)
                        *reinterpret_cast<float4 *>(st_lds + pr * a.H + c) = *reinterpret_cast<const float4 *>(srow + c);
                }
            }
        }
    }
    RSTAMP(1);
    if (__builtin_amdgcn_readfirstlane(go_v) == 0) {             // the search has ended (block-uniform)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (no LDS-DMA may land after the workgroup has gone)
        return;
    }
    if (__builtin_amdgcn_readfirstlane(done_v)) {                // frozen image: everything carried over unchanged (block-uniform)
        if (merger) {
            if (wave == 0 && lane < beam) {
                a.src_row[base + lane] = base + lane;
                a.last_out[base + lane] = my_last;
                a.score_out[base + lane] = my_score;
                a.len_out[base + lane] = my_len;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pos = lane + 64 * q;
                if (pos < T) a.words_out[(long long)row * T + pos] = wv[q];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (a wave reads back only what it fetched itself)
            if (a.state_out)
                for (int pl = 0; pl < a.state_planes; ++pl)
                    for (int c = lane * 4; c < a.H; c += 256)
                        *reinterpret_cast<float4 *>(a.state_out + ((long long)pl * rows_all + row) * a.H + c) =
                            *reinterpret_cast<const float4 *>(st_lds + (pl * beam + wrow) * a.H + c);
        }
        return;
    }
    // ---- which parents have ended, where each parent's candidates start, how many there are: scalar work on one ballot
    const int ncand = a.t == 0 ? 1 : beam;
    const unsigned endm = a.t > 0 ? (unsigned)__ballot(my_last == a.eos_id) & ((1u << beam) - 1u) : 0u;
    const int n = ncand * beam - (beam - 1) * __builtin_popcount(endm & ((1u << ncand) - 1u));
    const int nrow = n < beam ? n : beam;             // (t == 0: one live parent still yields `beam` rows)
    float myraw = -INFINITY;              // lane k < beam of a merge wave: the row's k-th best (raw logit, word id)
    int myid = 0;
    if (merger) {
#pragma unroll
        for (int q = 0; q < ISC_SEL_TILES_PER_LANE; ++q)
            if (lane + 64 * q >= n_tile) {
#pragma unroll
                for (int s_ = 0; s_ < 8; ++s_) cvv[q][s_] = -INFINITY;
            }
        // ---- k-way merge over the tiles' list heads: per round ONE 32-bit wave maximum of the lanes' best heads; ties
        // (equal values) go to the smaller word id = the smaller tile index = the first q with a lane at the maximum, then
        // its lowest lane - four ballots and scalar bit scans, no 64-bit keys in the lanes
        unsigned long long own[ISC_SEL_TILES_PER_LANE];         // lanes whose tile q exists (wave-uniform)
#pragma unroll
        for (int q = 0; q < ISC_SEL_TILES_PER_LANE; ++q) own[q] = __ballot(lane + 64 * q < n_tile);
        RSTAMP(2);
        for (int k = 0; k < beam; ++k) {
            const float mx = sel_wave_fmax(fmaxf(fmaxf(cvv[0][0], cvv[1][0]), fmaxf(cvv[2][0], cvv[3][0])));
            const unsigned long long h0 = __ballot(cvv[0][0] == mx) & own[0], h1 = __ballot(cvv[1][0] == mx) & own[1];
            const unsigned long long h2 = __ballot(cvv[2][0] == mx) & own[2], h3 = __ballot(cvv[3][0] == mx) & own[3];
            // the winning list as four lane masks, three of them empty (all empty: a NaN maximum - nothing moves)
            const unsigned long long f0 = h0, f1 = h0 ? 0 : h1, f2 = (h0 | h1) ? 0 : h2, f3 = (h0 | h1 | h2) ? 0 : h3;
            const unsigned long long fa = f0 | f1 | f2 | f3;
            const int wl = fa ? __builtin_ctzll(fa) : 0;        // its lowest lane
            const bool mel = lane == wl;
            const bool me0 = mel & (f0 != 0), me1 = mel & (f1 != 0), me2 = mel & (f2 != 0), me3 = mel & (f3 != 0);
            const int hidv = me0 ? cid[0][0] : me1 ? cid[1][0] : me2 ? cid[2][0] : cid[3][0];
            const int wid = fa ? __builtin_amdgcn_readlane(hidv, wl) : 0;
            SEL_SHIFT(0, 0, 4) SEL_SHIFT(1, 0, 4) SEL_SHIFT(2, 0, 4) SEL_SHIFT(3, 0, 4)
            if (beam > 5) {
                SEL_SHIFT(0, 4, 7) SEL_SHIFT(1, 4, 7) SEL_SHIFT(2, 4, 7) SEL_SHIFT(3, 4, 7)
                cvv[0][7] = me0 ? -INFINITY : cvv[0][7]; cvv[1][7] = me1 ? -INFINITY : cvv[1][7];
                cvv[2][7] = me2 ? -INFINITY : cvv[2][7]; cvv[3][7] = me3 ? -INFINITY : cvv[3][7];
            }
            myraw = lane == k ? (fa ? mx : -INFINITY) : myraw;
            myid = lane == k ? wid : myid;
        }
        RSTAMP(3);
    } else {
        // ---- side wave: the row's normaliser - max over the tile maxima, sum of the rescaled tile sums - and its word
        // list -> LDS
#pragma unroll
        for (int q = 0; q < ISC_SEL_TILES_PER_LANE; ++q)
            if (lane + 64 * q >= n_tile) { pm[q] = -INFINITY; ps[q] = 0.f; }
        const float gmax = sel_wave_fmax(fmaxf(fmaxf(pm[0], pm[1]), fmaxf(pm[2], pm[3])));
        float ssum = 0.f;
#pragma unroll
        for (int q = 0; q < ISC_SEL_TILES_PER_LANE; ++q) ssum += ps[q] * expf(pm[q] - gmax);
        ssum = sel_wave_sum(ssum);
        if (lane == 0) {
            if (!(fabsf(gmax) <= 3.0e38f && ssum <= 3.0e38f)) isc_flag_pw(ISC_STATUS_WORD_STATS);     // NaN fails both
            nrm[wrow][0] = gmax;
            nrm[wrow][1] = logf(ssum);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pos = lane + 64 * q;
            if (pos < T) wl[wrow * T + pos] = wv[q];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's share of the state image has landed
    }
    __syncthreads();                                              // normalisers, word lists, state image
    if (merger) {
        const float myv = (myraw - nrm[wave][0]) - nrm[wave][1];  // (-inf stays -inf)
        if (a.top_val && lane < beam) {
            a.top_val[(long long)row * beam + lane] = myv;
            a.top_idx[(long long)row * beam + lane] = (unsigned)myid < (unsigned)a.V ? myid : 0;
        }
        // ---- parent `wave`'s candidates (captioner.py:378-411; beam_merge_kernel's rules): an ended parent is carried as
        // ONE candidate with its score, a live one expands into its `beam` best words
        if (wave < ncand) {
            const int ended = (endm >> wave) & 1u;
            const int ck = wave * beam - (beam - 1) * __builtin_popcount(endm & ((1u << wave) - 1u));
            const double sc_k = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_score), wave),
                                                 __builtin_amdgcn_readlane(__double2loint(my_score), wave));
            if (lane < (ended ? 1 : beam)) {
                cs[ck + lane] = ended ? sc_k : sc_k + (double)myv;
                ctk[ck + lane] = ended ? a.eos_id : (long long)((unsigned)myid < (unsigned)a.V ? myid : 0);
            }
        }
    }
    RSTAMP(4);
    __syncthreads();                                              // candidates
    RSTAMP(5);
    if (merger) {
        if (wave < ncand) {
            // stable descending rank of candidate c = ck + j: lanes 8 j .. 8 j + 7 each count an eighth of the others
            const int ended = (endm >> wave) & 1u;
            const int ck = wave * beam - (beam - 1) * __builtin_popcount(endm & ((1u << wave) - 1u));
            const int j = lane >> 3, s_ = lane & 7, c = ck + j;
            const bool mine_ok = j < (ended ? 1 : beam);
            const double mine = cs[mine_ok ? c : 0];
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int it = s_ + 8 * u;
                const double o = cs[it];                        // (it < 64: inside the array; counted only below n)
                cnt += (it < n) & ((int)(o > mine) | ((int)(o == mine) & (int)(it < c)));
            }
            cnt += isc_dpp<ISC_DPP_XOR1>(cnt);
            cnt += isc_dpp<ISC_DPP_XOR2>(cnt);
            cnt += isc_dpp<ISC_DPP_HALF_MIRROR>(cnt);
            if (s_ == 0 && mine_ok && cnt < beam) {
                const int len = __builtin_amdgcn_readlane(my_len, wave);
                const int dst = base + cnt;
                const long long tok = ctk[c];
                w_par[cnt] = wave; w_car[cnt] = ended; w_len[cnt] = len; w_tok[cnt] = tok;
                a.score_out[dst] = mine;
                a.last_out[dst] = tok;
                a.src_row[dst] = base + wave;
                a.len_out[dst] = len + (ended ? 0 : 1);
            }
        }
        if (tid == 0) {
            const unsigned all = (1u << ncand) - 1u;
            if ((endm & all) == all) a.done[i] = 1;
            else atomicAdd(&a.live[a.t + 1], 1);
        }
    }
    __syncthreads();                                              // winners
    RSTAMP(6);
    if (wrow >= nrow) return;
    const int par = w_par[wrow];
    if (merger) {
        // ---- new row `wave`'s word list: its parent's, with the new token at the parent's length
        const int car = w_car[wrow], len = w_len[wrow];
        const long long tok = w_tok[wrow];
        for (int pos = lane; pos < T; pos += 64)
            a.words_out[(long long)row * T + pos] = (!car && pos == len) ? tok : wl[par * T + pos];
    } else if (a.state_out) {
        // ---- the recurrent state follows the candidates: new row r continues its parent's
        for (int pl = 0; pl < a.state_planes; ++pl)
            for (int c = lane * 4; c < a.H; c += 256)
                *reinterpret_cast<float4 *>(a.state_out + ((long long)pl * rows_all + row) * a.H + c) =
                    *reinterpret_cast<const float4 *>(st_lds + (pl * beam + par) * a.H + c);
    }
    RSTAMP(7);
}
#undef SEL_SHIFT

extern "C" int isc_beam_select(const isc_beam_select_args *args, void *stream) {
    if (!args) return ISC_E_NULL;
    const isc_beam_select_args &a = *args;
    if (!a.part_max || !a.part_sum || !a.cand_val || !a.cand_idx || !a.score_in || !a.score_out || !a.last_in ||
        !a.last_out || !a.words_in || !a.words_out || !a.len_in || !a.len_out || !a.done || !a.src_row || !a.live)
        return ISC_E_NULL;
    if ((a.top_val == nullptr) != (a.top_idx == nullptr)) return ISC_E_NULL;
    if ((a.state_in == nullptr) != (a.state_out == nullptr)) return ISC_E_NULL;
    if (a.state_out && (a.state_planes <= 0 || a.H <= 0 || (a.H & 3) || a.state_in == a.state_out)) return ISC_E_SHAPE;
    if (a.state_out && (!isc_aligned16(a.state_in) || !isc_aligned16(a.state_out))) return ISC_E_ALIGN;
    if (a.n_img <= 0 || a.beam <= 0 || a.beam > ISC_BEAM_MAX || a.T <= 0 || a.t < 0 || a.t >= a.T || a.V <= 0)
        return ISC_E_SHAPE;
    if (a.n_tile < 1 || a.n_tile > 64 * ISC_SEL_TILES_PER_LANE) return ISC_E_SHAPE;
    if (!isc_aligned16(a.cand_val) || !isc_aligned16(a.cand_idx)) return ISC_E_ALIGN;
    size_t lds = (((size_t)a.beam * a.T * 8 + 15) & ~(size_t)15);                // the parents' word lists
    if (a.state_out) lds += (size_t)a.state_planes * a.beam * a.H * 4 + 1024;    // + the image's rows of the state (whole 1 KB pieces)
    if (lds > 150000) return ISC_E_SHAPE;
    if (a.T > 256) return ISC_E_SHAPE;
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&beam_select_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
        if (e != hipSuccess) return (int)e;
        attr_set.store(true);
    }
    // one merge wave + one side wave per beam row
    hipLaunchKernelGGL(beam_select_kernel, dim3(a.n_img), dim3(128 * a.beam), lds, (hipStream_t)stream, a);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ XECriterion
// Single workgroup, fixed reduction order => bitwise reproducible loss.
__global__ __launch_bounds__(256) void xe_loss_kernel(const float *logp, const int64_t *target,
                                                      const int *lengths, int B, int T, int V,
                                                      float *out2) {
    __shared__ float ss[4], sn[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f, n = 0.f;
    for (int i = tid; i < B * T; i += 256) {
        const int b = i / T, t = i % T;
        if (t < lengths[b]) {
            s -= logp[(long long)i * V + target[i]];
            n += 1.f;
        }
    }
    s = wave_sum(s);
    n = wave_sum(n);
    if (lane == 0) { ss[wave] = s; sn[wave] = n; }
    __syncthreads();
    if (tid == 0) {
        out2[0] = (ss[0] + ss[1]) + (ss[2] + ss[3]);
        out2[1] = (sn[0] + sn[1]) + (sn[2] + sn[3]);
    }
}

extern "C" int isc_xe_loss_fwd(const float *logp, const int64_t *target, const int32_t *lengths,
                               int B, int T, int V, float *out2, void *stream) {
    if (!logp || !target || !lengths || !out2) return ISC_E_NULL;
    if (B <= 0 || T <= 0 || V <= 0) return ISC_E_SHAPE;
    hipLaunchKernelGGL(xe_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp, target, lengths,
                       B, T, V, out2);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// ------------------------------------------------------------------ criteria on RAW logits (training: no [B,T,V] log-probs)
// The criteria of this path read ONE column per (caption, step) row - XECriterion the target (captioner.py:427-440), the
// REINFORCE term the drawn token (captioner.py:336).  log p(id) = (x[id] - max) - log(sum exp) comes straight from the raw
// logits and the classifier's tile statistics: the [B,T,V] log-prob tensor (F.log_softmax at captioner.py:183 over every
// step) is then never written or read in a training iteration (0.56 ms of a 15.6 ms iteration at B = 1024).  Same
// expression as logsoftmax_apply_steps_kernel: the values are the bits that tensor would have held.
// One wave per row; rows in [B,T] order (ids / out), logits row (b,t) at b*ld_b + t*ld_t, statistics row t*step_rows + b.
__global__ __launch_bounds__(256) void gather_logp_raw_kernel(const float *raw, long long ld_b, long long ld_t, int B, int T,
                                                              const float *pmax, const float *psum, int n_tile,
                                                              int step_rows, const int64_t *ids, const float *live,
                                                              float *out) {
    const int lane = threadIdx.x & 63;
    const int mi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (mi >= B * T) return;
    const int b = mi / T, t = mi - b * T;
    const long long ms = (long long)t * step_rows + b;
    float gmax, S;
    int gi;
    fold_row_stats(pmax + ms * n_tile, psum + ms * n_tile, nullptr, n_tile, lane, gmax, gi, S);
    if (lane == 0) {
        const float x = raw[(long long)b * ld_b + (long long)t * ld_t + ids[mi]];
        const float lp = (x - gmax) - logf(S);
        out[mi] = live ? lp * live[t] : lp;
    }
}

extern "C" int isc_gather_logp_raw(const float *raw, int64_t ld_b, int64_t ld_t, int B, int T, int V, const float *part_max,
                                   const float *part_sum, int step_rows, const int64_t *ids, const float *live, float *out,
                                   void *stream) {
    if (!raw || !part_max || !part_sum || !ids || !out) return ISC_E_NULL;
    if (B <= 0 || T <= 0 || V <= 0 || (long long)B * T > 2147483647LL) return ISC_E_SHAPE;
    if (step_rows == 0) step_rows = B;
    if (step_rows < B) return ISC_E_SHAPE;
    hipLaunchKernelGGL(gather_logp_raw_kernel, dim3((unsigned)((B * T + 3) / 4)), dim3(256), 0, (hipStream_t)stream, raw,
                       (long long)ld_b, (long long)ld_t, B, T, part_max, part_sum, (V + 127) / 128, step_rows, ids, live, out);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// XECriterion on per-token log-probs tlp [B,T] = log p(target): out2 = { -sum_{t < len_b} tlp, count }.  The summation
// order of xe_loss_kernel (single workgroup, thread-strided) - the same loss bits.
__global__ __launch_bounds__(256) void xe_loss_tokens_kernel(const float *tlp, const int *lengths, int B, int T, float *out2) {
    __shared__ float ss[4], sn[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f, n = 0.f;
    for (int i = tid; i < B * T; i += 256) {
        const int b = i / T, t = i % T;
        if (t < lengths[b]) {
            s -= tlp[i];
            n += 1.f;
        }
    }
    s = wave_sum(s);
    n = wave_sum(n);
    if (lane == 0) { ss[wave] = s; sn[wave] = n; }
    __syncthreads();
    if (tid == 0) {
        out2[0] = (ss[0] + ss[1]) + (ss[2] + ss[3]);
        out2[1] = (sn[0] + sn[1]) + (sn[2] + sn[3]);
    }
}

extern "C" int isc_xe_loss_tokens_fwd(const float *tlp, const int32_t *lengths, int B, int T, float *out2, void *stream) {
    if (!tlp || !lengths || !out2) return ISC_E_NULL;
    if (B <= 0 || T <= 0) return ISC_E_SHAPE;
    hipLaunchKernelGGL(xe_loss_tokens_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tlp, lengths, B, T, out2);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

extern "C" int isc_abi_version(void) { return 1; }
extern "C" const char *isc_target_arch(void) { return "gfx950"; }
