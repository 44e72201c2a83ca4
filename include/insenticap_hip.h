/*
 * insenticap_hip.h - C ABI of libinsenticap_hip.so (gfx950 / MI355X).
 *
 * The upstream project (ezeli/InSentiCap_model) has no FFI or plugin layer: its
 * caption decoder is pure Python on stock torch ops. The drop-in boundary is the
 * Python class surface of `Captioner` / `Detector`; this header
 * is the native boundary underneath it: one entry point per
 * fused unit of the reference's decode step. Each declaration cites the reference
 * lines it replaces.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless the
 *     name ends in _host; tensors are row-major fp32, ids are int64 ("LongTensor");
 *   - no allocation, no ownership transfer, no device synchronisation: work is enqueued on
 *     `stream` (a hipStream_t passed as void*);
 *   - no data-carrying global state: entry points may be called from several host threads on
 *     distinct streams at once (tests/test_gpu_threads.py).  What the library keeps is (a) two
 *     process-wide tuning words, isc_set_tile_override / isc_set_h3_mode / isc_set_gemv_rows (atomic; they pick a
 *     kernel, never change a result beyond fp32 summation order), (b) launch counters (atomic),
 *     and (c) the split-f16 weights scopes, which are keyed by stream (isc_h3_weights_begin);
 *   - return value: 0 ok; <0 bad argument / unsupported shape (ISC_E_*);
 *     >0 a hipError_t from the launch. Never throws, never exits.
 *   - leading dimensions are in elements; every row start must be 16-byte aligned
 *     and every contraction length a multiple of 32.
 */
#ifndef INSENTICAP_HIP_H
#define INSENTICAP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISC_OK 0
#define ISC_E_NULL (-1)      /* required pointer is null */
#define ISC_E_SHAPE (-2)     /* unsupported size (K %32, dims %32, too many segments ...) */
#define ISC_E_ALIGN (-3)     /* pointer or leading dimension not 16-byte aligned */
#define ISC_E_WORKSPACE (-4) /* caller-provided workspace too small */
#define ISC_E_STATE (-5)     /* nothing to resume (isc_h3_weights_resume) */

#define ISC_MAX_SEG 4

/* Library / build identification ("gfx950", ABI version). */
int isc_abi_version(void);
const char *isc_target_arch(void);
/* Tuning / test hook: force the GEMM tile shape of isc_linear_fwd / isc_lstm_fwd / isc_vocab_fwd /
 * isc_gemm_bwd (0 = 128x128, 1 = 64x128, 2 = 32x128, 3 = 256x128 LDS-DMA ring, 4 = 128x128 LDS-DMA [3 and 4: NT
 * layout only; other layouts fall back to 0]); -1 (default) = the cost model picks.  Results are identical up to fp32
 * summation order for every shape; process-wide, not meant to be flipped while launches are in flight
 * on other threads.  Returns the previous value. */
int isc_set_tile_override(int tile);
/* Split-f16 GEMM path of the forward entry points (isc_linear_fwd / isc_lstm_fwd / isc_vocab_fwd): fp32 operands
 * are split into two f16 planes each (x = hi + lo * 2^-11) inside the caller's workspace and contracted with three
 * f16 MFMAs per k-step into fp32 accumulators - fp32 in, fp32 out, error against an fp64 contraction no larger than
 * an fp32 FMA chain's (tests/test_gpu_h3.py), at 2-2.5x the fp32 MFMA rate.  Operand domain |x| < 65504.
 * mode 0 = off (fp32 MFMA tiles only); 1 = auto (default): on a stream that holds a weights scope a launch takes the
 * skinny split-f16 kernel (one launch per GEMM, no split-K slabs; 32 x 32 or 64 x 64 tiles, whichever needs fewer
 * rounds of the chip) while that is at most ~4 rounds of small tiles, else the large split-f16 kernels; without a
 * scope, launches of >= 160 128x128 tiles take the large kernels and the rest the fp32 tiles; 2 = the large kernels
 * whenever shapes and workspace allow; 3 / 4 = the skinny kernel with 32 x 32 / 64 x 64 tiles whenever shapes allow
 * (test hooks: without a scope the weight planes go to the workspace).  The planes of a launch must fit the
 * workspace.  A tile override (>= 0) also disables the path.  Returns the previous mode. */
int isc_set_h3_mode(int mode);
/* Number of launches that went out on the split-f16 path so far (process-wide; measurement / test hook);
 * isc_h3x_launches: those of them that took the 256x128 eight-wave tile (launches of >= 224 such tiles). */
long long isc_h3_launches(void);
/* Few rows (beam rows of one image, a handful of captions): isc_linear_fwd / isc_lstm_fwd / isc_vocab_fwd launches
 * whose problems all have M <= rows (default and maximum 8) and K <= 4096 go out as ONE fused matrix-vector kernel
 * each (gemv_rows_kernel: weights read as fp32 with the whole launch's traffic in flight, exact fp32 FMA, activations
 * staged in LDS, LSTM cell / vocabulary statistics in the epilogue - north_star's "fused LSTM gate GEMV +
 * sigmoid/tanh").  Modes 0 and 1 of isc_set_h3_mode only (2-4 force the split-f16 kernels).  rows = 0 switches the
 * path off; returns the previous value.  isc_gemv_launches: launches that went out this way (test hook). */
int isc_set_gemv_rows(int rows);
long long isc_gemv_launches(void);
long long isc_h3x_launches(void);
/* Launches that took the skinny split-f16 kernel (few rows: one launch per GEMM instead of split-K + reduce). */
long long isc_h3s_launches(void);
/* Weights scope of the split-f16 path.  Between _begin and _end the caller guarantees that no weight matrix passed
 * to the forward entry points changes (a roll-out's decode loop): each weight operand is then split into its planes
 * once, into `buf` (device memory, 256-byte aligned; 64 MB holds the decoder's matrices), and later launches with
 * the same weight segments reuse them instead of re-splitting per step.  A scope belongs to ONE stream: only
 * launches on `stream` see it, and they must be issued in one order (as a stream's launches are anyway); up to 64
 * streams may hold a scope - active or suspended - at once (ISC_E_WORKSPACE beyond that: run without; a suspended scope
 * of another stream is never taken over).  Without a scope, or when `buf` is
 * full, every launch splits its weights into the workspace.  _begin discards the stream's earlier entries; _end
 * closes its scope (`buf` may then be reused). */
int isc_h3_weights_begin(void *buf, long long bytes, void *stream);
int isc_h3_weights_end(void *stream);
/* _suspend leaves the stream's scope (no launch sees it any more) but keeps its planes; _resume(buf, stream) re-opens
 * it as it was - the caller vouches that no weight has changed since (Captioner keys it on the parameters' version
 * counters, so consecutive eval-mode calls split the weights once, not once per call) - or returns ISC_E_STATE when
 * the slot has meanwhile been recycled (then: _begin).  Between the two, launches on the stream run without a scope. */
int isc_h3_weights_suspend(void *stream);
int isc_h3_weights_resume(void *buf, void *stream);
/* Re-split every weight the stream's scope (active or suspended) holds planes of from the weights' current values - for an
 * optimiser that has just updated them in place on the same stream: the scope can then be resumed instead of rebuilt entry
 * by entry (few batched split launches instead of one per weight operand).  ISC_E_STATE: the stream has no scope. */
int isc_h3_weights_refresh(void *stream);

/* One K-segment of a contraction: acc += A[M,K] * W[N,K]^T.  Replaces the
 * torch.cat([...],1) + nn.Linear / nn.LSTMCell pattern of captioner.py:174-175,180-181. */
typedef struct {
    const float *A; /* [M, K] activations, leading dim lda */
    const float *W; /* [N, K] weight slice (row-major, K contiguous), leading dim ldw */
    int32_t lda, ldw, K, _pad;
    /* Optional (forward entry points, split-f16 path): the f16 planes of A as a producer wrote them
     * (isc_lstm_problem.h_hi / h_lo ...): hi = f16(x), lo = f16((x - hi) * 2048), in ONE buffer of 2*M*K halfs with the
     * two planes interleaved per 32-wide k-block (so that a chunk consumes whole 128-byte lines):
     *     hi(m, k) = buf[m*2K + (k/32)*64 + k%32],   lo(m, k) = the same + 32;   A_hi = buf, A_lo = buf + 32 halfs.
     * Every plane pointer pair of this header uses that layout for its own [rows, width] tensor (K % 32 == 0).
     * When present and the launch takes the split-f16 path, A is read from the planes instead of being split again;
     * `A` must still be valid (the fp32 tiles read it).  Null = split on the fly. */
    const void *A_hi, *A_lo;
} isc_seg;

/* y = act(sum_seg A_s W_s^T + bias0 + bias1) [* keep_mask * mask_scale]
 * Replaces nn.Linear(+ReLU)(+Dropout) of captioner.py:137-154 (fc_embed, cpt2fc, att_embed,
 * att2att, senti2att) and the per-step projections h2att / h2word / label2word
 * (captioner.py:26,51-52) and cont2att+senti2att+h2att of the gate (captioner.py:107-110).
 * Up to 3 independent problems are grouped in one launch (n_prob). */
typedef struct {
    isc_seg seg[ISC_MAX_SEG];
    int32_t nseg, M, N, relu;
    const float *bias0, *bias1, *bias2; /* [N] or null (the gate sums three nn.Linear outputs) */
    const uint8_t *keep_mask;   /* [M,N] contiguous 0/1 dropout keep-mask or null */
    float mask_scale;           /* 1/(1-p) */
    int32_t ldc;
    float *C;                   /* [M,N] */
    float *C_pre;               /* optional: activation before the mask (captioner.fc_feats attr) */
    int32_t accumulate;         /* C += result (gradient accumulation) */
    int32_t _pad;
    /* Optional split-K workspace (of the FIRST problem of a launch): when a launch has too few tiles to
     * occupy the chip (small M), the contraction is split over ksplit workgroups per tile, partial tiles go
     * to [ksplit, M, N] slabs in this buffer and a second kernel reduces them in fixed order and applies
     * the epilogue (bitwise reproducible). NULL disables it. */
    float *splitk_ws;
    int64_t splitk_ws_floats;
} isc_linear_problem;

int isc_linear_fwd(const isc_linear_problem *probs_host, int n_prob, void *stream);

/* Backward-pass contractions of the same layers (what autograd derives for nn.Linear /
 * nn.LSTMCell / the classifier in train_xe.py:190, decoder.py:165), on the same MFMA kernel:
 *   ISC_LAYOUT_NN  C[M,N] (+)= sum_s A_s[M,K_s] * W_s[K_s,N]      dX = dY * W
 *                  (W_s row-major [K_s,N]: an nn.Linear weight [out=K_s, in=N] as stored)
 *   ISC_LAYOUT_TN  C[M,N] (+)= sum_s A_s[K_s,M]^T * W_s[K_s,N]    dW = dY^T * X
 *                  (contraction over the K_s batch rows; A_s = dY [rows, M], W_s = X [rows, N])
 * bias/relu/mask fields are ignored except bias0..2 (added), M/N/ldc %4 == 0. */
#define ISC_LAYOUT_NN 1
#define ISC_LAYOUT_TN 2
int isc_gemm_bwd(const isc_linear_problem *probs_host, int n_prob, int layout, void *stream);

/* Fused LSTMCell: gates = sum_seg A_s W_s^T + b_ih + b_hh (row blocks i,f,g,o of 4H),
 * c' = sig(f) c + sig(i) tanh(g), h' = sig(o) tanh(c').  Replaces nn.LSTMCell at
 * captioner.py:175 (att_lstm) and :181 (lang_lstm).  h_out/c_out may not alias inputs.
 * gates_out (optional, [M,4H], activated i,f,g,o) is saved for the backward pass.
 * h_keep_mask (optional) additionally writes hdrop_out = h' * mask * scale
 * (the nn.Dropout on h_lang, captioner.py:182). */
typedef struct {
    isc_seg seg[ISC_MAX_SEG];
    int32_t nseg, M, H, _pad;
    const float *b_ih, *b_hh; /* [4H] */
    const float *c_prev;      /* [M,H] contiguous */
    float *h_out, *c_out;     /* [M,H] contiguous */
    float *gates_out;         /* [M,4H] or null */
    const uint8_t *h_keep_mask; /* [M,H] or null */
    float mask_scale;
    float *hdrop_out;         /* [M,H], required iff h_keep_mask */
    void *h_hi, *h_lo;        /* optional: f16 planes of h_out ([M,H] each) for consumers' isc_seg.A_hi / A_lo */
    /* Hoisted step-invariant inputs (optional): gates += pre[m,:] + tab[tab_ids[m*tab_ids_stride],:].
     * `pre` [M,4H] holds fc W_fc^T + label W_x^T + b_ih + b_hh computed once per call (b_ih/b_hh may
     * then be null); `tab` [V,4H] = relu(Emb) W_x^T replaces the word-embedding K-segment when the
     * weights are frozen (inference). */
    const float *pre;
    const float *tab;
    const int64_t *tab_ids;
    int64_t tab_ids_stride;
    float *splitk_ws;           /* optional split-K workspace, see isc_linear_problem */
    int64_t splitk_ws_floats;
} isc_lstm_problem;

int isc_lstm_fwd(const isc_lstm_problem *prob_host, void *stream);

/* Vocabulary projection with fused log-softmax statistics (captioner.py:183):
 * logits = h W^T + b.  Per 128-column tile the kernel emits (max, argmax, sum exp(x-max));
 * the [M,V] logits are only stored when `logits` is non-null (XE / beam / sampling).
 * n_tile = ceil(V/128); part_* are [M, n_tile].  With a split-K workspace (optional, see
 * isc_linear_problem) launches of few rows contract K in parallel slices and a reduce kernel forms the
 * same outputs (V % 4 == 0 required for that route; otherwise the single-pass kernel runs).
 * h_hi / h_lo (optional): f16 planes of h ([M,K] contiguous, see isc_seg.A_hi). */
int isc_vocab_fwd(const float *h, int ldh, const float *W, int ldw, const float *bias,
                  int M, int V, int K, float *logits, int64_t ld_logits,
                  float *part_max, float *part_sum, int32_t *part_idx,
                  const void *h_hi, const void *h_lo,
                  float *splitk_ws, int64_t splitk_ws_floats, void *stream);

/* logp[m, :] = logits[m, :] - logsumexp(row) in place, using the tile statistics.
 * (F.log_softmax of captioner.py:183 when the full [B,V] row is an API output.) */
int isc_logsoftmax_apply(float *logits, int64_t ld_logits, int M, int V,
                         const float *part_max, const float *part_sum, float *lse_out,
                         void *stream);
/* The same for all T steps of a teacher-forced unroll in ONE launch: logits [B,T,V] with row (b,t) at b*ld_b + t*ld_t,
 * tile statistics stacked per step [T,B,n_tile] (each step's isc_vocab_fwd / isc_step_fwd wrote its slice).
 * src (may be NULL = in place): raw logits stacked per step, contiguous [T*B, V] - what ONE isc_vocab_fwd over all
 * steps' h_lang [T*B, H] writes (torch.stack(outputs, dim=1) of captioner.py:232 happens in this kernel's store). */
int isc_logsoftmax_apply_steps(float *logits, int64_t ld_b, int64_t ld_t, int B, int T, int V,
                               const float *part_max, const float *part_sum, const float *src, int step_rows,
                               void *stream);
/* step_rows (0 = B): rows per step of the statistics / src stacks - row (t, b) sits at t*step_rows + b.  The stacks of a
 * merged unroll (isc_step_plan.pair_rows_c) hold both branches' rows per step; each branch's [B,T,V] log-probs come from
 * its own call with part_max / part_sum / src pointing at the branch's first row. */

/* Additive attention scan (ContentAttention captioner.py:23-35 / SentiAttention :50-62):
 *   e_r = w . tanh(P[b,r,:] + q[b,:] (+ q2[b,:])) + *w_bias ; alpha = softmax_r(e) ;
 *   out[b,:] = sum_r alpha_r V[b,r,:]
 * P: [B,R,A] projected features, V: [B,R,D] features, q: [B,A] (h projection incl. bias),
 * q2: optional [B,A] (label2word term), alpha_out: [B, alpha_ld] slice of the per-call
 * weights tensor.  Two independent scans (content + sentiment words) can be issued in
 * one launch. */
typedef struct {
    const float *P, *V, *q, *q2, *w;
    const float *w_bias; /* device pointer to the scalar bias of the alpha layer */
    int32_t R, A, D;
    int32_t rows;     /* rows of THIS problem; 0 = the launch's B.  The two problems of a launch may differ: the merged
                       * step of two sibling unrolls (isc_step_plan.pair_rows_c) scans the regions for its first rows and
                       * the sentiment words for the rest in ONE launch. */
    float *out;       /* [B,D] */
    float *alpha_out; /* [B,*] row stride alpha_ld, R values written per row */
    int64_t alpha_ld;
    void *out_hi, *out_lo; /* optional: f16 planes of `out` ([B,D] each, see isc_seg.A_hi) */
    /* Optional gather mode: region r of row b is row row_ids[b*row_ids_ld + r] of P / V, which are then tables
     * [n_rows, A] / [n_rows, D] shared by all rows.  The sentiment words of captioner.py:307-312 in eval mode:
     * word_embed(id) and senti2att(word_embed(id)) depend on the id alone, so two vocabulary-sized tables
     * (cache-resident) replace the per-caption [B,M,.] copies that were streamed from HBM every step. */
    const int64_t *row_ids;
    int64_t row_ids_ld;
} isc_scan_problem;

int isc_attn_scan_fwd(const isc_scan_problem *probs_host, int n_prob, int B, void *stream);

/* Both scans, the gate sum and the gate mix of one decode step (captioner.py:96-118) in ONE launch (few rows, inference):
 * G[i] holds the rows of scan i's V carried through the gate's projection - G[0][b,r,:] = cont2att.weight V_c[b,r,:],
 * G[1] = senti2att.weight V_s rows (a table indexed by scan[1].row_ids in gather mode), no bias - so that
 *   z = zh + (b_gc + sum_r alpha_r G[0][b,r,:]) + (b_gs + sum_m alpha'_m G[1][.,:]) = h2att(h) + cont2att(v) + senti2att(s),
 *   beta = sigmoid(w_gate . tanh(z) + *b_gate),  f = beta v + (1 - beta) s
 * come out of the scan's own attention weights: the step's gate GEMM and gate-mix launches disappear, for R x A more
 * floats streamed per row (why the host uses it for few rows only).  scan[0] = content, scan[1] = sentiment words; their
 * .out / planes are optional here (v and s are only needed through f); A == D <= 1024 for both; zh [B,A] is the h-term
 * incl. its bias.  alpha outputs as in isc_attn_scan_fwd. */
typedef struct {
    isc_scan_problem scan[2];
    const float *G[2];
    const float *zh;
    const float *b_gc, *b_gs;     /* [A] biases of cont2att / senti2att */
    const float *w_gate;          /* [A] attention.att_alpha.weight */
    const float *b_gate;          /* device scalar or NULL */
    float *f;                     /* [B,D] */
    void *f_hi, *f_lo;            /* optional planes of f */
    float *beta;                  /* optional, row stride beta_ld */
    int64_t beta_ld;
} isc_scan_gate_args;
int isc_attn_scan_gate_fwd(const isc_scan_gate_args *args_host, int B, void *stream);

/* Gate fusion (captioner.py:111-117): beta = sigmoid(w . tanh(z[b,:]) + *w_bias);
 * out = beta*v + (1-beta)*s.  z = cont2att(v)+senti2att(s)+h2att(h) from isc_linear_fwd. */
int isc_gate_mix_fwd(const float *z, const float *w, const float *w_bias, const float *v,
                     const float *s, int B, int A, int D, float *out, float *beta_out,
                     int64_t beta_ld, void *out_hi, void *out_lo, void *stream);   /* out_hi/lo: optional planes of out */

/* xt[b,:] = relu(Emb[ids[b]]) (+ add[b,:])   (captioner.py:170-172), ids int64. */
int isc_embed_relu_fwd(const float *emb, int V, int W, const int64_t *ids, int64_t ids_stride,
                       const float *add, int B, float *out, void *stream);

/* out[b,:] = mean_c relu(Emb[ids[b,c]])   (captioner.py:201-202) */
int isc_embed_relu_mean_fwd(const float *emb, int V, int W, const int64_t *ids, int C, int B,
                            float *out, void *stream);

/* out[b,m,:] = relu(Emb[m==0 ? pad_id : ids[b,m-1]]) [* mask*scale]  (captioner.py:307-311) */
int isc_embed_senti_words_fwd(const float *emb, int V, int W, const int64_t *ids, int n_words,
                              int64_t pad_id, int B, const uint8_t *keep_mask, float mask_scale,
                              float *out, void *stream);

/* Device-side state of one roll-out (forward_rl, captioner.py:317-344): no host sync per
 * step; `alive` is a [T+1] int32 counter array (alive[t] = #unfinished rows before step t,
 * alive[0] preset to B) that reproduces the reference's early `break`. */
typedef struct {
    int32_t B, V, T, t;          /* t = current step */
    int32_t n_tile, W;           /* vocab tiles, word-emb dim */
    const float *part_max, *part_sum;
    const int32_t *part_idx;
    const float *logits;         /* [B,V] row stride ld_logits; needed iff forced != null or sample_u != null */
    int64_t ld_logits;
    const int64_t *forced;       /* optional [B,T] tokens to replay instead of argmax */
    const float *sample_u;       /* optional [B,T] uniforms for inverse-CDF sampling */
    int64_t eos_id;
    int64_t *seq;                /* [B,T] */
    float *seq_logprobs;         /* [B,T] */
    float *seq_masks;            /* [B,T] */
    int32_t *unfinished;         /* [B] */
    int32_t *alive;              /* [T+1] */
    int64_t *raw_tokens;         /* optional [B,T]: token before the `* unfinished` masking */
    const float *emb;            /* word embedding [V,W] */
    const float *xt_add;         /* optional [B,W] sentiment-label embedding */
    float *xt_next;              /* [B,W] input of step t+1 */
} isc_rollout_step;

int isc_rollout_finalize(const isc_rollout_step *s_host, void *stream);
long long isc_rollout_finalize_launches(void);   /* launches so far (tests: isc_rows_ext.fin_prev saves them) */

/* Scheduled sampling of the teacher-forced unrolls (captioner.py:219-228): out_ids[b] = u_select[b] < ss_prob
 * ? a draw from exp(logp[b,:]) (inverse CDF with uniform u_draw[b], vocabulary order) : base_ids[b*stride].
 * logp = the previous step's normalised output, part_* = that step's tile statistics from isc_vocab_fwd. */
int isc_sched_sample(const float *logp, int64_t ld, int M, int V, const float *part_max,
                     const float *part_sum, const int32_t *part_idx, const float *u_select,
                     const float *u_draw, float ss_prob, const int64_t *base_ids, int64_t base_stride,
                     int64_t *out_ids, void *stream);
/* The same draw from the row's RAW logits (the statistics describe them: exp(x - max) against the tile masses, as
 * isc_rollout_finalize draws): an unroll with scheduled sampling then normalises its logits once, after its last step
 * (isc_logsoftmax_apply_steps), instead of once per step. */
int isc_sched_sample_raw(const float *logits, int64_t ld, int M, int V, const float *part_max,
                     const float *part_sum, const int32_t *part_idx, const float *u_select,
                     const float *u_draw, float ss_prob, const int64_t *base_ids, int64_t base_stride,
                     int64_t *out_ids, void *stream);

/* Beam step (sample(), captioner.py:390-409), batched over images: for every live beam row
 * apply the -inf masks (PAD,SOS,UNK, last word), take its top-`beam` (value, id) pairs
 * from logp = logits - lse (descending, ties to the smaller word id).  The candidate merge with the reference's
 * stable ordering and fp64 score sums is isc_beam_merge below (or the host mirror's Python / numpy forms). */
int isc_beam_topk(const float *logits, int64_t ld_logits, const float *part_max,
                  const float *part_sum, int n_tile, int rows, int V, int beam,
                  const int64_t *last_word, int64_t pad_id, int64_t sos_id, int64_t unk_id,
                  int mask_special, int decoding_constraint, float *top_val, int64_t *top_idx,
                  void *stream);

/* Candidate merge of a batched beam search on the device (captioner.py:378-411), one step: from the step's top-`beam`
 * (value, id) pairs per live row, form every image's candidates in the reference's insertion order (an ended
 * candidate carries itself, a live one contributes its `beam` children), add scores in fp64 (the reference sums
 * Python floats) and keep the first `beam` of a STABLE descending sort.  State (scores, last words, word lists,
 * lengths) is double-buffered by the caller; `gather` receives, per new row, its source row inside
 * [next-state rows ; current-state rows]; `done[i]` latches when every candidate of image i has ended (the image is
 * then frozen); live[t+1] counts the images still searching.  beam <= 8.  No host read is needed between steps. */
typedef struct {
    int32_t n_img, beam, T, t;
    int64_t eos_id;
    const float *top_val;     /* [n_img*beam, beam] log-probs of isc_beam_topk */
    const int64_t *top_idx;   /* [n_img*beam, beam] */
    const double *score_in;   /* [n_img*beam] */
    double *score_out;
    const int64_t *last_in;   /* [n_img*beam] last word of every candidate = the token fed to the next step */
    int64_t *last_out;
    const int64_t *words_in;  /* [n_img*beam, T] */
    int64_t *words_out;
    const int32_t *len_in;    /* [n_img*beam] */
    int32_t *len_out;
    int32_t *done;            /* [n_img] */
    int64_t *gather;          /* [n_img*beam] */
    int32_t *live;            /* [T+1] */
} isc_beam_merge_args;

int isc_beam_merge(const isc_beam_merge_args *args_host, void *stream);
/* State re-ordering of a beam step: out[p, r, :] = (gather[r] < rows ? state_next : state_cur)[p, gather[r] % rows, :]
 * for the `planes` [rows, H] planes of the recurrent state (h|c x layer), gather as written by isc_beam_merge. */
int isc_beam_gather(const float *state_next, const float *state_cur, const int64_t *gather, float *out,
                    int planes, int rows, int H, void *stream);

/* Up to ISC_COPY_MULTI_MAX device-to-device copies in ONE launch (dst[i] <- src[i], bytes[i] each; non-overlapping): the
 * input copies in front of a graph replay (features, word ids, labels into the graph's static buffers). */
#define ISC_COPY_MULTI_MAX 8
int isc_copy_multi(void *const *dst, const void *const *src, const int64_t *bytes, int n, void *stream);

/* Masked NLL (XECriterion, captioner.py:427-440): returns sum and token count in out[0..1].
 * logp [B,T,V] contiguous, target [B,T] int64, lengths [B] int32. */
int isc_xe_loss_fwd(const float *logp, const int64_t *target, const int32_t *lengths, int B,
                    int T, int V, float *out2, void *stream);

/* ------------------------------------------------------------------ whole decode step */

/* One forward_step (captioner.py:168-186) as ONE host call: enqueues the att-LSTM, the attention
 * projections (+ the h-term of the gate), the attention scan(s), the gate contraction + mix, the
 * lang-LSTM, the vocabulary projection and (optionally) the in-place log-softmax on `stream`.
 * It exists to take ~10 descriptor-building FFI calls per step off the host: at B <= 128 the step
 * is launch-bound.  Every pointer is a device pointer; nn.Linear weights are contiguous [out,in].
 * Branch selection like captioner.py:96-118: att_e == NULL -> sentiment words only (seq2seq),
 * words_e == NULL -> regions only (xe), both -> both + gate. */
typedef struct {
    int32_t rows, H, E, A, W, V, R, Mw;
    /* weights */
    const float *Wih1, *Whh1;           /* att_lstm.weight_ih [4H, H+E+W] (ld = H+E+W), weight_hh [4H,H] */
    const float *Wih2, *Whh2;           /* lang_lstm.weight_ih [4H, E+H], weight_hh */
    const float *b_ih2, *b_hh2;
    const float *W_h2att, *b_h2att, *w_alpha_c, *b_alpha_c;      /* attention.cont_att.* */
    const float *W_h2word, *b_h2word, *w_alpha_s, *b_alpha_s;    /* attention.senti_att.* */
    const float *W_gh, *b_gh, *W_gc, *b_gc, *W_gs, *b_gs, *w_gate, *b_gate; /* attention.{h2att,cont2att,senti2att,att_alpha} */
    const float *W_cls, *b_cls;
    /* step-invariant activations */
    const float *pre1;                  /* [rows,4H] hoisted fc/label/bias term of the att-LSTM */
    const float *tab;                   /* [V,4H] relu(Emb) W_x^T or NULL */
    const float *att_p, *att_e;         /* [rows,R,A], [rows,R,E] or NULL */
    const float *words_p, *words_e;     /* [rows,Mw,A], [rows,Mw,W] or NULL */
    const float *label_w;               /* [rows,A] or NULL */
    /* per step */
    const float *xt;                    /* [rows,W] relu(Emb[token]) (ignored when tab) */
    const int64_t *tok;                 /* token ids (row stride tok_stride), required iff tab */
    int64_t tok_stride;
    const float *h1_prev, *c1_prev, *h2_prev, *c2_prev;   /* [rows,H] */
    float *h1, *c1, *h2, *c2;
    float *g1, *g2;                     /* optional activated gates [rows,4H] (training) */
    float *qa, *v, *qw, *s, *z, *f;     /* [rows,A] / [rows,E] workspaces (branch-dependent) */
    float *alpha_c, *alpha_s, *beta;    /* optional attention-weight outputs */
    int64_t alpha_c_ld, alpha_s_ld, beta_ld;
    const uint8_t *out_mask;            /* dropout keep-mask on h_lang or NULL */
    float out_scale;
    int32_t apply_logsoftmax;           /* normalise `logits` in place after the projection */
    float *hdrop;                       /* [rows,H], required iff out_mask */
    float *logits;                      /* [rows,V] (row stride ld_logits) or NULL */
    int64_t ld_logits;
    float *pmax, *psum;                 /* [rows, ceil(V/128)] */
    int32_t *pidx;
    float *splitk_ws;                   /* optional split-K workspace shared by the step's launches */
    int64_t splitk_ws_floats;
    /* Optional f16 planes of the recurrent state (split-f16 path, see isc_seg.A_hi): [rows,H] each.  The *_prev
     * planes must hold the split of h1_prev / h2_prev (zeros for a zero state); the step writes the planes of h1 / h2
     * from the LSTM epilogues, so a roll-out that swaps (prev, next) every step never splits its state again.  All
     * eight or none. */
    const void *h1_prev_hi, *h1_prev_lo, *h2_prev_hi, *h2_prev_lo;
    void *h1_hi, *h1_lo, *h2_hi, *h2_lo;
    /* Optional plane workspace of the step's own intermediates v, s, f ([rows,E] f16 each; used with the state
     * planes): the scans and the gate write them, the gate sum and the lang-LSTM read them. */
    void *v_hi, *v_lo, *s_hi, *s_lo, *f_hi, *f_lo;
    /* Optional gather mode of the sentiment-word scan (isc_scan_problem.row_ids): words_p / words_e are then the
     * [V,A] / [V,W] tables and words_ids [rows, Mw] (row stride words_ids_ld) the word ids incl. the leading <PAD>. */
    const int64_t *words_ids;
    int64_t words_ids_ld;
    /* Optional (few rows, inference; both attentions): the scans' features through the gate's projection -
     * gate_Gc [rows,R,A] = cont2att.weight att_e rows, gate_Gs = senti2att.weight words_e rows ([rows,Mw,A], or the
     * [V,A] table in gather mode).  Both non-NULL: scans + gate sum + gate mix run as isc_attn_scan_gate_fwd. */
    const float *gate_Gc, *gate_Gs;
    /* Merged step of two sibling unrolls (train_xe.py:160-181: the XE unroll over image regions and the seq2seq unroll
     * over sentiment words share every LSTM / classifier weight, captioner.py:168-186).  pair_rows_c > 0: the first
     * pair_rows_c of the `rows` rows attend to the regions only (att_p / att_e / qa / v / alpha_c index them from 0), the
     * remaining rows to the sentiment words only (words_p / words_e / label_w / qw / s / alpha_s index THEM from 0), no
     * gate; s must equal v + pair_rows_c * E (one [rows,E] attended-feature block feeds the lang-LSTM).  Both LSTM cells
     * and the classifier then run once over all rows, the two h-projections as two problems of one launch and the two
     * scans as the two problems of one isc_attn_scan_fwd launch.  fp32 rows only (no state planes).  0 = off. */
    int32_t pair_rows_c, _pad2;
} isc_step_plan;

int isc_step_fwd(const isc_step_plan *plan_host, void *stream);

/* Stream gate: the general kernels' form of isc_rows_ext.live_in.  While a gate is set for `stream`, the forward
 * launches enqueued on it - isc_linear_fwd, isc_lstm_fwd, isc_vocab_fwd (hence every launch of isc_step_fwd),
 * isc_attn_scan_fwd, isc_attn_scan_gate_fwd, isc_beam_topk - carry `flag` (a device int32) and return at once when it
 * reads 0 at RUN time.  A batched beam search (captioner.py:351-420 for many images at a time) gates step t by live[t]
 * of isc_beam_merge, the images still searching before it: the host looks at that counter only every fourth step
 * (graphs of four steps), and the up to three steps enqueued past the end of the search then cost their launches
 * instead of a full step each.  The small launches of a step (embedding gather, state re-order, merge) are not gated:
 * after the end they copy the finished images' rows along, as they always did.
 * Host state read at enqueue / capture time, per stream; flag == NULL clears it - do that as soon as the gated run of
 * launches is enqueued.  The reference has no counterpart (its loop breaks on the host, captioner.py:379-381). */
int isc_set_stream_gate(const int32_t *flag, void *stream);

/* ------------------------------------------------------------------ decode rows (at most 8 rows, inference)
 * The same forward_step (captioner.py:168-186) for the few-row regimes of the reference: the `beam_size` rows of one
 * image inside sample() (captioner.py:380-411, one batch-1 forward_step per live candidate there), a greedy roll-out
 * of a handful of captions (captioner.py:317-344).  Every contraction is then a few matrix-vector products that
 * stream the weights once and every launch is a few microseconds of dependent memory round trips; csrc/rows.hip holds
 * kernels built for that (weights straight to registers ahead of everything else, 16-lane-per-weight-row layout,
 * index chains in a wave of their own).  Five launches: att-LSTM, the three projections of h_att, the gated scan,
 * lang-LSTM, classifier.  The plan is isc_step_fwd's; this entry point requires both attentions with the
 * pre-projected gate rows (gate_Gc / gate_Gs), A == E == W <= 512, no saved gates, no dropout mask, no in-place
 * log-softmax (isc_rows_step_supported tells) and ignores the f16 plane pointers (nothing on this path reads planes).
 *
 * isc_rows_ext adds what the few-row callers fold into the step:
 *  - src_row: row r of h*_prev / c*_prev is read from row src_row[r] (the beam search's re-ordering of the recurrent
 *    state after each merge, captioner.py:405-409, as an index on the loads instead of a gather launch); NULL = r;
 *  - stats_tile: the column-tile width of pmax / psum / pidx ([rows, ceil(V / stats_tile)]; isc_rows_stats_tile(V):
 *    about V / 256 so that one round of workgroups covers the chip - NOT isc_vocab_fwd's 128);
 *  - beam > 0: per (row, tile) the 8 largest MASKED logits and their word ids, descending, ties to the smaller id
 *    (cand_val / cand_idx [rows, n_tile, 8]; masks as captioner.py:394-399: <PAD>, <SOS>, <UNK> when mask_special,
 *    the row's last word when decoding_constraint) - the input of isc_beam_select.
 *  - row_div > 1: the step-invariant tensors of the plan that belong to an IMAGE (pre1, att_p, att_e, gate_Gc, label_w,
 *    words_ids; per-row words_p / words_e / gate_Gs as well) hold one entry per image and row r uses entry r / row_div -
 *    the `beam` rows of an image share its regions (captioner.py:366-377 expands nothing either: it decodes candidate by
 *    candidate), so a search keeps one copy per image instead of `beam`; rows % row_div == 0;
 *  - live_in: optional device int; when it reads 0 every launch of the step returns at once (and isc_beam_select with it):
 *    a search captured as ONE graph of T steps ends itself on the device (live[t] of isc_beam_select: images still
 *    searching before step t) instead of the host reading that counter between graphs of a few steps.
 * `logits` of the plan is optional here as well (row stride ld_logits). */
typedef struct {
    const int64_t *src_row;
    int32_t stats_tile, beam;
    float *cand_val;
    int32_t *cand_idx;
    const int64_t *last_word;     /* [rows] required iff decoding_constraint */
    int64_t pad_id, sos_id, unk_id;
    int32_t mask_special, decoding_constraint;
    int32_t row_div, _pad;
    const int32_t *live_in;
    /* optional: the PREVIOUS step's greedy roll-out finalize folded into this step's first launch (a roll-out of T steps is
     * then T finalize launches shorter but one: only the last step's runs on its own).  fin_prev describes that step as
     * isc_rollout_finalize would get it (t = the previous step; forced, sample_u and xt_next NULL: arg-max decoding with the
     * token table), except that its `unfinished` is only READ - the updated flags go to fin_unfinished_out ([rows], another
     * array: every workgroup of the launch reads the old flags) - and the plan's `tok` is ignored: the launch derives the
     * fed tokens from the previous step's tile statistics (still in pmax / psum / pidx when it runs).  Needs plan->tab,
     * row_div <= 1, no src_row, no live_in, at most 4 rows (measured: B = 1 0.70 -> 0.65 ms per roll-out, B = 4 0.76 ->
     * 0.74; at 8 rows the fold outlasts the launch it saves: 0.86 -> 0.90). */
    const isc_rollout_step *fin_prev;
    int32_t *fin_unfinished_out;
} isc_rows_ext;
int isc_rows_stats_tile(int V);
int isc_rows_step_supported(const isc_step_plan *plan_host);
int isc_rows_step_fwd(const isc_step_plan *plan_host, const isc_rows_ext *ext_host, void *stream);
/* The classifier launch of that step alone (tests, tools): statistics per stats_tile columns (+ tile candidates). */
int isc_rows_vocab_fwd(const float *h, int ldh, const float *W, int ldw, const float *bias, int M, int V, int K,
                       float *part_max, float *part_sum, int32_t *part_idx, float *logits, int64_t ld_logits,
                       const isc_rows_ext *ext_host, void *stream);
/* Cache policy of the rows kernels' weight streams: 0 = default policy everywhere, 1 (default) = the classifier's weights
 * non-temporal (read once per step: they pass through without evicting the LSTM / projection weights that the XCD L2s
 * keep from step to step), 2 = every stream non-temporal.  Returns the previous value. */
int isc_set_rows_nt(int mode);
long long isc_rows_launches(void);    /* launches of rows kernels so far (tests assert the path was taken) */
/* isc_attn_scan_gate_fwd (and with it isc_step_fwd's gated scan) runs launches of up to this many rows on the rows scan
 * kernel - one 1024-thread workgroup per row, the row's rows of P / V / G in flight at once - when the shape is its own
 * (v / s not wanted on their own, R <= isc_set_rows_scan_regions, Mw <= 12, A = D <= 512); default 256, 0 = never (attn_scan_gate_kernel walks
 * the regions).  rows < 0 only queries.  Returns the previous value. */
int isc_set_rows_scan_max(int rows);
/* ... and up to this many content regions (default 256; up to 36 all of a row's regions are in flight at once, larger
 * grids - the reference encoder's 14 x 14 = 196 - are walked 36 regions at a time by the same kernel).  regions < 0
 * only queries.  Returns the previous value. */
int isc_set_rows_scan_regions(int regions);
/* Vocabulary launches of the skinny split-f16 path (isc_vocab_fwd / isc_step_fwd at a few hundred rows, one activation
 * segment): 1 (default) = gemm_h3v_kernel (k-block stages shared by the workgroup: the activation block fetched once, two
 * blocks in flight; taken up to 512 workgroups), 0 = the per-wave-ring gemm_h3s_kernel form it replaced (tests, A/B runs),
 * n > 1 = on, taken up to n workgroups (tuning).  Returns the previous value. */
int isc_set_h3v(int on);
/* Long contractions on few large split-f16 tiles (one linear problem of < ~200 128 x 128 tiles and >= 64 k-blocks - the
 * classifier's dX over all T x B rows of a training iteration): the k-blocks are cut into up to 8 slices per tile so that
 * the launch fills the chip, partial tiles go to slabs in the caller's workspace and the split-K reduce kernel sums them
 * in fixed order and applies the epilogue (deterministic).  1 (default) = for isc_gemm_bwd's NN launches (dX = dY W),
 * 2 = for forward launches as well (tests: a forward launch otherwise keeps one summation order at every batch size),
 * 0 = never.  Returns the previous value. */
int isc_set_h3_ksplit(int on);

/* Top-k + candidate merge of one beam step in ONE launch (captioner.py:390-411), from the tile statistics and tile
 * candidates isc_rows_step_fwd left: per row the log-softmax normaliser is folded from (pmax, psum), the row's top-`beam`
 * (raw logit descending, ties to the smaller word id; log-prob = (x - max) - log(sum)) is merged out of its tiles'
 * sorted candidate lists, then the image's candidates are formed, scored in fp64 and stably ranked exactly as
 * isc_beam_merge does.  src_row[r] receives the PARENT row of new row r (usable as the next step's isc_rows_ext.src_row; an
 * ended candidate's state is never read again, captioner.py:383-385, so carried rows point at their parent's state as well);
 * top_val / top_idx (optional, [n_img*beam, beam]) receive the rows' top-k for inspection. */
typedef struct {
    int32_t n_img, beam, T, t;
    int32_t n_tile, V;
    int64_t eos_id;
    const float *part_max, *part_sum;     /* [n_img*beam, n_tile] */
    const float *cand_val;                /* [n_img*beam, n_tile, 8] */
    const int32_t *cand_idx;
    const double *score_in;
    double *score_out;
    const int64_t *last_in;
    int64_t *last_out;
    const int64_t *words_in;
    int64_t *words_out;
    const int32_t *len_in;
    int32_t *len_out;
    int32_t *done;
    int64_t *src_row;
    int32_t *live;
    float *top_val;
    int64_t *top_idx;
    const int32_t *live_in;               /* optional: reads 0 -> the launch returns at once (see isc_rows_ext.live_in) */
    /* optional: the recurrent state re-ordered here as well (what isc_beam_gather does in a launch of its own):
     * state_out[p, r, :] = state_in[p, parent(r), :] for the state_planes [n_img*beam, H] planes (h | c x layer) the
     * step has just written; the next step then reads plain rows (no isc_rows_ext.src_row: its kernels stage their
     * activations without a dependent index load in front).  state_out must not alias state_in; H % 4 == 0. */
    const float *state_in;
    float *state_out;
    int32_t state_planes, H;
} isc_beam_select_args;
int isc_beam_select(const isc_beam_select_args *args_host, void *stream);

/* Reverse-sweep counterpart (one BPTT step, see autograd.py): lang-LSTM cell backward, input-gradient
 * contractions, gate / scan backward, att-LSTM cell backward, recurrent gradients for step t-1.
 * `first` = last time step (no incoming recurrent gradients), `last` = step 0 (no outgoing ones). */
typedef struct {
    int32_t rows, H, E, A, W, R, Mw, first, last;
    int32_t pair_rows_c;                /* > 0: the merged step of isc_step_plan.pair_rows_c - rows [0, pair_rows_c) carry the
                                           content scan, the rest the sentiment scan; d_feat is one [rows,E] block, dqa / de_c /
                                           dwc_rows index the content rows from 0, dqw / de_s / dws_rows the sentiment rows */
    const float *Wih1, *Whh1, *Wih2, *Whh2;
    const float *W_h2att, *w_alpha_c, *W_h2word, *w_alpha_s, *W_gh, *W_gc, *W_gs, *w_gate;
    const float *att_p, *att_e, *words_p, *words_e, *label_w;
    /* saved activations of this step */
    const float *g1, *c1_prev, *c1, *g2, *c2_prev, *c2;
    const float *qa, *qw, *v, *s, *z;
    const float *alpha_c, *alpha_s, *beta;
    int64_t alpha_c_ld, alpha_s_ld, beta_ld;
    /* gradients */
    const float *dhd;                   /* [rows,H] d loss / d h_lang of this step (classifier path) */
    float *dG1, *dG2;                   /* [rows,4H] this step's slices of the time-stacked buffers */
    float *dG1_sum;                     /* [rows,4H] running sum over time */
    float *d_feat, *dh1, *dv, *ds;      /* [rows,E] / [rows,H] scratch */
    float *dh2_rec, *dh1_rec;           /* recurrent gradients (in: from t+1, out: for t-1) */
    const float *dc1_in, *dc2_in;       /* cell-state gradients from t+1 (ignored when first) */
    float *dc1_out, *dc2_out;
    float *dqa, *dqw, *dz;              /* [rows,A] this step's slices */
    float *dP_att, *dV_att, *dP_w, *dV_w;       /* accumulated over time */
    float *dwc_rows, *dws_rows, *dwg_rows, *dbg_rows;
    float *splitk_ws;                   /* optional split-K workspace */
    int64_t splitk_ws_floats;
    float *de_c, *de_s;                 /* optional [rows,R] / [rows,Mw]: this step's d e of the two scans (isc_scan_bwd_problem.de_out);
                                           dP_att / dP_w and dV_att / dV_w may then be NULL (isc_attn_dp_from_de, _dv_from_alpha) */
} isc_step_bwd_plan;

int isc_step_bwd(const isc_step_bwd_plan *plan_host, void *stream);

/* ------------------------------------------------------------------ backward (BPTT) */

/* d logits = d logp - softmax * rowsum(d logp)  (autograd of F.log_softmax, captioner.py:183).
 * Output rows have stride ld_out >= V; the padding columns are zero-filled. remap_T > 0
 * writes input row (b,t) = b*T+t to output row t*B+b (time-major copy for the BPTT sweep). */
int isc_logsoftmax_bwd(const float *dlogp, const float *logp, int64_t ld_in, int M, int V,
                       float *dlogits, int64_t ld_out, int remap_T, void *stream);

/* Pointwise part of the LSTMCell backward (captioner.py:175,181): from d h (= dh + dh2),
 * d c_next and the saved activated gates produces d(pre-activation gates) [M,4H] and d c_prev.
 * dgates_sum (optional) += dgates: the step-invariant fc / label inputs of the att-LSTM only
 * need the sum over time. */
int isc_lstm_bwd(const float *dh, const float *dh2, const float *dc_next, const float *gates,
                 const float *c_prev, const float *c, int M, int H, float *dgates, float *dc_prev,
                 float *dgates_sum, void *stream);

/* Backward of isc_attn_scan_fwd for one step. dP / dV / dw_rows accumulate over time steps when
 * `accumulate` is set (first processed step writes). dq [B,A] is also the gradient of q2.
 * dw_rows [B,A]: per-row partial of d w (column-summed once after the loop; deterministic). */
typedef struct {
    const float *P, *V, *q, *q2, *w, *alpha, *dout;
    int64_t alpha_ld;
    int32_t R, A, D, accumulate;
    float *dP, *dV, *dq, *dw_rows;
    float *de_out;            /* optional [B,R]: this step's d e (softmax backward); required when dP is NULL */
    int32_t rows, _pad;       /* rows of THIS problem; 0 = the launch's B (see isc_scan_problem.rows) */
} isc_scan_bwd_problem;

int isc_attn_scan_bwd(const isc_scan_bwd_problem *probs_host, int n_prob, int B, void *stream);
/* dV may be NULL in a problem above: the caller then forms dV once after its sweep from the per-step gradients,
 * dV[b,r,:] = sum_{t = T-1 .. 0} alpha[b*alpha_ld_b + t*alpha_ld_t + r] * dout[(t*B + b)*D + :] (the sweep's order of
 * additions: bit-identical to the per-step accumulation), instead of re-reading and re-writing [B,R,D] at every step. */
int isc_attn_dv_from_alpha(const float *alpha, int64_t alpha_ld_b, int64_t alpha_ld_t, const float *dout,
                           int B, int T, int R, int D, float *dV, int dout_step_rows, void *stream);
/* dout_step_rows (0 = B): rows per step of `dout` - dout row (t, b) at t*dout_step_rows + b (merged unroll, as above). */
/* dP may be NULL as well (with de_out given): dP[b,r,a] = sum_{t = T-1 .. 0} de[(t*B + b)*R + r] * w[a] *
 * (1 - tanh^2(P[b,r,a] + q[(t*B + b)*A + a] (+ q2[b*A + a]))) is then formed once after the sweep - P read once, the
 * tanh terms recomputed; same expression and order of additions as the per-step accumulation.  Any R (the grid walks
 * region chunks: the reference encoder's 14 x 14 = 196 regions are six chunks at A = 512), any A, D with A % 4 == D % 4 == 0. */
int isc_attn_dp_from_de(const float *P, const float *q, const float *q2, const float *w, const float *de,
                        int B, int T, int R, int A, float *dP, void *stream);

/* Backward of isc_gate_mix_fwd: dv = beta*dfeat, ds = (1-beta)*dfeat, dz, per-row partials of
 * d w (dw_rows [B,A]) and d w_bias (db_rows [B]). */
int isc_gate_mix_bwd(const float *z, const float *w, const float *v, const float *s, const float *beta,
                     int64_t beta_ld, const float *dfeat, int B, int A, int D, float *dv, float *ds,
                     float *dz, float *dw_rows, float *db_rows, int accumulate, void *stream);

/* Embedding + ReLU backward: demb[id(r),:] += scale * dout[r / rows_per_grad,:] * (emb[id(r),:] > 0)
 * [* mask]. pad_first = n_words+1 selects the sentiment-word layout (row 0 of every image = <PAD>).
 * Deterministic (no floating-point atomics): the positions of one id are summed in ascending order by the
 * workgroup of its first occurrence.  Rows with id == skip_id are left untouched (nn.Embedding's
 * padding_idx row, whose gradient the caller discards; -1 = none).  W <= 1024. */
int isc_embed_relu_bwd(const float *emb, int V, int W, const int64_t *ids, int64_t ids_stride,
                       int n_rows, int rows_per_grad, int pad_first, int64_t pad_id, const float *dout,
                       float scale, const uint8_t *keep_mask, float mask_scale, float *demb,
                       int64_t skip_id, void *stream);
/* The same gradient, bit for bit, through a position index built in `workspace` (>= (4 V + 64 + n_rows) * 4 bytes,
 * 16-byte aligned): count per id, offsets, every position dropped into its id's segment, one workgroup per
 * occurring id that sorts its segment and sums in ascending position order.  Linear in n_rows where the entry
 * point above is quadratic (each workgroup scans the id array); falls back to it without a workspace or below
 * 1024 positions. */
int isc_embed_relu_bwd_ws(const float *emb, int V, int W, const int64_t *ids, int64_t ids_stride,
                          int n_rows, int rows_per_grad, int pad_first, int64_t pad_id,
                          const float *dout, float scale, const uint8_t *keep_mask, float mask_scale,
                          float *demb, int64_t skip_id, void *workspace, int64_t workspace_bytes, void *stream);

/* out[n] (+)= sum_m x[m,n]  (bias gradients). With a workspace of >= 64*N floats tall matrices are
 * summed in two deterministic stages (row chunks in parallel, then the chunks in order). */
int isc_colsum(const float *x, int64_t ld, int M, int N, float *out, int accumulate, float *workspace,
               int64_t workspace_floats, void *stream);

/* Several column sums in at most two launches (the bias gradients of one backward sweep): job i sums x_i [M_i, N_i]
 * over its rows into each of its n_out outputs (tied biases share one reduction), or adds to them (accumulate).
 * Fixed order, deterministic.  workspace: sum over jobs with M >= 256 of ceil-chunks * N floats (<= 64 N each). */
#define ISC_COLSUM_MAX_JOBS 24
#define ISC_COLSUM_MAX_OUT 3
typedef struct {
    const float *x;
    int64_t ld;
    int32_t M, N;
    float *out[ISC_COLSUM_MAX_OUT];
    int32_t n_out, accumulate;
} isc_colsum_job;
int isc_colsum_multi(const isc_colsum_job *jobs_host, int n_jobs, float *workspace, int64_t workspace_floats,
                     void *stream);

/* dz = dy * (y > 0) [* mask * scale]  (ReLU + Dropout backward of the prologue layers);
 * y == NULL skips the ReLU test (pure nn.Dropout backward, captioner.py:182). */
int isc_relu_mask_bwd(const float *dy, const float *y, const uint8_t *keep_mask, float scale, int64_t n,
                      float *dz, void *stream);

/* XECriterion backward: dlogp (pre-zeroed [B,T,V]) gets -gout/count at every unmasked target;
 * sum_count = the two floats written by isc_xe_loss_fwd. */
int isc_xe_loss_bwd(const int64_t *target, const int32_t *lengths, int B, int T, int V,
                    const float *gout, const float *sum_count, float *dlogp, void *stream);

/* XECriterion backward in the sparse form isc_logsoftmax_bwd_sparse takes: coef [B,T] = -gout/count at unmasked
 * tokens and 0 elsewhere (the column of row (b,t) is target[b,t] itself) - no [B,T,V] tensor. */
int isc_xe_loss_bwd_sparse(const int32_t *lengths, int B, int T, const float *gout, const float *sum_count,
                           float *coef, void *stream);

/* RewardCriterion (self_critical/utils.py:169-177; called at models/decoder.py:160): the masked REINFORCE loss
 * -sum(logp * mask * reward) / sum(mask) over [B,T] fp32 tensors, one launch each way.
 * fwd: out2 = { sum(-logp*mask*reward), sum(mask) } (the caller divides: keeps sum and count for the backward and for
 * data-parallel normalisation).  bwd: d_seq_logprobs[b,t] = -gout[0] * mask * reward / sum_count[1]. */
int isc_reward_loss_fwd(const float *seq_logprobs, const float *seq_masks, const float *reward, int B, int T,
                        float *out2, void *stream);
int isc_reward_loss_bwd(const float *seq_masks, const float *reward, int B, int T, const float *gout,
                        const float *sum_count, float *d_seq_logprobs, void *stream);

/* Log-softmax backward with the criteria's gradient handed over SPARSE: XECriterion (captioner.py:427-440) and the
 * REINFORCE gather (captioner.py:336) touch one column per (caption, step) row, so their d log-prob is `coef[m]` at
 * column `ids[m]` - up to ISC_SPARSE_MAX such (ids, coef) pairs (HOST arrays of device pointers to [M] vectors, rows
 * in [B,T] order) plus an optional dense part `dlogp_dense` [M, ld_in] (NULL when every consumer of the log-probs was
 * one of the two criteria: no [B,T,V] gradient tensor is then written or read at all).
 *   dlogits[m', v] = scale * (dense[m,v] + sum_j coef_j[m] [v == ids_j[m]] - exp(logp[m,v]) * (sum_v dense + sum_j coef_j))
 * `scale`: optional device scalar (isc_grad_scale's out2[0]); rows are written time-major when remap_T > 0 and
 * columns V..ld_out-1 are zero-filled, as isc_logsoftmax_bwd does. */
#define ISC_SPARSE_MAX 2
int isc_logsoftmax_bwd_sparse(const float *dlogp_dense, const float *logp, int64_t ld_in, int M, int V,
                              const int64_t *const *ids_host, const float *const *coef_host, int n_sparse,
                              const float *scale, float *dlogits, int64_t ld_out, int remap_T, int out_step_rows,
                              void *stream);
/* out_step_rows (0 = M / remap_T): rows per step of the time-major output - input row (b,t) goes to output row
 * t*out_step_rows + b (a merged unroll's d logits hold both branches' rows per step; dlogits then points at this
 * branch's first row). */

/* The criteria on RAW logits (training iterations that never materialise the [B,T,V] log-probs, captioner.py:183 + 232):
 * the criteria read one column per (caption, step) row, and log p(id) = (x[id] - max) - log(sum exp) follows from the raw
 * logits and the classifier's tile statistics - the same bits isc_logsoftmax_apply_steps would have stored.
 * Logits row (b,t) at raw + b*ld_b + t*ld_t; statistics row of (b,t) = t*step_rows + b (0 = B); ids / out / coef in
 * [B,T] order.
 *   isc_gather_logp_raw:     out[b,t] = log p(ids[b,t]) (* live[t] when given: captioner.py:336 after the early break)
 *   isc_xe_loss_tokens_fwd:  XECriterion on such per-token log-probs: out2 = { -sum_{t < len_b} tlp[b,t], count }
 *   isc_logsoftmax_bwd_raw:  d logits (time-major rows t*out_step_rows + b, zero-padded to ld_out) from up to
 *                            ISC_SPARSE_MAX (ids, coef) pairs, as isc_logsoftmax_bwd_sparse forms it from stored log-probs. */
int isc_gather_logp_raw(const float *raw, int64_t ld_b, int64_t ld_t, int B, int T, int V, const float *part_max,
                        const float *part_sum, int step_rows, const int64_t *ids, const float *live, float *out,
                        void *stream);
int isc_xe_loss_tokens_fwd(const float *tlp, const int32_t *lengths, int B, int T, float *out2, void *stream);
int isc_logsoftmax_bwd_raw(const float *raw, int64_t ld_b, int64_t ld_t, int B, int T, int V, const float *part_max,
                           const float *part_sum, int step_rows, const int64_t *const *ids_host,
                           const float *const *coef_host, int n_sparse, const float *scale, float *dlogits, int64_t ld_out,
                           int out_step_rows, void *stream);

/* Power-of-two gradient scale for a backward sweep whose contractions run on the split-f16 engine: out4[0..1] = { S, 1/S },
 * S = 2^k with max |x| over the given tensors (HOST arrays of device pointers / element counts, <= ISC_SCALE_SRC_MAX)
 * brought into [2^-4, 2^-3); S = 1 when they are all zero.  out4[2..3] are state words of the multi-workgroup reduction:
 * the caller hands them in ZEROED (once; every call leaves them zeroed).  The caller multiplies what enters the sweep by
 * S (exact) and the parameter gradients by 1/S (exact): gradients of a token-mean loss are <= 1/N_tokens, below the f16
 * normal range at training batch sizes, where the planes x = hi + lo 2^-11 would keep ~22 bits relative to 2^-14 instead
 * of to the element. */
#define ISC_SCALE_SRC_MAX 4
int isc_grad_scale(const float *const *src_host, const int64_t *numel_host, int n_src, float *out4, void *stream);

/* Workspace sizes (the library allocates nothing; these say what to hand it).
 * isc_splitk_workspace_bytes: bytes that let a launch with an [M, N] output use the deepest K split (16 slabs);
 *   isc_*_problem.splitk_ws may be smaller (fewer slabs) or NULL (no split, no out-of-scope split-f16 launches).
 * isc_h3_weights_workspace_bytes: bytes of an isc_h3_weights_begin buffer that holds the f16 planes (hi + lo) of
 *   `weight_elements` fp32 weight values, twice that when the backward's transposed planes are wanted too. */
int64_t isc_splitk_workspace_bytes(int64_t M, int64_t N);
int64_t isc_h3_weights_workspace_bytes(int64_t weight_elements, int with_transposes);

/* Sticky numerics status.  The caller registers two 32-bit words of HOST memory that the device can write (pinned /
 * mapped: hipHostMalloc, a pinned torch tensor) with isc_set_status_words - per device, after selecting it; NULL
 * unregisters.  A kernel that meets a non-finite value stores 1 into its word (error path only; nothing is written
 * otherwise), and isc_status reads the words on the host WITHOUT any device call: it reports what the work the host has
 * already waited for has flagged.  Bits of the return value:
 *   ISC_STATUS_NONFINITE_STATS (1)  a vocabulary-statistics row (max / sum exp) was non-finite when a decode step folded
 *                                   it (roll-out finalize, scheduled sampling, log-softmax passes, beam top-k)
 *   ISC_STATUS_NONFINITE_LINEAR (2) a split-f16 linear launch over caller data staged as fp32 rows (the prologue's raw
 *                                   region features, training-mode activations) produced a non-finite pre-activation
 * Either means an operand left the split-f16 domain |x| < 65504 (its hi plane is inf) or was NaN / inf on entry (the
 * linear epilogues' ReLU lets NaN through, as torch's does, so it reaches the statistics); results since the last clean
 * read are not to be trusted.  isc_set_h3_mode(0) runs the exact-fp32 tiles, whose domain is fp32's.
 * reset != 0 clears the words when any bit was set.  Both calls act on the CURRENT device (hipGetDevice). */
#define ISC_STATUS_NONFINITE_STATS 1
#define ISC_STATUS_NONFINITE_LINEAR 2
int isc_set_status_words(unsigned int *host_words2);
int isc_status(int reset);

/* clip_gradient (train_xe.py:19-23, decoder.py:14-18: elementwise clamp_ to +-clip, in place)
 * followed by torch.optim.Adam's update (captioner.py:422-423), all tensors in one launch.
 * Pointer tables are HOST arrays of device pointers. clip <= 0 disables the clamp. */
#define ISC_ADAM_MAX_TENSORS 48
int isc_clamp_adam(float *const *params_host, float *const *grads_host, float *const *exp_avg_host,
                   float *const *exp_avg_sq_host, const int64_t *numel_host, int n_tensors, double lr,
                   double beta1, double beta2, double eps, double weight_decay, double clip, int step,
                   void *stream);
/* The same launch with its step-dependent scalars in DEVICE memory: hyper3_dev = {lr, 1 - beta1^step,
 * sqrt(1 - beta2^step)} as floats (what isc_clamp_adam derives from lr / step on the host, in double).  For a training
 * iteration captured into a HIP graph: kernel arguments are frozen at capture, the caller rewrites these three
 * floats before every replay (train_graph.py). */
int isc_clamp_adam_hyper(float *const *params_host, float *const *grads_host, float *const *exp_avg_host,
                         float *const *exp_avg_sq_host, const int64_t *numel_host, int n_tensors,
                         const float *hyper3_dev, double beta1, double beta2, double eps, double weight_decay,
                         double clip, void *stream);

#ifdef __cplusplus
}
#endif
#endif
