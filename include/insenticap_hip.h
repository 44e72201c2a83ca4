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
 *   - no allocation, no ownership transfer, no global mutable state, no device
 *     synchronisation: work is enqueued on `stream` (a hipStream_t passed as void*);
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

#define ISC_MAX_SEG 4

/* Library / build identification ("gfx950", ABI version). */
int isc_abi_version(void);
const char *isc_target_arch(void);

/* One K-segment of a contraction: acc += A[M,K] * W[N,K]^T.  Replaces the
 * torch.cat([...],1) + nn.Linear / nn.LSTMCell pattern of captioner.py:174-175,180-181. */
typedef struct {
    const float *A; /* [M, K] activations, leading dim lda */
    const float *W; /* [N, K] weight slice (row-major, K contiguous), leading dim ldw */
    int32_t lda, ldw, K, _pad;
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
} isc_linear_problem;

int isc_linear_fwd(const isc_linear_problem *probs_host, int n_prob, void *stream);

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
} isc_lstm_problem;

int isc_lstm_fwd(const isc_lstm_problem *prob_host, void *stream);

/* Vocabulary projection with fused log-softmax statistics (captioner.py:183):
 * logits = h W^T + b.  Per 128-column tile the kernel emits (max, argmax, sum exp(x-max));
 * the [M,V] logits are only stored when `logits` is non-null (XE / beam / sampling).
 * n_tile = ceil(V/128); part_* are [M, n_tile]. */
int isc_vocab_fwd(const float *h, int ldh, const float *W, int ldw, const float *bias,
                  int M, int V, int K, float *logits, int64_t ld_logits,
                  float *part_max, float *part_sum, int32_t *part_idx, void *stream);

/* logp[m, :] = logits[m, :] - logsumexp(row) in place, using the tile statistics.
 * (F.log_softmax of captioner.py:183 when the full [B,V] row is an API output.) */
int isc_logsoftmax_apply(float *logits, int64_t ld_logits, int M, int V,
                         const float *part_max, const float *part_sum, float *lse_out,
                         void *stream);

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
    int32_t R, A, D, _pad;
    float *out;       /* [B,D] */
    float *alpha_out; /* [B,*] row stride alpha_ld, R values written per row */
    int64_t alpha_ld;
} isc_scan_problem;

int isc_attn_scan_fwd(const isc_scan_problem *probs_host, int n_prob, int B, void *stream);

/* Gate fusion (captioner.py:111-117): beta = sigmoid(w . tanh(z[b,:]) + *w_bias);
 * out = beta*v + (1-beta)*s.  z = cont2att(v)+senti2att(s)+h2att(h) from isc_linear_fwd. */
int isc_gate_mix_fwd(const float *z, const float *w, const float *w_bias, const float *v,
                     const float *s, int B, int A, int D, float *out, float *beta_out,
                     int64_t beta_ld, void *stream);

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

/* Beam step (sample(), captioner.py:390-409), batched over images: for every live beam row
 * apply the -inf masks (PAD,SOS,UNK, last word), take its top-`beam` (value, id) pairs
 * from logp = logits - lse.  Candidate merge + stable ordering is done by the host mirror
 * in fp64 exactly like the reference's Python floats. */
int isc_beam_topk(const float *logits, int64_t ld_logits, const float *part_max,
                  const float *part_sum, int n_tile, int rows, int V, int beam,
                  const int64_t *last_word, int64_t pad_id, int64_t sos_id, int64_t unk_id,
                  int mask_special, int decoding_constraint, float *top_val, int64_t *top_idx,
                  void *stream);

/* Masked NLL (XECriterion, captioner.py:427-440): returns sum and token count in out[0..1].
 * logp [B,T,V] contiguous, target [B,T] int64, lengths [B] int32. */
int isc_xe_loss_fwd(const float *logp, const int64_t *target, const int32_t *lengths, int B,
                    int T, int V, float *out2, void *stream);

#ifdef __cplusplus
}
#endif
#endif
