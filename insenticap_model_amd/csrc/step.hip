// Whole-step host drivers: isc_step_fwd / isc_step_bwd enqueue every kernel of one decode step (or
// one BPTT step) from C++, so the host language pays one FFI call per step instead of ~10
// descriptor-building calls (include/insenticap_hip.h, "whole decode step").  Pure host code: it
// only fills the per-kernel descriptors and calls the library's own entry points.
#include "common.h"
#include <atomic>

static inline isc_seg seg(const float *A, int lda, const float *W, int ldw, int K, const void *hi = nullptr,
                          const void *lo = nullptr) {
    isc_seg s = {A, W, lda, ldw, K, 0, hi, lo};
    return s;
}

#define RET(x)            \
    do {                  \
        int rc__ = (x);   \
        if (rc__) return rc__; \
    } while (0)

extern "C" int isc_step_fwd(const isc_step_plan *p, void *stream) {
    if (!p) return ISC_E_NULL;
    const int rows = p->rows, H = p->H, E = p->E, A = p->A, W = p->W, V = p->V;
    if (rows <= 0 || H <= 0) return ISC_E_SHAPE;
    const bool has_c = p->att_e != nullptr, has_s = p->words_e != nullptr;
    if (!has_c && !has_s) return ISC_E_NULL;
    // merged step of two sibling unrolls: rows [0, rows_c) scan the regions, rows [rows_c, rows) the sentiment words
    const bool pair = p->pair_rows_c > 0;
    if (p->pair_rows_c < 0 || (pair && (!has_c || !has_s || p->pair_rows_c >= rows))) return ISC_E_SHAPE;
    const bool gate = has_c && has_s && !pair;
    const int rows_c = pair ? p->pair_rows_c : rows, rows_s = pair ? rows - p->pair_rows_c : rows;
    const long long off_s = pair ? p->pair_rows_c : 0;           // first sentiment row inside the step's row block
    if (pair && (p->s != p->v + off_s * E || p->gate_Gc || p->gate_Gs)) return ISC_E_SHAPE;
    const int ld1 = H + E + W, ld2 = E + H;
    // f16 planes of the recurrent state (split-f16 path): all eight or none
    const bool planes = p->h1_prev_hi && p->h1_prev_lo && p->h2_prev_hi && p->h2_prev_lo && p->h1_hi && p->h1_lo &&
                        p->h2_hi && p->h2_lo;
    if (!planes && (p->h1_prev_hi || p->h1_prev_lo || p->h2_prev_hi || p->h2_prev_lo || p->h1_hi || p->h1_lo ||
                    p->h2_hi || p->h2_lo))
        return ISC_E_NULL;
    if (pair && planes) return ISC_E_SHAPE;
#define PL(x) (planes ? (x) : nullptr)
    const bool wplanes = planes && (!has_c || (p->v_hi && p->v_lo)) && (!has_s || (p->s_hi && p->s_lo)) &&
                         (!gate || (p->f_hi && p->f_lo));
#define PW(x) (wplanes ? (x) : nullptr)

    // att-LSTM over cat[h_lang_prev, fc, xt] (captioner.py:174-175); fc/label/bias are in pre1
    {
        isc_lstm_problem l = {};
        int n = 0;
        l.seg[n++] = seg(p->h2_prev, H, p->Wih1, ld1, H, PL(p->h2_prev_hi), PL(p->h2_prev_lo));
        if (!p->tab) l.seg[n++] = seg(p->xt, W, p->Wih1 + H + E, ld1, W);
        l.seg[n++] = seg(p->h1_prev, H, p->Whh1, H, H, PL(p->h1_prev_hi), PL(p->h1_prev_lo));
        l.nseg = n; l.M = rows; l.H = H;
        l.h_hi = PL(p->h1_hi); l.h_lo = PL(p->h1_lo);
        l.c_prev = p->c1_prev; l.h_out = p->h1; l.c_out = p->c1; l.gates_out = p->g1;
        l.pre = p->pre1; l.tab = p->tab; l.tab_ids = p->tok; l.tab_ids_stride = p->tok_stride;
        l.splitk_ws = p->splitk_ws; l.splitk_ws_floats = p->splitk_ws_floats;
        RET(isc_lstm_fwd(&l, stream));
    }
    // projections of h_att: h2att (content), h2word (sentiment), h2att of the gate - one launch
    {
        isc_linear_problem q[3] = {};
        int n = 0;
        auto lin = [&](const float *Wm, const float *b, float *out, long long row0, int M) {
            isc_linear_problem &x = q[n++];
            x.seg[0] = seg(p->h1 + row0 * H, H, Wm, H, H, row0 ? nullptr : PL(p->h1_hi), row0 ? nullptr : PL(p->h1_lo));
            x.nseg = 1; x.M = M; x.N = A; x.bias0 = b; x.ldc = A; x.C = out;
        };
        if (has_c) lin(p->W_h2att, p->b_h2att, p->qa, 0, rows_c);
        if (has_s) lin(p->W_h2word, p->b_h2word, p->qw, off_s, rows_s);
        if (gate) lin(p->W_gh, p->b_gh, p->z, 0, rows);
        q[0].splitk_ws = p->splitk_ws; q[0].splitk_ws_floats = p->splitk_ws_floats;
        RET(isc_linear_fwd(q, n, stream));
    }
    // few rows, inference: both scans + gate sum + gate mix as one launch (isc_attn_scan_gate_fwd)
    const bool fused_gate = gate && p->gate_Gc && p->gate_Gs;
    // attention scans (captioner.py:23-35, 50-62)
    {
        isc_scan_problem sc[2] = {};
        int n = 0;
        if (has_c) {
            isc_scan_problem &x = sc[n++];
            x.P = p->att_p; x.V = p->att_e; x.q = p->qa; x.w = p->w_alpha_c; x.w_bias = p->b_alpha_c;
            x.R = p->R; x.A = A; x.D = E; x.out = p->v; x.alpha_out = p->alpha_c; x.alpha_ld = p->alpha_c_ld;
            x.out_hi = PW(p->v_hi); x.out_lo = PW(p->v_lo);
            x.rows = rows_c;
        }
        if (has_s) {
            isc_scan_problem &x = sc[n++];
            x.P = p->words_p; x.V = p->words_e; x.q = p->qw; x.q2 = p->label_w; x.w = p->w_alpha_s;
            x.w_bias = p->b_alpha_s; x.R = p->Mw; x.A = A; x.D = W; x.out = p->s; x.alpha_out = p->alpha_s;
            x.alpha_ld = p->alpha_s_ld;
            x.out_hi = PW(p->s_hi); x.out_lo = PW(p->s_lo);
            x.row_ids = p->words_ids; x.row_ids_ld = p->words_ids_ld;
            x.rows = rows_s;
        }
        if (fused_gate) {
            isc_scan_gate_args g = {};
            g.scan[0] = sc[0]; g.scan[1] = sc[1];
            g.scan[0].out = nullptr; g.scan[0].out_hi = g.scan[0].out_lo = nullptr;     // v and s live on in f only
            g.scan[1].out = nullptr; g.scan[1].out_hi = g.scan[1].out_lo = nullptr;
            g.G[0] = p->gate_Gc; g.G[1] = p->gate_Gs;
            g.zh = p->z; g.b_gc = p->b_gc; g.b_gs = p->b_gs; g.w_gate = p->w_gate; g.b_gate = p->b_gate;
            g.f = p->f; g.f_hi = PW(p->f_hi); g.f_lo = PW(p->f_lo); g.beta = p->beta; g.beta_ld = p->beta_ld;
            RET(isc_attn_scan_gate_fwd(&g, rows, stream));
        } else {
            RET(isc_attn_scan_fwd(sc, n, rows, stream));
        }
    }
    const float *feat = has_c ? p->v : p->s;          // (pair: s continues v - one [rows,E] block)
    const void *feat_hi = has_c ? PW(p->v_hi) : PW(p->s_hi), *feat_lo = has_c ? PW(p->v_lo) : PW(p->s_lo);
    if (fused_gate) {
        feat = p->f; feat_hi = PW(p->f_hi); feat_lo = PW(p->f_lo);
    } else if (gate) {  // z += cont2att(v) + senti2att(s); beta, mix (captioner.py:107-117)
        isc_linear_problem x = {};
        x.seg[0] = seg(p->v, E, p->W_gc, E, E, PW(p->v_hi), PW(p->v_lo));
        x.seg[1] = seg(p->s, W, p->W_gs, W, W, PW(p->s_hi), PW(p->s_lo));
        x.nseg = 2; x.M = rows; x.N = A; x.bias0 = p->b_gc; x.bias1 = p->b_gs; x.ldc = A; x.C = p->z;
        x.accumulate = 1;
        x.splitk_ws = p->splitk_ws; x.splitk_ws_floats = p->splitk_ws_floats;
        RET(isc_linear_fwd(&x, 1, stream));
        RET(isc_gate_mix_fwd(p->z, p->w_gate, p->b_gate, p->v, p->s, rows, A, E, p->f, p->beta, p->beta_ld,
                             PW(p->f_hi), PW(p->f_lo), stream));
        feat = p->f; feat_hi = PW(p->f_hi); feat_lo = PW(p->f_lo);
    }
    // lang-LSTM over cat[feat, h_att] (captioner.py:180-181) (+ dropout on h_lang, :182)
    {
        isc_lstm_problem l = {};
        l.seg[0] = seg(feat, E, p->Wih2, ld2, E, feat_hi, feat_lo);
        l.seg[1] = seg(p->h1, H, p->Wih2 + E, ld2, H, PL(p->h1_hi), PL(p->h1_lo));
        l.seg[2] = seg(p->h2_prev, H, p->Whh2, H, H, PL(p->h2_prev_hi), PL(p->h2_prev_lo));
        l.nseg = 3; l.M = rows; l.H = H; l.b_ih = p->b_ih2; l.b_hh = p->b_hh2;
        l.h_hi = PL(p->h2_hi); l.h_lo = PL(p->h2_lo);
        l.c_prev = p->c2_prev; l.h_out = p->h2; l.c_out = p->c2; l.gates_out = p->g2;
        l.h_keep_mask = p->out_mask; l.mask_scale = p->out_scale; l.hdrop_out = p->hdrop;
        l.splitk_ws = p->splitk_ws; l.splitk_ws_floats = p->splitk_ws_floats;
        RET(isc_lstm_fwd(&l, stream));
    }
    // classifier + log-softmax statistics (captioner.py:183); pmax == NULL: the caller projects all steps' h_lang at
    // once after its unroll (teacher-forced training: nothing reads a step's logits before the unroll ends)
    if (p->pmax)
    RET(isc_vocab_fwd(p->out_mask ? p->hdrop : p->h2, H, p->W_cls, H, p->b_cls, rows, V, H, p->logits,
                      p->ld_logits, p->pmax, p->psum, p->pidx, p->out_mask ? nullptr : PL(p->h2_hi),
                      p->out_mask ? nullptr : PL(p->h2_lo), p->splitk_ws, p->splitk_ws_floats, stream));
#undef PL
#undef PW
    if (p->apply_logsoftmax) {
        if (!p->logits || !p->pmax) return ISC_E_NULL;
        RET(isc_logsoftmax_apply(p->logits, p->ld_logits, rows, V, p->pmax, p->psum, nullptr, stream));
    }
    return ISC_OK;
}

static inline isc_linear_problem nn_problem(const float *A, int lda, const float *Wm, int ldw, int K, int M,
                                            int N, float *C, int accumulate) {
    isc_linear_problem x = {};
    x.seg[0] = seg(A, lda, Wm, ldw, K);
    x.nseg = 1; x.M = M; x.N = N; x.ldc = N; x.C = C; x.accumulate = accumulate;
    return x;
}

extern "C" int isc_step_bwd(const isc_step_bwd_plan *p, void *stream) {
    if (!p) return ISC_E_NULL;
    const int rows = p->rows, H = p->H, E = p->E, A = p->A, W = p->W;
    const bool has_c = p->att_e != nullptr, has_s = p->words_e != nullptr;
    const bool pair = p->pair_rows_c > 0;             // merged step of two sibling unrolls (isc_step_plan.pair_rows_c)
    if (p->pair_rows_c < 0 || (pair && (!has_c || !has_s || p->pair_rows_c >= rows))) return ISC_E_SHAPE;
    const bool gate = has_c && has_s && !pair;
    const int rows_c = pair ? p->pair_rows_c : rows, rows_s = pair ? rows - p->pair_rows_c : rows;
    const long long off_s = pair ? p->pair_rows_c : 0;
    const int ld1 = H + E + W, ld2 = E + H, G = 4 * H;
    const int acc = p->first ? 0 : 1;      // time-accumulated buffers: the first processed step writes

    // lang-LSTM cell
    RET(isc_lstm_bwd(p->dhd, p->first ? nullptr : p->dh2_rec, p->first ? nullptr : p->dc2_in, p->g2,
                     p->c2_prev, p->c2, rows, H, p->dG2, p->dc2_out, nullptr, stream));
    {   // d feat, d h_att (lang-LSTM input part), d h_lang_prev (recurrent part)
        isc_linear_problem q[3] = {nn_problem(p->dG2, G, p->Wih2, ld2, G, rows, E, p->d_feat, 0),
                                   nn_problem(p->dG2, G, p->Wih2 + E, ld2, G, rows, H, p->dh1, 0),
                                   nn_problem(p->dG2, G, p->Whh2, H, G, rows, H, p->dh2_rec, 0)};
        q[0].splitk_ws = p->splitk_ws; q[0].splitk_ws_floats = p->splitk_ws_floats;
        RET(isc_gemm_bwd(q, 3, ISC_LAYOUT_NN, stream));
    }
    const float *dv = p->d_feat, *dsw = p->d_feat + off_s * E;
    if (gate) {
        RET(isc_gate_mix_bwd(p->z, p->w_gate, p->v, p->s, p->beta, p->beta_ld, p->d_feat, rows, A, E, p->dv,
                             p->ds, p->dz, p->dwg_rows, p->dbg_rows, acc, stream));
        isc_linear_problem q[3] = {nn_problem(p->dz, A, p->W_gc, E, A, rows, E, p->dv, 1),
                                   nn_problem(p->dz, A, p->W_gs, W, A, rows, W, p->ds, 1),
                                   nn_problem(p->dz, A, p->W_gh, H, A, rows, H, p->dh1, 1)};
        q[0].splitk_ws = p->splitk_ws; q[0].splitk_ws_floats = p->splitk_ws_floats;
        RET(isc_gemm_bwd(q, 3, ISC_LAYOUT_NN, stream));
        dv = p->dv;
        dsw = p->ds;
    }
    {
        isc_scan_bwd_problem sc[2] = {};
        int n = 0;
        if (has_c) {
            isc_scan_bwd_problem &x = sc[n++];
            x.P = p->att_p; x.V = p->att_e; x.q = p->qa; x.w = p->w_alpha_c; x.alpha = p->alpha_c;
            x.alpha_ld = p->alpha_c_ld; x.dout = dv; x.R = p->R; x.A = A; x.D = E; x.accumulate = acc;
            x.dP = p->dP_att; x.dV = p->dV_att; x.dq = p->dqa; x.dw_rows = p->dwc_rows; x.de_out = p->de_c;
            x.rows = rows_c;
        }
        if (has_s) {
            isc_scan_bwd_problem &x = sc[n++];
            x.P = p->words_p; x.V = p->words_e; x.q = p->qw; x.q2 = p->label_w; x.w = p->w_alpha_s;
            x.alpha = p->alpha_s; x.alpha_ld = p->alpha_s_ld; x.dout = dsw; x.R = p->Mw; x.A = A; x.D = W;
            x.accumulate = acc; x.dP = p->dP_w; x.dV = p->dV_w; x.dq = p->dqw; x.dw_rows = p->dws_rows; x.de_out = p->de_s;
            x.rows = rows_s;
        }
        RET(isc_attn_scan_bwd(sc, n, rows, stream));
    }
    if (pair) {   // d h_att[content rows] += dqa W_h2att, d h_att[sentiment rows] += dqw W_h2word: two problems, one launch
        isc_linear_problem q[2] = {nn_problem(p->dqa, A, p->W_h2att, H, A, rows_c, H, p->dh1, 1),
                                   nn_problem(p->dqw, A, p->W_h2word, H, A, rows_s, H, p->dh1 + off_s * H, 1)};
        q[0].splitk_ws = p->splitk_ws; q[0].splitk_ws_floats = p->splitk_ws_floats;
        RET(isc_gemm_bwd(q, 2, ISC_LAYOUT_NN, stream));
    } else {   // d h_att += dqa W_h2att + dqw W_h2word
        isc_linear_problem x = {};
        int n = 0;
        if (has_c) x.seg[n++] = seg(p->dqa, A, p->W_h2att, H, A);
        if (has_s) x.seg[n++] = seg(p->dqw, A, p->W_h2word, H, A);
        x.nseg = n; x.M = rows; x.N = H; x.ldc = H; x.C = p->dh1; x.accumulate = 1;
        x.splitk_ws = p->splitk_ws; x.splitk_ws_floats = p->splitk_ws_floats;
        RET(isc_gemm_bwd(&x, 1, ISC_LAYOUT_NN, stream));
    }
    // att-LSTM cell
    RET(isc_lstm_bwd(p->dh1, p->first ? nullptr : p->dh1_rec, p->first ? nullptr : p->dc1_in, p->g1,
                     p->c1_prev, p->c1, rows, H, p->dG1, p->dc1_out, p->dG1_sum, stream));
    if (!p->last) {  // recurrent gradients for step t-1
        isc_linear_problem q[2] = {nn_problem(p->dG1, G, p->Wih1, ld1, G, rows, H, p->dh2_rec, 1),
                                   nn_problem(p->dG1, G, p->Whh1, H, G, rows, H, p->dh1_rec, 0)};
        q[0].splitk_ws = p->splitk_ws; q[0].splitk_ws_floats = p->splitk_ws_floats;
        RET(isc_gemm_bwd(q, 2, ISC_LAYOUT_NN, stream));
    }
    return ISC_OK;
}

// ------------------------------------------------------------------ stream gate
// The decode steps a batched beam search enqueues past its own end (the host looks at the live-image counter every fourth
// step only) should cost their launches, not a full step each - the few-row step has isc_rows_ext.live_in for that; the
// general kernels take the flag from here instead of from every problem struct of the ABI: while a gate is set
// for a stream, the forward launches enqueued on it (isc_linear_fwd, isc_lstm_fwd, isc_vocab_fwd - hence isc_step_fwd -,
// isc_attn_scan_fwd, isc_attn_scan_gate_fwd, isc_beam_topk) carry the pointer and return at once when it reads 0.
// Host state, read at enqueue (= capture) time; the caller clears it when the gated run of launches is enqueued.
#include <mutex>
#define ISC_GATE_SLOTS 64
static struct { void *stream; const int *flag; bool used; } g_gates[ISC_GATE_SLOTS] = {};
static std::mutex g_gates_mu;
static std::atomic<int> g_gates_set{0};            // (fast path: nobody has a gate)

const int *isc_stream_gate_(void *stream) {
    if (g_gates_set.load(std::memory_order_relaxed) == 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_gates_mu);
    for (int i = 0; i < ISC_GATE_SLOTS; ++i)
        if (g_gates[i].used && g_gates[i].stream == stream) return g_gates[i].flag;
    return nullptr;
}

extern "C" int isc_set_stream_gate(const int32_t *flag, void *stream) {
    std::lock_guard<std::mutex> lk(g_gates_mu);
    int free_slot = -1;
    for (int i = 0; i < ISC_GATE_SLOTS; ++i) {
        if (g_gates[i].used && g_gates[i].stream == stream) {
            if (flag) { g_gates[i].flag = flag; return ISC_OK; }
            g_gates[i].used = false;
            g_gates_set.fetch_sub(1);
            return ISC_OK;
        }
        if (!g_gates[i].used && free_slot < 0) free_slot = i;
    }
    if (!flag) return ISC_OK;                      // nothing to clear
    if (free_slot < 0) return ISC_E_WORKSPACE;
    g_gates[free_slot].stream = stream; g_gates[free_slot].flag = flag; g_gates[free_slot].used = true;
    g_gates_set.fetch_add(1);
    return ISC_OK;
}

// ------------------------------------------------------------------ numerics status
int isc_set_status_gemm_(unsigned int *p);         // gemm_f32.hip
int isc_set_status_pw_(unsigned int *p);           // pointwise.hip
int isc_set_status_rows_(unsigned int *p);         // rows.hip
#define ISC_STATUS_MAX_DEVICES 64
static unsigned int *g_status_host[ISC_STATUS_MAX_DEVICES] = {};     // per device: the words registered for it

extern "C" int isc_set_status_words(unsigned int *words2) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ISC_STATUS_MAX_DEVICES) return 1;
    if (isc_set_status_gemm_(words2) || isc_set_status_pw_(words2) || isc_set_status_rows_(words2)) return 1;
    g_status_host[dev] = words2;
    return ISC_OK;
}

extern "C" int isc_status(int reset) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ISC_STATUS_MAX_DEVICES) return -1;
    unsigned int *w = g_status_host[dev];
    if (!w) return 0;
    volatile unsigned int *vw = w;
    const int bits = (vw[0] ? ISC_STATUS_NONFINITE_STATS : 0) | (vw[1] ? ISC_STATUS_NONFINITE_LINEAR : 0);
    if (reset && bits) { vw[0] = 0; vw[1] = 0; }
    return bits;
}
