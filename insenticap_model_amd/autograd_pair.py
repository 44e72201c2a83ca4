"""The two sibling unrolls of one training iteration as ONE step chain.

The reference runs the XE unroll (image regions, `forward_xe`, captioner.py:194-240) and the seq2seq unroll (sentiment
words, `forward_seq2seq`, captioner.py:242-288) of an iteration as two separate chains of per-step op launches
(train_xe.py:160-181, models/decoder.py:138-157).  Both go through the same `forward_step` (captioner.py:168-186) with
the same att-LSTM / lang-LSTM / classifier / word-embedding weights; only the attention differs (content attention over
the regions vs sentiment attention over the words, no gate in either).  At the reference's batch sizes (128 + 80 rows)
every launch of a step is latency-bound, so two chains cost twice the launches and nothing overlaps (two streams of
one-workgroup-per-CU kernels share nothing but the queue).

Here the rows of both calls form one row block per time step - XE rows first, seq2seq rows after them:
  * both LSTM cells, the classifier, every dX contraction of the reverse sweep and every dW contraction after it run
    ONCE over all rows;
  * the two h-projections are two problems of one launch, the two attention scans the two problems of one scan launch
    (isc_step_plan.pair_rows_c / isc_step_bwd_plan.pair_rows_c);
  * time-stacked activations are [T, B1 + B2, .]; what belongs to one branch only (its attention weights, scores,
    projections) stays in per-branch tensors.
Captions of different lengths (T1 != T2): steps past the shorter unroll run on the longer branch's rows alone; the
all-row contractions then see zeros in the idle rows (the stacks are zero-initialised in that case).

The reverse sweep always accumulates into zeroed recurrent-gradient buffers (no `first` special case): x + 0 == x, so a
branch that joins the sweep late (the shorter unroll) needs nothing special.
"""
import torch

from . import _lib, ops
from .autograd import NN, TN, _Saved, _pad32, _weights_scope, cap_pre


def use_pair(cap, in_graph):
    """Whether an iteration's two unrolls go through the merged step chain.  `captioner.pair_unrolls` (or the environment
    variable ISC_PAIR_UNROLLS = 0 / 1, for A/B runs) forces it; None = by measurement (MI355X, B = 128 + 80, V = 10k):
      eager steps         merged 4.9 ms   vs 6.4 ms one chain per unroll on two streams (the host enqueues ~600 launches)
      from HIP graphs     merged 5.3 ms   vs 4.85 ms two chains as two branches of the graph
    - the step kernels at 208 rows are two rounds of 32 x 32 tiles bound by operand traffic, so one chain saves launches,
    not kernel time, and a graph's two branches fill each other's tails.  Hence: merged when the host issues the
    launches, two branches when a graph does."""
    import os
    env = os.environ.get('ISC_PAIR_UNROLLS')
    v = getattr(cap, 'pair_unrolls', None)
    if env in ('0', '1'):
        v = env == '1'
    if v is None:
        return not in_graph
    return bool(v)


def pair_applicable(cap, masks1, masks2):
    """Both branches must agree on whether h_lang is dropped out (one keep-mask tensor and scale serve all rows)."""
    if ops.TIMER.armed or ops.TIMER.arm_step is not None:
        return False
    p_drop = cap.drop.p

    def has_out_masks(m):
        if m is not None:
            return any(k.startswith('out') for k in m)
        return cap.training and p_drop > 0.0
    return has_out_masks(masks1) == has_out_masks(masks2)


class _Fill:
    """Pointer arithmetic over row blocks of time-stacked tensors (host-side only: no views are created per step)."""

    def __init__(self, Bt):
        self.Bt = Bt

    def at(self, x, t, row, width):
        """Address of row `row` of step `t` of a contiguous fp32 [T, Bt, width] stack."""
        return None if x is None else x.data_ptr() + 4 * (t * self.Bt + row) * width


# ------------------------------------------------------------------------------ forward
def _pair_forward(cap, xe, s2s, lazy=None):
    """xe = (fc, att, cpt_words, tokens_in [B1,T1], senti_labels, ss_prob, masks); s2s = (cpt_words, senti_words,
    tokens_in [B2,T2], senti_labels, ss_prob, masks).  Returns (logp1 [B1,T1,V], logp2 [B2,T2,V], S) - or, with
    lazy = (targets1 [B1,T1], targets2 [B2,T2]) (Captioner.token_logprobs), log p(target) [B1,T1] / [B2,T2] straight from
    the raw logits: the two [B,T,V] log-prob tensors are then never formed (autograd._train_forward, `lazy`)."""
    p = cap._p()
    st = cap.settings
    E, A, H, Wd, V = st['feat_emb_dim'], st['att_hid_dim'], st['rnn_hid_dim'], st['word_emb_dim'], cap.vocab_size
    fc, att, cpt1, tok1, lab1, ss1, masks1 = xe
    cpt2, sw2, tok2, lab2, ss2, masks2 = s2s
    (B1, T1), (B2, T2) = tok1.shape, tok2.shape
    Bt, T, Tmin = B1 + B2, max(T1, T2), min(T1, T2)
    ragged = T1 != T2
    new, zeros = cap._new, cap._zeros
    stack = zeros if ragged else new          # stacks the all-row contractions read: idle rows must hold zeros
    S = _Saved()
    S.p, S.B1, S.B2, S.T1, S.T2 = p, B1, B2, T1, T2
    pre1 = new(Bt, 4 * H)
    P1 = cap._prologue(p, 'xe', fc, att, cpt1, None, lab1, masks1, pre1_out=pre1[:B1])
    S.fc_feats1, S.cpt_feats1 = cap.fc_feats, cap.cpt_feats
    P2 = cap._prologue(p, 'seq2seq', None, None, cpt2, sw2, lab2, masks2, pre1_out=pre1[B1:])
    S.cpt_feats2 = cap.cpt_feats
    P1.fc_pre, P1.cpt_pre = S.fc_feats1, S.cpt_feats1
    P2.fc_pre, P2.cpt_pre = None, S.cpt_feats2
    S.P1, S.P2, S.pre1 = P1, P2, pre1
    R, Mw = P1.R, P2.Mw
    S.h1, S.c1, S.h2, S.c2 = zeros(4, T + 1, Bt, H).unbind(0)
    S.g1, S.g2 = new(T, Bt, 4 * H), new(T, Bt, 4 * H)
    S.xt = stack(T, Bt, Wd)
    S.tok = torch.full((T, Bt), cap.pad_id, dtype=torch.int64, device=cap._dev) if ragged else \
        torch.empty(T, Bt, dtype=torch.int64, device=cap._dev)
    S.qa, S.aC = new(T1, B1, A), new(B1, T1, R)
    S.qw, S.aS = new(T2, B2, A), new(B2, T2, Mw)
    S.feat = stack(T, Bt, E)                  # attended feature per row: v for the XE rows, s for the seq2seq rows
    n_tile = (V + 127) // 128
    pm, ps = new(T, Bt, n_tile), new(T, Bt, n_tile)
    pi = new(T, Bt, n_tile, dtype=torch.int32)
    out1, out2 = (new(B1, T1, V), new(B2, T2, V)) if lazy is None else (new(B1, T1), new(B2, T2))
    raw = new(T, Bt, V)                       # raw logits, time-major: normalised per branch after the last step
    emb = p['word_embed.0.weight']

    # dropout on h_lang (captioner.py:182): one [T, Bt, H] keep-mask for all rows
    mf1, mf2 = cap._mask_source(masks1), cap._mask_source(masks2)
    S.out_masks, S.out_scale, S.hdrop = None, 1.0, None
    if masks1 is None and masks2 is None:
        if cap.training and cap.drop.p > 0.0:
            S.out_masks = torch.empty(T, Bt, H, dtype=torch.uint8, device=cap._dev).bernoulli_(1.0 - cap.drop.p)
            S.out_scale = 1.0 / (1.0 - cap.drop.p) if cap.drop.p < 1.0 else 0.0
    else:                                     # explicit masks (tests replay the reference's)
        m0, sc0 = mf1('out0', B1, H)
        if m0 is not None:
            S.out_masks = torch.ones(T, Bt, H, dtype=torch.uint8, device=cap._dev)
            S.out_scale = sc0
            for t in range(T1):
                S.out_masks[t, :B1] = mf1('out%d' % t, B1, H)[0]
            for t in range(T2):
                S.out_masks[t, B1:] = mf2('out%d' % t, B2, H)[0]
    if S.out_masks is not None:
        S.hdrop = stack(T, Bt, H)

    sampling = cap.training and (ss1 > 0.0 or ss2 > 0.0)
    base = None
    if sampling:       # ground-truth tokens of both branches, time-major: the base ids of every step's draw
        base = torch.full((T, Bt), cap.pad_id, dtype=torch.int64, device=cap._dev)
        base[:T1, :B1].copy_(tok1.t())
        base[:T2, B1:].copy_(tok2.t())
        S.tok[0].copy_(base[0])
        ops.embed_relu_fwd(emb, S.tok[0], S.xt[0])
        u_all = torch.rand(max(T - 1, 1), 2, Bt, device=cap._dev)
    else:              # every fed token is known up front: two copies, one gather; the classifier runs once afterwards
        S.tok[:T1, :B1].copy_(tok1.t())
        S.tok[:T2, B1:].copy_(tok2.t())
        ops.embed_relu_fwd(emb, S.tok.view(-1), S.xt.view(T * Bt, Wd))

    # plans: both branches / XE rows alone / seq2seq rows alone (the last two only past the shorter unroll)
    Pm = type(P1)()
    Pm.B, Pm.R, Pm.Mw = Bt, R, Mw
    Pm.att_e3, Pm.att_p3 = P1.att_e3, P1.att_p3
    Pm.words_e3, Pm.words_p3, Pm.label_w = P2.words_e3, P2.words_p3, P2.label_w
    Pm.pre1 = pre1
    plan_pair = cap._make_plan(p, Pm, Bt)
    plan_pair.pair_rows_c = B1
    plans = {(True, True): plan_pair}
    if ragged:
        Pc = type(P1)()
        Pc.B, Pc.R, Pc.att_e3, Pc.att_p3, Pc.pre1 = B1, R, P1.att_e3, P1.att_p3, pre1[:B1]
        Ps = type(P1)()
        Ps.B, Ps.Mw, Ps.words_e3, Ps.words_p3, Ps.label_w, Ps.pre1 = B2, Mw, P2.words_e3, P2.words_p3, P2.label_w, pre1[B1:]
        plans[(True, False)] = cap._make_plan(p, Pc, B1)
        plans[(False, True)] = cap._make_plan(p, Ps, B2)
    F = _Fill(Bt)
    step_cls = sampling                        # per-step classifier only when a step's logits feed the next draw

    def run_step(t):
        a1, a2 = t < T1, t < T2
        r0, r1 = (0 if a1 else B1), (Bt if a2 else B1)
        pl = plans[(a1, a2)]
        pl.rows = r1 - r0
        pl.xt = F.at(S.xt, t, r0, Wd)
        pl.h1_prev, pl.h2_prev = F.at(S.h1, t, r0, H), F.at(S.h2, t, r0, H)
        pl.c1_prev, pl.c2_prev = F.at(S.c1, t, r0, H), F.at(S.c2, t, r0, H)
        pl.h1, pl.h2 = F.at(S.h1, t + 1, r0, H), F.at(S.h2, t + 1, r0, H)
        pl.c1, pl.c2 = F.at(S.c1, t + 1, r0, H), F.at(S.c2, t + 1, r0, H)
        pl.g1, pl.g2 = F.at(S.g1, t, r0, 4 * H), F.at(S.g2, t, r0, 4 * H)
        if a1:
            pl.qa, pl.v = S.qa.data_ptr() + 4 * t * B1 * A, F.at(S.feat, t, 0, E)
            pl.alpha_c, pl.alpha_c_ld = S.aC.data_ptr() + 4 * t * R, S.aC.stride(0)
        if a2:
            pl.qw, pl.s = S.qw.data_ptr() + 4 * t * B2 * A, F.at(S.feat, t, B1, E)
            pl.alpha_s, pl.alpha_s_ld = S.aS.data_ptr() + 4 * t * Mw, S.aS.stride(0)
        if S.out_masks is not None:
            pl.out_mask = S.out_masks.data_ptr() + (t * Bt + r0) * H
            pl.out_scale, pl.hdrop = S.out_scale, F.at(S.hdrop, t, r0, H)
        else:
            pl.out_mask, pl.out_scale, pl.hdrop = None, 1.0, None
        pl.apply_logsoftmax = 0
        if step_cls:
            pl.logits, pl.ld_logits = F.at(raw, t, r0, V), V
            pl.pmax, pl.psum, pl.pidx = F.at(pm, t, r0, n_tile), F.at(ps, t, r0, n_tile), F.at(pi, t, r0, n_tile)
        else:
            pl.logits, pl.ld_logits, pl.pmax, pl.psum, pl.pidx = None, 0, None, None, None
        ops.step_fwd(pl)

    with _weights_scope(cap):
        for t in range(T):
            if sampling and t >= 1:           # scheduled sampling (captioner.py:219-228): select + draw on the device
                u = u_all[t - 1]
                a1, a2 = t < T1, t < T2
                if a1 and a2 and ss1 == ss2:
                    spans = [(0, Bt, ss1)]
                else:
                    spans = ([(0, B1, ss1)] if a1 else []) + ([(B1, Bt, ss2)] if a2 else [])
                for lo, hi, prob in spans:
                    ops.sched_sample(raw[t - 1, lo:hi], pm[t - 1, lo:hi], ps[t - 1, lo:hi], pi[t - 1, lo:hi],
                                     u[0, lo:hi], u[1, lo:hi], prob, base[t, lo:hi], S.tok[t, lo:hi], raw=True)
                r0, r1 = (0 if a1 else B1), (Bt if a2 else B1)
                ops.embed_relu_fwd(emb, S.tok[t, r0:r1], S.xt[t, r0:r1])
            run_step(t)
        if not step_cls:                       # the classifier once over every step's h_lang [T*Bt, H]
            hs = S.hdrop if S.hdrop is not None else S.h2[1:]
            ops.vocab_fwd(hs.reshape(T * Bt, H), p['classifier.weight'], p['classifier.bias'], pm.view(T * Bt, n_tile),
                          ps.view(T * Bt, n_tile), pi.view(T * Bt, n_tile), raw.view(T * Bt, V))
        S.lazy = None
        if lazy is None:
            ops.logsoftmax_apply_steps(out1, pm[:T1, :B1], ps[:T1, :B1], src_tbv=raw[:T1, :B1], step_rows=Bt)
            ops.logsoftmax_apply_steps(out2, pm[:T2, B1:], ps[:T2, B1:], src_tbv=raw[:T2, B1:], step_rows=Bt)
        else:                                  # log p(target) per row from the raw logits; the backward reads them again
            S.lazy = (raw, pm, ps, lazy[0].contiguous(), lazy[1].contiguous())
            ops.gather_logp_raw(raw, V, Bt * V, B1, T1, V, pm, ps, Bt, S.lazy[3], out1)
            ops.gather_logp_raw(raw[0, B1:], V, Bt * V, B2, T2, V, pm[0, B1:], ps[0, B1:], Bt, S.lazy[4], out2)
    del raw
    # the state the reference's attributes are in after its second call (forward_seq2seq): sentiment weights only
    cap._set_weights(None, S.aS, None, T2)
    S.logp1, S.logp2 = out1, out2
    return out1, out2, S


# ------------------------------------------------------------------------------ backward
def _pair_backward(cap, S, d1, d2, sparse1, sparse2, d_fc_feats1, d_cpt_feats1, d_cpt_feats2):
    """{param name: gradient} of both unrolls.  d1 / d2: dense d log-prob of the two outputs (None when the criteria
    handed theirs over sparse: sparse1 / sparse2 = [(ids, coef)]); optional gradients of the attribute tensors.

    With a gradient sink on the captioner (`cap._grad_sink`, dp.GradSink: the data-parallel step) every gradient is
    written straight into its view of the flat arena, the parameters are finished bucket by bucket - classifier (before
    the sweep), lang-LSTM + attention, att-LSTM + projections, embeddings + fc - and each bucket's all-reduce starts as
    soon as its last contraction is enqueued; the returned dictionary is then empty.  Same kernels, same values."""
    p, P1, P2 = S.p, S.P1, S.P2
    B1, B2, T1, T2 = S.B1, S.B2, S.T1, S.T2
    Bt, T = B1 + B2, max(T1, T2)
    ragged = T1 != T2
    st = cap.settings
    E, A, H, Wd, V = st['feat_emb_dim'], st['att_hid_dim'], st['rnn_hid_dim'], st['word_emb_dim'], cap.vocab_size
    R, Mw = P1.R, P2.Mw
    new, zeros = cap._new, cap._zeros
    stack = zeros if ragged else new
    G = {}
    TB = T * Bt
    sink = getattr(cap, '_grad_sink', None)
    if (P1.label_e is None) != (P2.label_e is None):
        raise ValueError('merged unrolls: sentiment labels for both calls or for neither')

    def nn(segs, out, acc=False):
        return ops.gemm_problem(segs, out, NN, accumulate=acc)

    def gout(name, *shape):
        """The tensor parameter `name`'s gradient is computed into: its arena view under a sink, else a new tensor."""
        t = sink.out(name) if sink is not None else new(*p[name].shape)
        if sink is None:
            G[name] = t                 # (parameter-shaped: autograd checks the shape of what the node returns)
        n = 1
        for d in shape:
            n *= d
        assert t.numel() == n, (name, tuple(t.shape), shape)
        return t.view(*shape)

    def tn(a, w, name):
        out = gout(name, a.shape[1], w.shape[1])
        ops.gemm_bwd([ops.gemm_problem([(a, w)], out, TN)], TN)
        return out

    pending_sums = []      # bias gradients of the bucket in progress: one isc_colsum_multi per bucket

    def csum(x, *names):
        """Column sum of x into the gradient(s) `names` (several: tied biases share one reduction), deferred."""
        pending_sums.append((x, [gout(n, x.shape[1]) for n in names], False))

    def zero_grad_of(name):
        if sink is None:
            G[name] = zeros(1)          # (under a sink the arena was zeroed before the backward)

    # ---- gradient scale (autograd._backward): one power of two for everything that enters the sweep
    # the sweep's zeroed buffers out of ONE fill (each fill is a launch of its own, ~4.5 us at any size): the gradient
    # scale's words, running sums and recurrent gradients (the sweep always accumulates), per-row partials of the alpha
    # weights, the h-projection gradients of each branch padded to all rows
    Bp = (Bt + 31) // 32 * 32
    f32 = torch.float32
    gs_z, dG1_sum_p, rec, dq2, dw_rows = cap._zeros_many(((4,), f32), ((Bp, 4 * H), f32), ((7, Bt, H), f32),
                                                         ((2, T, Bt, A), f32), ((Bt, A), f32))
    gs = None
    if getattr(cap, 'grad_scaling', True):
        gs = gs_z
        srcs = [c for _, c in list(sparse1) + list(sparse2)]
        srcs += [x.contiguous() for x in (d_fc_feats1, d_cpt_feats1, d_cpt_feats2) if x is not None]
        dense = [d.abs().amax().reshape(1) for d in (d1, d2) if d is not None]
        if len(srcs) + (1 if dense else 0) > _lib_scale_max():
            dense += [x.abs().amax().reshape(1) for x in srcs[_lib_scale_max() - 1:]]
            srcs = srcs[:_lib_scale_max() - 1]
        if dense:
            srcs.append(torch.cat(dense))
        ops.grad_scale(srcs, gs)
        d_fc_feats1 = d_fc_feats1 * gs[0] if d_fc_feats1 is not None else None
        d_cpt_feats1 = d_cpt_feats1 * gs[0] if d_cpt_feats1 is not None else None
        d_cpt_feats2 = d_cpt_feats2 * gs[0] if d_cpt_feats2 is not None else None
    scale = gs[0:1] if gs is not None else None

    def bucket_done(b):
        """Every gradient of bucket b (dp.GradSink.STARTS) is enqueued: its bias sums go out, then - under a sink - the
        bucket is unscaled in one launch and its all-reduce starts."""
        ops.colsum_multi(pending_sums)
        del pending_sums[:]
        if sink is not None:
            sink.ready(b, unscale=gs[1] if gs is not None else None)

    # ---- classifier + log-softmax over all T*Bt rows (time-major)
    Vp = _pad32(V)
    idle1, idle2 = d1 is None and not sparse1, d2 is None and not sparse2
    dlogits = zeros(TB, Vp) if (ragged or idle1 or idle2) else new(TB, Vp)
    if S.lazy is not None:                     # d1 / d2 arrived as [B,T] coefficients of the target columns (DecodePairFn)
        raw, pm, ps = S.lazy[:3]
        if not idle1:
            ops.logsoftmax_bwd_raw(raw, V, Bt * V, B1, T1, V, pm, ps, Bt, list(sparse1), dlogits, scale=scale,
                                   out_step_rows=Bt)
        if not idle2:
            ops.logsoftmax_bwd_raw(raw[0, B1:], V, Bt * V, B2, T2, V, pm[0, B1:], ps[0, B1:], Bt, list(sparse2),
                                   dlogits[B1:], scale=scale, out_step_rows=Bt)
    else:
        if not idle1:
            ops.logsoftmax_bwd_sparse(d1, S.logp1, list(sparse1), dlogits, B1 * T1, V, remap_T=T1, scale=scale,
                                      out_step_rows=Bt)
        if not idle2:
            ops.logsoftmax_bwd_sparse(d2, S.logp2, list(sparse2), dlogits[B1:], B2 * T2, V, remap_T=T2, scale=scale,
                                      out_step_rows=Bt)
    Wc = p['classifier.weight']
    hdrop_tb = (S.hdrop if S.hdrop is not None else S.h2[1:]).reshape(TB, H)
    dhd = new(TB, H)
    Vm = V // 32 * 32
    if Vp != V and TB >= 8192:
        Wc_k = zeros(Vp, H)
        Wc_k[:V].copy_(Wc)
        ops.gemm_bwd([nn([(dlogits, Wc_k)], dhd)], NN)
    elif Vm != V and Vm >= 4096:
        with _weights_scope(cap):
            ops.gemm_bwd([nn([(dlogits[:, :Vm], Wc[:Vm])], dhd)], NN)
        ops.gemm_bwd([nn([(dlogits[:, Vm:], Wc[Vm:])], dhd, True)], NN)
    else:
        with _weights_scope(cap):
            ops.gemm_bwd([nn([(dlogits, Wc)], dhd)], NN)
    if V % 4 == 0:
        dWc = gout('classifier.weight', V, H)
        ops.gemm_bwd([ops.gemm_problem([(dlogits[:, :V], hdrop_tb)], dWc, TN)], TN)
    else:
        dWp = new(Vp, H)
        ops.gemm_bwd([ops.gemm_problem([(dlogits, hdrop_tb)], dWp, TN)], TN)
        gout('classifier.weight', V, H).copy_(dWp[:V])
    if Vp != V:
        db = new(Vp)
        ops.colsum(dlogits, db)
        gout('classifier.bias', V).copy_(db[:V])
    else:
        ops.colsum(dlogits, gout('classifier.bias', V))
    bucket_done(3)                      # classifier: its exchange runs behind the whole reverse sweep
    if S.hdrop is not None:
        ops.relu_mask_bwd(dhd, None, dhd, keep_mask=S.out_masks.view(TB, H), scale=S.out_scale)

    Wih1 = p['att_lstm.weight_ih']
    dG1, dG2 = stack(T, Bt, 4 * H), stack(T, Bt, 4 * H)
    d_feat_all = stack(T, Bt, E)
    # zeroed: running sums and recurrent gradients (the sweep always accumulates), per-row partials of the alpha weights,
    # and the h-projection gradients of each branch padded to all rows (their dW contractions run over all T*Bt rows)
    # (rows padded to a multiple of 32 with zeros: its two dW contractions - over the B1 + B2 rows, e.g. 208 - then run on
    # the split-f16 kernels, whose contraction length is a multiple of 32; 190 us on the fp32 tiles otherwise)
    dG1_sum = dG1_sum_p[:Bt]
    dh1 = rec[6]
    dqa, dqw = dq2.unbind(0)
    dwc_rows, dws_rows = dw_rows[:B1], dw_rows[B1:]
    de_c, de_s = new(T1, B1, R), new(T2, B2, Mw)
    dP_att, dV_att = new(B1, R, A), new(B1, R, E)
    dP_w, dV_w = new(B2, Mw, A), new(B2, Mw, Wd)

    def bwd_plan(a1, a2):
        bp = _lib.StepBwdPlan()
        bp.H, bp.E, bp.A, bp.W, bp.R, bp.Mw = H, E, A, Wd, R, Mw
        for field, key in (('Wih1', 'att_lstm.weight_ih'), ('Whh1', 'att_lstm.weight_hh'),
                           ('Wih2', 'lang_lstm.weight_ih'), ('Whh2', 'lang_lstm.weight_hh'),
                           ('W_h2att', 'attention.cont_att.h2att.weight'),
                           ('w_alpha_c', 'attention.cont_att.att_alpha.weight'),
                           ('W_h2word', 'attention.senti_att.h2word.weight'),
                           ('w_alpha_s', 'attention.senti_att.word_alpha.weight')):
            setattr(bp, field, p[key].data_ptr())
        if a1:
            bp.att_p, bp.att_e = P1.att_p3.data_ptr(), P1.att_e3.data_ptr()
            bp.alpha_c_ld, bp.dwc_rows = S.aC.stride(0), dwc_rows.data_ptr()
        if a2:
            bp.words_p, bp.words_e, bp.label_w = P2.words_p3.data_ptr(), P2.words_e3.data_ptr(), P2.label_w.data_ptr()
            bp.alpha_s_ld, bp.dws_rows = S.aS.stride(0), dws_rows.data_ptr()
        bp.pair_rows_c = B1 if (a1 and a2) else 0
        bp.first = 0
        skws = ops.splitk_ws(cap._dev)
        bp.splitk_ws, bp.splitk_ws_floats = skws.data_ptr(), skws.numel()
        return bp

    plans = {(True, True): bwd_plan(True, True)}
    if ragged:
        plans[(True, False)] = bwd_plan(True, False)
        plans[(False, True)] = bwd_plan(False, True)
    F = _Fill(Bt)
    rows_at = lambda x, r0, w: x.data_ptr() + 4 * r0 * w           # noqa: E731  ([Bt, w] buffers)
    with _weights_scope(cap):
        for t in range(T - 1, -1, -1):
            a1, a2 = t < T1, t < T2
            r0, r1 = (0 if a1 else B1), (Bt if a2 else B1)
            bp = plans[(a1, a2)]
            cur, nxt = t & 1, (t + 1) & 1
            bp.rows, bp.last = r1 - r0, int(t == 0)
            bp.g1, bp.c1_prev, bp.c1 = F.at(S.g1, t, r0, 4 * H), F.at(S.c1, t, r0, H), F.at(S.c1, t + 1, r0, H)
            bp.g2, bp.c2_prev, bp.c2 = F.at(S.g2, t, r0, 4 * H), F.at(S.c2, t, r0, H), F.at(S.c2, t + 1, r0, H)
            bp.dhd, bp.dG1, bp.dG2 = dhd.data_ptr() + 4 * (t * Bt + r0) * H, F.at(dG1, t, r0, 4 * H), F.at(dG2, t, r0, 4 * H)
            bp.d_feat = F.at(d_feat_all, t, r0, E)
            bp.dG1_sum, bp.dh1 = rows_at(dG1_sum, r0, 4 * H), rows_at(dh1, r0, H)
            bp.dh2_rec, bp.dh1_rec = rows_at(rec[0], r0, H), rows_at(rec[1], r0, H)
            bp.dc1_in, bp.dc1_out = rows_at(rec[2 + nxt], r0, H), rows_at(rec[2 + cur], r0, H)
            bp.dc2_in, bp.dc2_out = rows_at(rec[4 + nxt], r0, H), rows_at(rec[4 + cur], r0, H)
            if a1:
                bp.qa, bp.v = S.qa.data_ptr() + 4 * t * B1 * A, F.at(S.feat, t, 0, E)
                bp.alpha_c = S.aC.data_ptr() + 4 * t * R
                bp.dqa, bp.de_c = F.at(dqa, t, 0, A), de_c.data_ptr() + 4 * t * B1 * R
            if a2:
                bp.qw, bp.s = S.qw.data_ptr() + 4 * t * B2 * A, F.at(S.feat, t, B1, E)
                bp.alpha_s = S.aS.data_ptr() + 4 * t * Mw
                bp.dqw, bp.de_s = F.at(dqw, t, B1, A), de_s.data_ptr() + 4 * t * B2 * Mw
            ops.step_bwd(bp)

    dG1f, dG2f = dG1.view(TB, 4 * H), dG2.view(TB, 4 * H)
    h1_prev, h1_cur = S.h1[:T].reshape(TB, H), S.h1[1:].reshape(TB, H)
    h2_prev = S.h2[:T].reshape(TB, H)
    feat_tb = S.feat.view(TB, E)
    dqaf, dqwf = dqa.view(TB, A), dqw.view(TB, A)
    emb = p['word_embed.0.weight']

    # ---- bucket 2: lang-LSTM and the attention's own parameters - one contraction over all T*Bt rows each
    gW2 = gout('lang_lstm.weight_ih', 4 * H, E + H)
    ops.gemm_bwd([ops.gemm_problem([(dG2f, feat_tb)], gW2[:, 0:E], TN),
                  ops.gemm_problem([(dG2f, h1_cur)], gW2[:, E:], TN),
                  ops.gemm_problem([(dG2f, h2_prev)], gout('lang_lstm.weight_hh', 4 * H, H), TN)], TN)
    csum(dG2f, 'lang_lstm.bias_ih', 'lang_lstm.bias_hh')
    tn(dqaf, h1_cur, 'attention.cont_att.h2att.weight')
    csum(dqaf, 'attention.cont_att.h2att.bias')
    csum(dwc_rows, 'attention.cont_att.att_alpha.weight')
    zero_grad_of('attention.cont_att.att_alpha.bias')          # softmax is shift invariant
    tn(dqwf, h1_cur, 'attention.senti_att.h2word.weight')
    csum(dqwf, 'attention.senti_att.h2word.bias')
    csum(dws_rows, 'attention.senti_att.word_alpha.weight')
    zero_grad_of('attention.senti_att.word_alpha.bias')
    d_label_w = None
    if P2.label_w is not None:          # label2word(label_e) enters every step's score: d label_w = sum_t dqw[t]
        d_label_w_all = new(Bt * A)
        ops.colsum(dqw.view(T, Bt * A), d_label_w_all)
        d_label_w = d_label_w_all.view(Bt, A)[B1:]
        tn(d_label_w, P2.label_e, 'attention.senti_att.label2word.weight')
        csum(d_label_w, 'attention.senti_att.label2word.bias')
    bucket_done(2)

    # ---- bucket 1: att-LSTM, region embedding + projection (XE branch), sentiment-word projection (seq2seq branch)
    pad = [_const_zeros(cap, Bp - Bt, E)] if Bp > Bt else []
    fc_e_all = torch.cat([P1.fc_e, P2.fc_e] + pad)
    label_e_all = None
    if P1.label_e is not None:
        label_e_all = torch.cat([P1.label_e, P2.label_e] + ([_const_zeros(cap, Bp - Bt, Wd)] if Bp > Bt else []))
    gW1 = gout('att_lstm.weight_ih', 4 * H, H + E + Wd)
    # (problems grouped by their dY operand: isc_gemm_bwd splits a shared dY once and runs the group as one launch)
    ops.gemm_bwd([ops.gemm_problem([(dG1f, h2_prev)], gW1[:, 0:H], TN),
                  ops.gemm_problem([(dG1f, S.xt.view(TB, Wd))], gW1[:, H + E:], TN),
                  ops.gemm_problem([(dG1f, h1_prev)], gout('att_lstm.weight_hh', 4 * H, H), TN)], TN)
    once = [ops.gemm_problem([(dG1_sum_p, fc_e_all)], gW1[:, H:H + E], TN)]
    if label_e_all is not None:      # xt = relu(Emb[tok]) + label_e: the label part contracts over the rows, once per caption
        once.append(ops.gemm_problem([(dG1_sum_p, label_e_all)], gW1[:, H + E:], TN, accumulate=True))
    ops.gemm_bwd(once, TN)
    csum(dG1f, 'att_lstm.bias_ih', 'att_lstm.bias_hh')
    ops.attn_dv_from_alpha(S.aC, d_feat_all[:T1, :B1], dV_att, step_rows=Bt)
    ops.attn_dp_from_de(P1.att_p3, S.qa, p['attention.cont_att.att_alpha.weight'], de_c, dP_att)
    ops.attn_dv_from_alpha(S.aS, d_feat_all[:T2, B1:], dV_w, step_rows=Bt)
    ops.attn_dp_from_de(P2.words_p3, S.qw, p['attention.senti_att.word_alpha.weight'], de_s, dP_w, q2=P2.label_w)
    BR = B1 * R
    att_e, att_p = P1.att_e3.view(BR, E), P1.att_p3.view(BR, A)
    dzp = new(BR, A)
    ops.relu_mask_bwd(dP_att.view(BR, A), att_p, dzp)
    tn(dzp, att_e, 'att2att.0.weight')
    csum(dzp, 'att2att.0.bias')
    dVa = dV_att.view(BR, E)
    ops.gemm_bwd([nn([(dzp, p['att2att.0.weight'])], dVa, True)], NN)
    dze = new(BR, E)
    ops.relu_mask_bwd(dVa, att_e, dze, keep_mask=P1.m_att, scale=P1.sc)
    tn(dze, P1.x_att, 'att_embed.0.weight')
    csum(dze, 'att_embed.0.bias')
    BM = B2 * Mw
    w_e, w_p = P2.words_e3.view(BM, Wd), P2.words_p3.view(BM, A)
    dzw = new(BM, A)
    ops.relu_mask_bwd(dP_w.view(BM, A), w_p, dzw)
    tn(dzw, w_e, 'senti2att.0.weight')
    csum(dzw, 'senti2att.0.bias')
    bucket_done(1)

    # ---- bucket 0: the word / label embeddings, fc_embed, cpt2fc
    d_fc_e = new(Bt, E)
    d_label_e = new(Bt, Wd) if label_e_all is not None else None
    dxt = new(TB, Wd)
    probs = [nn([(dG1_sum, Wih1[:, H:H + E])], d_fc_e), nn([(dG1f, Wih1[:, H + E:])], dxt)]
    if d_label_e is not None:
        probs.append(nn([(dG1_sum, Wih1[:, H + E:])], d_label_e))
    with _weights_scope(cap):
        ops.gemm_bwd(probs, NN)
    dEmb = gout('word_embed.0.weight', V, Wd)
    if sink is None:
        dEmb.zero_()
    ops.embed_relu_bwd(emb, S.tok.view(-1), dxt, dEmb, TB, skip_id=cap.pad_id)
    if d_label_w is not None:
        ops.gemm_bwd([nn([(d_label_w, p['attention.senti_att.label2word.weight'])], d_label_e[B1:], True)], NN)
    if d_label_e is not None:
        dL = gout('senti_label_embed.0.weight', p['senti_label_embed.0.weight'].shape[0], Wd)
        if sink is None:
            dL.zero_()
        for P, lo, hi in ((P1, 0, B1), (P2, B1, Bt)):
            ops.embed_relu_bwd(p['senti_label_embed.0.weight'], P.label_ids, d_label_e[lo:hi], dL, hi - lo,
                               keep_mask=P.m_label, mask_scale=P.sc)
    dVw = dV_w.view(BM, Wd)
    ops.gemm_bwd([nn([(dzw, p['senti2att.0.weight'])], dVw, True)], NN)
    ops.embed_relu_bwd(emb, P2.sw_ids, dVw, dEmb, BM, pad_first=Mw, pad_id=cap.pad_id,
                       keep_mask=P2.m_words, mask_scale=P2.sc, skip_id=cap.pad_id)
    # fc_embed (XE rows) and cpt2fc (XE rows: only through the cpt_feats attribute; seq2seq rows: fc_e := dropout(cpt))
    dzf = new(B1, E)
    ops.relu_mask_bwd(d_fc_e[:B1], P1.fc_e, dzf, keep_mask=P1.m_fc, scale=P1.sc)
    if d_fc_feats1 is not None:
        extra = new(B1, E)
        ops.relu_mask_bwd(d_fc_feats1.contiguous(), cap_pre(P1, 'fc'), extra)
        dzf = dzf + extra
    tn(dzf, P1.x_fc, 'fc_embed.0.weight')
    csum(dzf, 'fc_embed.0.bias')
    d_cpt = zeros(Bt, E) if d_cpt_feats1 is None else new(Bt, E)
    if d_cpt_feats1 is not None:
        ops.relu_mask_bwd(d_cpt_feats1.contiguous(), P1.cpt, d_cpt[:B1])
    ops.relu_mask_bwd(d_fc_e[B1:], P2.cpt, d_cpt[B1:], keep_mask=P2.m_cpt, scale=P2.sc)
    if d_cpt_feats2 is not None:
        extra = new(B2, E)
        ops.relu_mask_bwd(d_cpt_feats2.contiguous(), cap_pre(P2, 'cpt'), extra)
        d_cpt[B1:] += extra
    cmean_all = torch.cat([P1.cmean, P2.cmean])
    tn(d_cpt, cmean_all, 'cpt2fc.0.weight')
    csum(d_cpt, 'cpt2fc.0.bias')
    dcm = new(Bt, Wd)
    ops.gemm_bwd([nn([(d_cpt, p['cpt2fc.0.weight'])], dcm)], NN)
    C = P1.cpt_ids.shape[1]
    if P2.cpt_ids.shape[1] != C:
        raise ValueError('merged unrolls: the two calls carry different numbers of concept words')
    cpt_ids = torch.cat([P1.cpt_ids, P2.cpt_ids]).view(-1)
    # (nn.Embedding(padding_idx=pad_id): the <PAD> row never gets a gradient - every accumulation into dEmb skips it)
    ops.embed_relu_bwd(emb, cpt_ids, dcm, dEmb, Bt * C, rows_per_grad=C, scale=1.0 / C, skip_id=cap.pad_id)
    bucket_done(0)
    if sink is None and gs is not None:
        torch._foreach_mul_(list(G.values()), gs[1])
    return G


def _const_zeros(cap, rows, width):
    """A read-only block of zeros, kept per captioner (padding rows of the merged dW contractions): no fill per call."""
    cache = cap.__dict__.setdefault('_zero_blocks', {})
    key = (rows, width, str(cap._dev))
    z = cache.get(key)
    if z is None:
        z = cache[key] = torch.zeros(rows, width, dtype=torch.float32, device=cap._dev)
    return z


def _lib_scale_max():
    return 4          # ISC_SCALE_SRC_MAX (include/insenticap_hip.h)


class DecodePairFn(torch.autograd.Function):
    """(xe inputs, seq2seq inputs, *params) -> (logp_xe, cpt_feats_xe, fc_feats_xe, logp_s2s, cpt_feats_s2s)."""
    LOGP_SLOTS = (0, 3)          # output numbers of the two log-prob tensors (the criteria's sparse side channels)

    @staticmethod
    def forward(ctx, cap, xe, s2s, names, lazy, *params):
        with torch.no_grad():
            out1, out2, S = _pair_forward(cap, xe, s2s, lazy)
        ctx.cap, ctx.S, ctx.names = cap, S, names
        ctx._isc_sparse = {0: [], 3: []}          # per log-prob output (autograd.sparse_channel)
        ctx.set_materialize_grads(False)
        return out1, S.cpt_feats1.clone(), S.fc_feats1.clone(), out2, S.cpt_feats2.clone()

    @staticmethod
    def backward(ctx, d1, d_cpt1, d_fc1, d2, d_cpt2):
        cap, S = ctx.cap, ctx.S
        sp, ctx._isc_sparse = ctx._isc_sparse, {0: [], 3: []}
        with torch.no_grad():
            if S.lazy is not None:             # d1 / d2 = d log p(target) [B,T]: one column per row
                if d1 is not None:
                    sp[0], d1 = [(S.lazy[3], d1.contiguous())], None
                if d2 is not None:
                    sp[3], d2 = [(S.lazy[4], d2.contiguous())], None
            G = _pair_backward(cap, S, d1.contiguous() if d1 is not None else None,
                               d2.contiguous() if d2 is not None else None, sp[0], sp[3], d_fc1, d_cpt1, d_cpt2)
        ctx.S = None
        return (None,) * 5 + tuple(G.get(n) for n in ctx.names)


def pair_with_grad(cap, fc, att, cpt_words, captions, senti_labels, ss_prob, s_captions, s_cpt_words, s_senti_words,
                   s_senti_labels, s_ss_prob, masks, s_masks, targets=None, s_targets=None):
    """forward_xe + forward_seq2seq of one iteration through one step chain.  Returns (logp_xe, logp_s2s, cpt_feats of
    the seq2seq call); captioner.fc_feats / .cpt_feats are left as the XE call leaves them (what the domain-align loss
    reads, train_xe.py:163)."""
    names = [n for n, q in cap.named_parameters() if q.requires_grad]
    params = [q for _, q in cap.named_parameters() if q.requires_grad]
    ids1, ids2 = cap._ids(captions), cap._ids(s_captions)
    tok1, tok2 = ids1[:, :-1].contiguous(), ids2[:, :-1].contiguous()
    xe = (fc, att, cpt_words, tok1, senti_labels, float(ss_prob), masks)
    s2s = (s_cpt_words, s_senti_words, tok2, s_senti_labels, float(s_ss_prob), s_masks)
    lazy = None
    if cap.__dict__.get('_token_logprobs'):
        lazy = (ids1[:, 1:].contiguous() if targets is None else cap._ids(targets).contiguous(),
                ids2[:, 1:].contiguous() if s_targets is None else cap._ids(s_targets).contiguous())
    outs = DecodePairFn.apply(cap, xe, s2s, names, lazy, *params)
    cap.cpt_feats, cap.fc_feats = outs[1], outs[2]
    return outs[0], outs[3], outs[4]
