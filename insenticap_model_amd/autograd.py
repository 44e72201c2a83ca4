"""Training path: forward with saved activations + hand-written BPTT on the HIP kernels.

The reference relies on stock torch autograd through its per-step op graph
(train_xe.py:189-190, decoder.py:164-165).  Here one `torch.autograd.Function` covers a whole
call (prologue + T-step unroll): its forward writes every per-step activation straight into
time-stacked buffers [T(+1), B, ...], and its backward is a reverse sweep in which
  * the recurrence only runs input-gradient contractions (isc_gemm_bwd NN) and the
    pointwise / scan backward kernels, and
  * every weight gradient is ONE contraction over all T*B rows after the sweep (isc_gemm_bwd TN)
    - weights are shared across time, so nothing is lost by deferring them.
torch is used for memory, the autograd hook-up and (train mode) random numbers only.
"""
import torch

from . import ops

NN, TN = ops.NN, ops.TN


class _Saved:
    pass



def _weights_scope(cap):
    """Weights scope keyed by the parameter VALUES (Captioner._weights_key): the forward unrolls of one training
    iteration (sampled roll-out, XE, the greedy baseline in between) and its backward sweeps all see the same weights,
    so the f16 planes - forward layout and transposes - are built once per iteration and stream, not once per scope
    (91 -> ~40 split launches per RL iteration at B=512)."""
    return ops.h3_weights_scope(cap._dev, key=cap._weights_key())


def _pad32(v):
    return (v + 31) // 32 * 32


# ------------------------------------------------------------------------------ forward
def _train_forward(cap, mode, fc, att, cpt_words, senti_words, tokens_in, senti_labels, ss_prob, masks, lazy=None):
    """Returns (logp [B,T,V], S). tokens_in [B,T]: ground-truth inputs (column 0 = <SOS>) - or, for the sampled
    roll-out with gradients, a dict {'T', 'u' (uniforms [B,T]) or 'forced' (raw draws [B,T])}: every step then
    draws its own next token on the device (isc_rollout_finalize) while the activations the backward pass
    needs are kept, so sampling and the differentiable forward are ONE unroll (captioner.py:290-349 does the
    same inside autograd).  The draws land in S.sample = (seq, masks, raw, alive).
    lazy: None, or the [B,T] int64 ids whose log-probs the caller's criterion reads (XECriterion's targets; True for the
    sampled roll-out: its own draws): the [B,T,V] log-probs are then NOT formed - the first return value is log p(id)
    [B,T] straight from the raw logits + tile statistics (isc_gather_logp_raw; the bits the tensor would have held), and
    the backward recomputes the softmax term from them (isc_logsoftmax_bwd_raw)."""
    p = cap._p()
    P = cap._prologue(p, mode, fc, att, cpt_words, senti_words, senti_labels, masks)
    st = cap.settings
    E, A, H, Wd, V = st['feat_emb_dim'], st['att_hid_dim'], st['rnn_hid_dim'], st['word_emb_dim'], cap.vocab_size
    sampling = isinstance(tokens_in, dict)
    if sampling:
        B, T = P.B, tokens_in['T']
    else:
        B, T = tokens_in.shape
    has_c, has_s = P.att_e3 is not None, P.words_e3 is not None
    S = _Saved()
    S.p, S.P, S.mode, S.B, S.T = p, P, mode, B, T
    new, zeros = cap._new, cap._zeros
    # Ragged unroll (Captioner.row_counts): the batch is sorted by caption length (the reference's collates,
    # dataloader.py:17,37,68,124), so the rows still inside their caption at step t are the prefix [0, counts[t]) - step t
    # runs on those rows only.  The criteria never read a position behind a caption's end (XECriterion's mask,
    # captioner.py:431-436), and its gradient is exactly zero: same loss, same gradients.  The [T,B] layout stays; what a
    # skipped row would have written must read as ZERO (gradients) or at least finite (activations next to a zero
    # gradient in the contractions over all T*B rows) - every buffer such a row belongs to comes out of a fill.
    counts = cap.__dict__.get('_row_counts')
    if counts is not None and (sampling or lazy is None or (cap.training and ss_prob > 0.0) or len(counts) != T
                               or counts[0] != B or ops.TIMER.armed):
        counts = None
    S.row_counts = counts
    S.h1, S.c1, S.h2, S.c2 = zeros(4, T + 1, B, H).unbind(0)             # slot 0 = initial zero state (one fill)
    S.g1, S.g2 = new(T, B, 4 * H), new(T, B, 4 * H)                      # (read back row by row, by the same step only)
    if counts is not None:
        new = zeros
    S.xt, S.tok = cap._new(T, B, Wd), torch.empty(T, B, dtype=torch.int64, device=cap._dev)   # (every row written)
    if has_c:
        S.qa, S.v, S.aC = new(T, B, A), new(T, B, E), new(B, T, P.R)
    if has_s:
        S.qw, S.s, S.aS = new(T, B, A), new(T, B, E), new(B, T, P.Mw)
    if has_c and has_s:
        S.z, S.f, S.bG = new(T, B, A), new(T, B, E), new(B, T)
    mask_for = cap._mask_source(masks)
    S.out_masks, S.out_scale = [], 1.0
    S.hdrop = None
    n_tile = (V + 127) // 128
    # per-step tile statistics [T, B, n_tile]: the logits become log-probs in ONE launch after the last step
    # (isc_logsoftmax_apply_steps) - in every form of the unroll: the draws of the sampled and the scheduled-sampling
    # forms read a step's RAW logits with its statistics (20 + 20 + 20 normalising launches per RL iteration -> 3)
    pm_st, ps_st = cap._new(T, B, n_tile), cap._new(T, B, n_tile)
    pi_st = cap._new(T, B, n_tile, dtype=torch.int32)
    fed_known_ = not sampling and not (cap.training and ss_prob > 0.0)
    out = None if (lazy is not None and fed_known_) else new(B, T, V)     # (lazy + every fed token known: raw logits only)
    emb = p['word_embed.0.weight']
    plan = cap._make_plan(p, P, B)
    if sampling:
        from ._lib import RolloutStep
        seq = zeros(B, T, dtype=torch.int64)
        seq_lp, seq_masks, raw = zeros(B, T), zeros(B, T), zeros(B, T, dtype=torch.int64)
        unfinished = torch.ones(B, dtype=torch.int32, device=cap._dev)
        alive = zeros(T + 1, dtype=torch.int32)
        alive[0:1].fill_(B)                     # a fill kernel (a scalar assignment would be a pageable H2D copy)
        forced, sample_u = tokens_in.get('forced'), tokens_in.get('u')
        rs = RolloutStep()
        rs.B, rs.V, rs.T, rs.n_tile, rs.W = B, V, T, n_tile, Wd
        rs.part_max, rs.part_sum, rs.part_idx = pm_st[0].data_ptr(), ps_st[0].data_ptr(), pi_st[0].data_ptr()
        rs.ld_logits = out.stride(0)          # (sampling: `out` always exists - every step's draw reads its raw logits)
        rs.forced, rs.sample_u, rs.eos_id = ops.ptr(forced), ops.ptr(sample_u), cap.eos_id
        rs.seq, rs.seq_logprobs, rs.seq_masks = seq.data_ptr(), seq_lp.data_ptr(), seq_masks.data_ptr()
        rs.unfinished, rs.alive, rs.raw_tokens = unfinished.data_ptr(), alive.data_ptr(), raw.data_ptr()
        rs.emb, rs.xt_add = emb.data_ptr(), None
        S.tok[0] = cap.sos_id
        ops.embed_relu_fwd(emb, S.tok[0], S.xt[0])
    mask_for.predraw('out', T, B, H)
    fed_known = not sampling and not (cap.training and ss_prob > 0.0)
    pm_all = ps_all = None
    if fed_known:                                         # every fed token is known up front: one copy, one gather
        S.tok.copy_(tokens_in.t())
        ops.embed_relu_fwd(emb, S.tok.view(-1), S.xt.view(T * B, Wd))
        # ... and nothing reads a step's logits before the unroll ends: the classifier runs ONCE over all steps'
        # h_lang [T*B, H] afterwards (at B = 128: 20 skinny launches of 24 us -> one [2560 x V] launch), and one
        # more launch turns its [T,B,V] logits into the [B,T,V] log-probs
        pm_all, ps_all, pi_all = pm_st, ps_st, pi_st
    # the weights are fixed for the whole unroll: their f16 planes are built once (few-row launches then take the
    # one-launch skinny split-f16 kernels instead of split-K + reduce pairs)
    # scheduled sampling: the uniforms of every step (select, draw) in ONE launch in front of the loop (was one per step)
    u_all = torch.rand(max(T - 1, 1), 2, B, device=cap._dev) if (not sampling and not fed_known) else None
    with _weights_scope(cap):
        for t in range(T):
            if sampling:
                if t >= 1:
                    S.tok[t].copy_(seq[:, t - 1])             # it * unfinished, written by the previous finalize
            elif not fed_known:
                if t >= 1:                                    # scheduled sampling, captioner.py:219-228:
                    u = u_all[t - 1]                          # select + draw on the device, no host test
                    ops.sched_sample(out[:, t - 1], pm_st[t - 1], ps_st[t - 1], pi_st[t - 1], u[0], u[1], ss_prob,
                                     tokens_in[:, t], S.tok[t], raw=True)
                else:
                    S.tok[t] = tokens_in[:, t]
                ops.embed_relu_fwd(emb, S.tok[t], S.xt[t])        # plain relu(Emb[tok]); label term is in P.pre1
            om, osc = mask_for('out%d' % t, B, H)
            save = {'g1': S.g1[t], 'g2': S.g2[t]}
            if om is not None:
                if S.hdrop is None:
                    S.hdrop = new(T, B, H)
                save['hdrop'] = S.hdrop[t]
                S.out_scale = osc
            S.out_masks.append(om)
            ws = {'_plan': plan} if pm_all is not None else {'pmax': pm_st[t], 'psum': ps_st[t], 'pidx': pi_st[t],
                                                              '_plan': plan}
            if has_c:
                ws['qa'], ws['v'] = S.qa[t], S.v[t]
            if has_s:
                ws['qw'], ws['s'] = S.qw[t], S.s[t]
            if has_c and has_s:
                ws['z'], ws['f'] = S.z[t], S.f[t]
            logits = out[:, t] if pm_all is None else None
            if counts is not None:
                plan.rows = counts[t]
            cap._step(p, P, ws, S.xt[t], (S.h1[t], S.h2[t]), (S.c1[t], S.c2[t]),
                      (S.h1[t + 1], S.h2[t + 1]), (S.c1[t + 1], S.c2[t + 1]),
                      S.aC[:, t] if has_c else None, S.aS[:, t] if has_s else None,
                      S.bG[:, t:t + 1] if (has_c and has_s) else None, logits, om, osc, save=save,
                      normalize=False)
            if sampling:                                      # draw on the raw logits (normalised after the loop)
                rs.t, rs.logits = t, logits.data_ptr()
                rs.part_max, rs.part_sum, rs.part_idx = pm_st[t].data_ptr(), ps_st[t].data_ptr(), pi_st[t].data_ptr()
                rs.xt_next = S.xt[t + 1].data_ptr() if t + 1 < T else None
                ops.rollout_finalize(rs)
        S.lazy = None
        S.packed = None
        if pm_all is not None and counts is not None and lazy is not None and sum(counts) < T * B:
            # ragged: the classifier and everything behind it see the N = sum(counts) rows inside their captions only -
            # h_lang gathered into packed time-major order (row n = (t, b), b < counts[t]), the statistics, raw logits,
            # log p(target) and - in the backward - d logits, d h and the classifier's gradients over N rows
            # (the row indices are built on the device from the T counts: a host-built index would travel through a
            # freshly pinned buffer every iteration; N is known here, so nonzero needs no read-back)
            N = int(sum(counts))
            cnt = ops.to_device(torch.tensor(counts, dtype=torch.int64), cap._dev)
            live = torch.arange(B, device=cap._dev).unsqueeze(0) < cnt.unsqueeze(1)              # [T,B]
            i_tb = torch.nonzero_static(live.reshape(-1), size=N).reshape(-1)                    # t*B + b, time-major
            idx = torch.stack([i_tb, (i_tb % B) * T + i_tb // B])
            # (N rounded up to whole tiles / k-blocks of the split-f16 kernels - the row count is the contraction length of
            # the classifier's dW; the pad rows are zero in h_lang and in d logits)
            Np = (N + 255) // 256 * 256
            hs = zeros(Np, H)
            torch.index_select((S.hdrop if S.hdrop is not None else S.h2[1:]).reshape(T * B, H), 0, idx[0], out=hs[:N])
            pm_p, ps_p = cap._new(Np, n_tile), cap._new(Np, n_tile)
            pi_p = cap._new(Np, n_tile, dtype=torch.int32)
            rawl = cap._new(Np, V)
            ops.vocab_fwd(hs, p['classifier.weight'], p['classifier.bias'], pm_p, ps_p, pi_p, rawl)
            ids_p = lazy.reshape(-1).index_select(0, idx[1]).contiguous()
            tlp_p = cap._new(N)
            ops.gather_logp_raw(rawl, V, 0, N, 1, V, pm_p, ps_p, N, ids_p, tlp_p)
            tlp = zeros(B * T).index_copy_(0, idx[1], tlp_p).view(B, T)
            S.packed = dict(N=N, Np=Np, idx_tb=idx[0], idx_bt=idx[1], hs=hs, raw=rawl, pm=pm_p, ps=ps_p, tlp=tlp)
            del rawl
        elif pm_all is not None:
            hs = S.hdrop if S.hdrop is not None else S.h2[1:]
            rawl = cap._new(T, B, V)                           # (every row is written: no fill in the ragged form)
            ops.vocab_fwd(hs.reshape(T * B, H), p['classifier.weight'], p['classifier.bias'], pm_all.view(T * B, n_tile),
                          ps_all.view(T * B, n_tile), pi_all.view(T * B, n_tile), rawl.view(T * B, V))
            if lazy is not None:
                S.lazy = (rawl, V, B * V)                     # raw logits time-major: row (b,t) at b*V + t*B*V
            else:
                ops.logsoftmax_apply_steps(out, pm_all, ps_all, src_tbv=rawl)
            del rawl
        elif lazy is not None:
            S.lazy = (out, out.stride(0), out.stride(1))      # raw logits [B,T,V]: row (b,t) at b*T*V + t*V
        else:
            ops.logsoftmax_apply_steps(out, pm_st, ps_st)     # in place: [B,T,V] raw logits -> log-probs
    if sampling:
        S.sample = (seq, seq_masks, raw, alive)
    cap._set_weights(S.aC if has_c else None, S.aS if has_s else None,
                     S.bG if (has_c and has_s) else None, T)
    if S.packed is not None:
        S.logp, S.lazy_ids, S.lazy_live = None, lazy.contiguous(), None
        return S.packed.pop('tlp'), S
    if S.lazy is not None:
        S.pm, S.ps = pm_st, ps_st
        S.lazy_live = None
        if sampling:      # log p(drawn token) * live (captioner.py:336; zero after the reference's early break)
            S.lazy_ids = raw
            S.lazy_live = (alive[:T] > 0).to(torch.float32)
        else:
            S.lazy_ids = lazy.contiguous()
        tlp = new(B, T)
        rl, ld_b, ld_t = S.lazy
        ops.gather_logp_raw(rl, ld_b, ld_t, B, T, V, pm_st, ps_st, B, S.lazy_ids, tlp, live=S.lazy_live)
        S.logp = None
        return tlp, S
    S.logp = out
    return out, S


# ------------------------------------------------------------------------------ backward
def _backward(cap, S, dlogp, d_fc_feats, d_cpt_feats, sparse=()):
    """Returns {param name: gradient}.  The gradient of the log-probs arrives as `dlogp` [B,T,V] (contiguous; None when
    every consumer handed its part over sparse) plus `sparse` = [(ids [B,T] int64, coef [B,T] fp32)]: coef at column ids
    of each row (XELossFn / GatherLogpFn below).  Optional gradients of the `fc_feats` (pre-dropout) and `cpt_feats`
    attributes.
    Gradient scale: the sweep is linear in what enters it, so everything entering is multiplied by a power of two S
    (isc_grad_scale: the largest entering |gradient| -> 2^-4..2^-3) and the parameter gradients by 1/S at the end -
    both exact - which keeps the f16 planes of the split-f16 contractions in their normal range."""
    p, P, B, T = S.p, S.P, S.B, S.T
    st = cap.settings
    E, A, H, Wd, V = st['feat_emb_dim'], st['att_hid_dim'], st['rnn_hid_dim'], st['word_emb_dim'], cap.vocab_size
    new, zeros = cap._new, cap._zeros
    has_c, has_s = P.att_e3 is not None, P.words_e3 is not None
    gate = has_c and has_s
    G = {}
    TB = T * B

    def nn(segs, out, acc=False):
        return ops.gemm_problem(segs, out, NN, accumulate=acc)

    def tn(a, w, shape=None):
        out = new(a.shape[1], w.shape[1])
        ops.gemm_bwd([ops.gemm_problem([(a, w)], out, TN)], TN)
        return out

    pending_sums = []      # (x, outs): every bias gradient of the sweep goes out in one isc_colsum_multi at the end

    def csum(x, copies=1):
        """Column sum of x, deferred; `copies` > 1: that many identical results (tied biases) from one reduction."""
        outs = [new(x.shape[1]) for _ in range(copies)]
        pending_sums.append((x, outs, False))
        return outs[0] if copies == 1 else outs

    # ---- classifier + log-softmax: outside the recurrence, all T*B rows at once (time-major rows)
    Vp = _pad32(V)
    pk = getattr(S, 'packed', None)
    dlogits = new(TB if pk is None else pk['Np'], Vp)
    # everything the sweep wants zeroed comes out of ONE fill (each fill is a launch of its own, ~4.5 us at any size)
    n_lab = p['senti_label_embed.0.weight'].shape[0] if P.label_e is not None else 1
    f32 = torch.float32
    gs_z, dG1_sum, zero_a, zero_b, dL_z, dEmb = cap._zeros_many(((4,), f32), ((B, 4 * H), f32), ((1,), f32), ((1,), f32),
                                                                ((n_lab, Wd), f32), ((V, Wd), f32))
    gs = None
    if getattr(cap, 'grad_scaling', True):
        gs = gs_z
        srcs = [c for _, c in sparse] + [d_fc_feats.contiguous() if d_fc_feats is not None else None,
                                         d_cpt_feats.contiguous() if d_cpt_feats is not None else None]
        if dlogp is not None:
            srcs.append(dlogp.abs().amax().reshape(1))        # (a caller-defined dense loss: one extra pass)
        ops.grad_scale(srcs, gs)
        if d_fc_feats is not None:
            d_fc_feats = d_fc_feats * gs[0]
        if d_cpt_feats is not None:
            d_cpt_feats = d_cpt_feats * gs[0]
    if pk is not None:
        # ragged, packed classifier block (see _train_forward): d logits over the N rows inside their captions
        N = pk['N']
        sp = [(i.reshape(-1).index_select(0, pk['idx_bt']).contiguous(), c.reshape(-1).index_select(0, pk['idx_bt']).contiguous())
              for i, c in sparse]
        if sp:
            ops.logsoftmax_bwd_raw(pk['raw'], V, 0, N, 1, V, pk['pm'], pk['ps'], N, sp, dlogits,
                                   scale=gs[0:1] if gs is not None else None, out_step_rows=N)
            if pk['Np'] > N:
                dlogits[N:].zero_()
        else:
            dlogits.zero_()
    elif dlogp is None and not sparse:
        dlogits.zero_()
    elif getattr(S, 'lazy', None) is not None:               # the log-probs were never formed: softmax from raw logits + stats
        rl, ld_b, ld_t = S.lazy
        ops.logsoftmax_bwd_raw(rl, ld_b, ld_t, B, T, V, S.pm, S.ps, B, list(sparse), dlogits,
                               scale=gs[0:1] if gs is not None else None, out_step_rows=B)
    else:
        ops.logsoftmax_bwd_sparse(dlogp, S.logp, list(sparse), dlogits, B * T, V, remap_T=T,
                                  scale=gs[0:1] if gs is not None else None)
    Wc = p['classifier.weight']
    hdrop_tb = (S.hdrop if S.hdrop is not None else S.h2[1:]).reshape(TB, H) if pk is None else pk['hs']
    n_rows = TB if pk is None else pk['Np']
    dhd = new(n_rows, H)
    # d h = d logits . W_c contracts over the vocabulary; the split-f16 kernels want a multiple of 32
    Vm = V // 32 * 32
    if Vp != V and n_rows >= 8192:
        # d logits is already zero-padded to Vp columns: give W_c the matching zero rows (a 20 MB copy; worth it from
        # ~100 GFLOP on, where the large kernels run the contraction)
        # NOT inside the weights scope: a scope keeps (pointer, planes) of every W operand it sees and the optimizer
        # re-splits them all after its step - from the pointer.  This copy dies with the call; its planes are built in
        # the workspace (one split launch).
        Wc_k = zeros(Vp, H)
        Wc_k[:V].copy_(Wc)
        ops.gemm_bwd([nn([(dlogits, Wc_k)], dhd)], NN)
    elif Vm != V and Vm >= 4096:
        # fewer rows (B = 128: [2560 x 512] over K = 10 000): the first Vm vocabulary rows on the K-split skinny tile,
        # the last V - Vm (< 32) on the fp32 tiles, accumulating
        with _weights_scope(cap):
            ops.gemm_bwd([nn([(dlogits[:, :Vm], Wc[:Vm])], dhd)], NN)
        ops.gemm_bwd([nn([(dlogits[:, Vm:], Wc[Vm:])], dhd, True)], NN)
    else:
        with _weights_scope(cap):
            ops.gemm_bwd([nn([(dlogits, Wc)], dhd)], NN)
    if V % 4 == 0:
        dWc = new(V, H)
        ops.gemm_bwd([ops.gemm_problem([(dlogits[:, :V], hdrop_tb)], dWc, TN)], TN)
    else:            # vocabulary not a multiple of 4: contract on the zero-padded columns, then trim
        dWp = new(Vp, H)
        ops.gemm_bwd([ops.gemm_problem([(dlogits, hdrop_tb)], dWp, TN)], TN)
        dWc = dWp[:V].contiguous()
    G['classifier.weight'] = dWc
    db = new(Vp)
    ops.colsum(dlogits, db)
    G['classifier.bias'] = db[:V].contiguous() if Vp != V else db
    if pk is not None:          # d h_lang back in [T,B] order, zero behind the captions' ends
        dhd = zeros(TB, H).index_copy_(0, pk['idx_tb'], dhd[:pk['N']])
    if S.hdrop is not None:     # nn.Dropout on h_lang (captioner.py:182)
        mk = torch.stack(S.out_masks).reshape(TB, H)
        ops.relu_mask_bwd(dhd, None, dhd, keep_mask=mk, scale=S.out_scale)
    dhd = dhd.view(T, B, H)

    Wih1, Whh1 = p['att_lstm.weight_ih'], p['att_lstm.weight_hh']
    Wih2, Whh2 = p['lang_lstm.weight_ih'], p['lang_lstm.weight_hh']
    # ragged unroll (see _train_forward): step t sweeps rows [0, counts[t]); everything a skipped row would have written
    # - its gradients, and the carries a row reads at the LAST step of its caption - is zero from a fill
    counts = getattr(S, 'row_counts', None)
    new_full = new
    if counts is not None:
        new = zeros
    if pk is not None:
        # packed: step t's gate gradients at rows [offs[t], offs[t] + counts[t]) - the sweep takes per-step pointers, so the
        # contractions over all rows behind it (both LSTM cells' dW groups, dxt, the bias sums) see N rows, no gather
        Np, offs = pk['Np'], [0]
        for m in counts:
            offs.append(offs[-1] + m)
        dG1, dG2 = new_full(Np, 4 * H), new_full(Np, 4 * H)
        if Np > pk['N']:
            dG1[pk['N']:].zero_()
            dG2[pk['N']:].zero_()
    else:
        dG1, dG2 = new(T, B, 4 * H), new(T, B, 4 * H)
    # d feat of every step is kept ([T,B,E]): where it is the scan's output gradient, dV = sum_t alpha_t x dout_t is
    # formed once after the sweep (ops.attn_dv_from_alpha) instead of a read-modify-write of [B,R,E] at every step
    d_feat_all, dh1 = new(T, B, E), new(B, H)
    dh2_rec, dh1_rec = new(B, H), new(B, H)
    dc1_rec, dc2_rec = [new(B, H), new(B, H)], [new(B, H), new(B, H)]
    if has_c:
        dqa, dv_all = new(T, B, A), (new(T, B, E) if gate else d_feat_all)
        dP_att, dV_att = new_full(B, P.R, A), new_full(B, P.R, E)
        de_c = new(T, B, P.R)           # d e of every step: dP is formed once after the sweep (ops.attn_dp_from_de)
        dwc_rows = new(B, A)
    if has_s:
        dqw, ds_all = new(T, B, A), (new(T, B, E) if gate else d_feat_all)
        dP_w, dV_w = new_full(B, P.Mw, A), new_full(B, P.Mw, Wd)
        de_s = new(T, B, P.Mw)
        dws_rows = new(B, A)
    if gate:
        dz = new(T, B, A)
        dwg_rows, dbg_rows = new(B, A), new(B)
    # reverse sweep: one library call per time step (isc_step_bwd enqueues the ~9 kernels of the step)
    from . import _lib
    bp = _lib.StepBwdPlan()
    bp.rows, bp.H, bp.E, bp.A, bp.W, bp.R, bp.Mw = B, H, E, A, Wd, P.R, P.Mw
    for field, key in (('Wih1', 'att_lstm.weight_ih'), ('Whh1', 'att_lstm.weight_hh'),
                       ('Wih2', 'lang_lstm.weight_ih'), ('Whh2', 'lang_lstm.weight_hh'),
                       ('W_h2att', 'attention.cont_att.h2att.weight'),
                       ('w_alpha_c', 'attention.cont_att.att_alpha.weight'),
                       ('W_h2word', 'attention.senti_att.h2word.weight'),
                       ('w_alpha_s', 'attention.senti_att.word_alpha.weight'),
                       ('W_gh', 'attention.h2att.weight'), ('W_gc', 'attention.cont2att.weight'),
                       ('W_gs', 'attention.senti2att.weight'), ('w_gate', 'attention.att_alpha.weight')):
        setattr(bp, field, p[key].data_ptr())
    ptr = ops.ptr
    for field, t_ in (('att_p', P.att_p3), ('att_e', P.att_e3), ('words_p', P.words_p3), ('words_e', P.words_e3),
                      ('label_w', P.label_w)):
        setattr(bp, field, ptr(t_))
    skws = ops.splitk_ws(cap._dev)
    bp.splitk_ws, bp.splitk_ws_floats = skws.data_ptr(), skws.numel()
    bp.dG1_sum, bp.dh1 = dG1_sum.data_ptr(), dh1.data_ptr()
    bp.dh2_rec, bp.dh1_rec = dh2_rec.data_ptr(), dh1_rec.data_ptr()
    if has_c:
        bp.dP_att, bp.dV_att, bp.dwc_rows = None, None, dwc_rows.data_ptr()
        bp.alpha_c_ld = S.aC.stride(0)
    if has_s:
        bp.dP_w, bp.dV_w, bp.dws_rows = None, None, dws_rows.data_ptr()
        bp.alpha_s_ld = S.aS.stride(0)
    if gate:
        bp.dwg_rows, bp.dbg_rows = dwg_rows.data_ptr(), dbg_rows.data_ptr()
        bp.beta_ld = S.bG.stride(0)
    # the weights do not change during the sweep: few-row launches take the one-launch skinny split-f16 kernel on
    # planes of W^T built once here (isc_gemm_bwd, NN layout), instead of fp32 split-K slabs + a reduce kernel per GEMM
    with _weights_scope(cap):
        for t in range(T - 1, -1, -1):
            cur, nxt = t & 1, (t + 1) & 1
            bp.first, bp.last = int(t == T - 1), int(t == 0)
            if counts is not None:
                # (no "first" step: a row's own last step is wherever its caption ends - carries and time-accumulated
                # buffers start from the fill instead)
                bp.rows, bp.first = counts[t], 0
            bp.g1, bp.c1_prev, bp.c1 = S.g1[t].data_ptr(), S.c1[t].data_ptr(), S.c1[t + 1].data_ptr()
            bp.g2, bp.c2_prev, bp.c2 = S.g2[t].data_ptr(), S.c2[t].data_ptr(), S.c2[t + 1].data_ptr()
            bp.dhd = dhd[t].data_ptr()
            bp.dG1, bp.dG2 = (dG1[t].data_ptr(), dG2[t].data_ptr()) if pk is None else \
                (dG1[offs[t]:].data_ptr(), dG2[offs[t]:].data_ptr())
            bp.d_feat = d_feat_all[t].data_ptr()
            if gate:
                bp.dv, bp.ds = dv_all[t].data_ptr(), ds_all[t].data_ptr()
            bp.dc1_in, bp.dc1_out = dc1_rec[nxt].data_ptr(), dc1_rec[cur].data_ptr()
            bp.dc2_in, bp.dc2_out = dc2_rec[nxt].data_ptr(), dc2_rec[cur].data_ptr()
            if has_c:
                bp.qa, bp.v, bp.alpha_c, bp.dqa = S.qa[t].data_ptr(), S.v[t].data_ptr(), S.aC[:, t].data_ptr(), \
                    dqa[t].data_ptr()
                bp.de_c = de_c[t].data_ptr()
            if has_s:
                bp.qw, bp.s, bp.alpha_s, bp.dqw = S.qw[t].data_ptr(), S.s[t].data_ptr(), S.aS[:, t].data_ptr(), \
                    dqw[t].data_ptr()
                bp.de_s = de_s[t].data_ptr()
            if gate:
                bp.z, bp.beta, bp.dz = S.z[t].data_ptr(), S.bG[:, t:t + 1].data_ptr(), dz[t].data_ptr()
            ops.step_bwd(bp)

    new = new_full
    if has_c:
        ops.attn_dv_from_alpha(S.aC, dv_all, dV_att)
        ops.attn_dp_from_de(P.att_p3, S.qa, p['attention.cont_att.att_alpha.weight'], de_c, dP_att)
    if has_s:
        ops.attn_dv_from_alpha(S.aS, ds_all, dV_w)
        ops.attn_dp_from_de(P.words_p3, S.qw, p['attention.senti_att.word_alpha.weight'], de_s, dP_w, q2=P.label_w)

    # ---- weight gradients: one contraction over all T*B rows each
    h1_prev, h1_cur = S.h1[:T].reshape(TB, H), S.h1[1:].reshape(TB, H)
    h2_prev = S.h2[:T].reshape(TB, H)
    feat_tb = (S.f if gate else (S.v if has_c else S.s)).view(TB, E)
    xt_tb, tok_tb, lstm_rows = S.xt.view(TB, Wd), S.tok.view(-1), TB
    h1_cur_l = h1_cur               # (the LSTM groups' copy: the attention contractions below stay on [T,B] rows)
    if pk is None:
        dG1f, dG2f = dG1.view(TB, 4 * H), dG2.view(TB, 4 * H)
    else:
        # packed gate gradients: their partners gathered to the same rows (pad rows zero / <PAD>)
        def pack(x):
            out = new(pk['Np'], x.shape[1], dtype=x.dtype)
            torch.index_select(x, 0, pk['idx_tb'], out=out[:pk['N']])
            if pk['Np'] > pk['N']:
                out[pk['N']:].zero_()
            return out
        dG1f, dG2f, lstm_rows = dG1, dG2, pk['Np']
        h1_prev, h1_cur_l, h2_prev, feat_tb, xt_tb = pack(h1_prev), pack(h1_cur), pack(h2_prev), pack(feat_tb), pack(xt_tb)
        tok_tb = pack(tok_tb.view(TB, 1)).view(-1)
        if pk['Np'] > pk['N']:
            tok_tb[pk['N']:].fill_(cap.pad_id)
    gW1 = new(4 * H, H + E + Wd)
    gW2, gwhh1, gwhh2 = new(4 * H, E + H), new(4 * H, H), new(4 * H, H)
    # problems grouped by their dY operand: isc_gemm_bwd splits a shared dY once and runs the group as one launch
    ops.gemm_bwd([ops.gemm_problem([(dG1f, h2_prev)], gW1[:, 0:H], TN),
                  ops.gemm_problem([(dG1f, xt_tb)], gW1[:, H + E:], TN),
                  ops.gemm_problem([(dG1f, h1_prev)], gwhh1, TN)], TN)
    ops.gemm_bwd([ops.gemm_problem([(dG2f, feat_tb)], gW2[:, 0:E], TN),
                  ops.gemm_problem([(dG2f, h1_cur_l)], gW2[:, E:], TN),
                  ops.gemm_problem([(dG2f, h2_prev)], gwhh2, TN)], TN)
    # xt = relu(Emb[tok]) + label_e: the per-step part contracted over T*B rows above, the label part over B here
    once = [ops.gemm_problem([(dG1_sum, P.fc_e)], gW1[:, H:H + E], TN)]
    if P.label_e is not None:
        once.append(ops.gemm_problem([(dG1_sum, P.label_e)], gW1[:, H + E:], TN, accumulate=True))
    ops.gemm_bwd(once, TN)
    G['att_lstm.weight_ih'] = gW1
    G['att_lstm.weight_hh'] = gwhh1
    G['lang_lstm.weight_ih'] = gW2
    G['lang_lstm.weight_hh'] = gwhh2
    G['att_lstm.bias_ih'], G['att_lstm.bias_hh'] = csum(dG1f, 2)
    G['lang_lstm.bias_ih'], G['lang_lstm.bias_hh'] = csum(dG2f, 2)
    # inputs of the att-LSTM: fc (step-invariant), xt = relu(Emb[tok]) + label_e
    d_fc_e = new(B, E)
    d_label_e = new(B, Wd) if P.label_e is not None else None
    dxt = new(lstm_rows, Wd)
    probs = [nn([(dG1_sum, Wih1[:, H:H + E])], d_fc_e), nn([(dG1f, Wih1[:, H + E:])], dxt)]
    if d_label_e is not None:
        probs.append(nn([(dG1_sum, Wih1[:, H + E:])], d_label_e))
    with _weights_scope(cap):      # dX over all T*B rows: split-f16 on planes of the W_ih slices' transposes
        ops.gemm_bwd(probs, NN)
    emb = p['word_embed.0.weight']
    # nn.Embedding(padding_idx=pad_id): the <PAD> row never gets a gradient - every accumulation into dEmb skips it
    ops.embed_relu_bwd(emb, tok_tb, dxt, dEmb, lstm_rows, skip_id=cap.pad_id)

    if has_c:
        dqaf = dqa.view(TB, A)
        G['attention.cont_att.h2att.weight'] = tn(dqaf, h1_cur)
        G['attention.cont_att.h2att.bias'] = csum(dqaf)
        G['attention.cont_att.att_alpha.weight'] = csum(dwc_rows).view(1, A)
        G['attention.cont_att.att_alpha.bias'] = zero_a        # softmax is shift invariant
    if has_s:
        dqwf = dqw.view(TB, A)
        G['attention.senti_att.h2word.weight'] = tn(dqwf, h1_cur)
        G['attention.senti_att.h2word.bias'] = csum(dqwf)
        G['attention.senti_att.word_alpha.weight'] = csum(dws_rows).view(1, A)
        G['attention.senti_att.word_alpha.bias'] = zero_b
        # label2word(label_e) enters every step's score: d label_w = sum_t dqw[t]
        d_label_w = new(B * A)
        ops.colsum(dqw.view(T, B * A), d_label_w)
        d_label_w = d_label_w.view(B, A)
        G['attention.senti_att.label2word.weight'] = tn(d_label_w, P.label_e)
        G['attention.senti_att.label2word.bias'] = csum(d_label_w)
        ops.gemm_bwd([nn([(d_label_w, p['attention.senti_att.label2word.weight'])], d_label_e, True)], NN)
    if gate:
        dzf = dz.view(TB, A)
        G['attention.cont2att.weight'] = tn(dzf, S.v.view(TB, E))
        G['attention.senti2att.weight'] = tn(dzf, S.s.view(TB, E))
        G['attention.h2att.weight'] = tn(dzf, h1_cur)
        G['attention.cont2att.bias'], G['attention.senti2att.bias'], G['attention.h2att.bias'] = csum(dzf, 3)
        G['attention.att_alpha.weight'] = csum(dwg_rows).view(1, A)
        G['attention.att_alpha.bias'] = csum(dbg_rows.view(B, 1))

    # ---- prologue backward
    if d_label_e is not None:
        dL = dL_z
        ops.embed_relu_bwd(p['senti_label_embed.0.weight'], P.label_ids, d_label_e, dL, B,
                           keep_mask=P.m_label, mask_scale=P.sc)
        G['senti_label_embed.0.weight'] = dL
    if has_c:
        BR = B * P.R
        att_e, att_p = P.att_e3.view(BR, E), P.att_p3.view(BR, A)
        dzp = new(BR, A)
        ops.relu_mask_bwd(dP_att.view(BR, A), att_p, dzp)
        G['att2att.0.weight'] = tn(dzp, att_e)
        G['att2att.0.bias'] = csum(dzp)
        dVa = dV_att.view(BR, E)
        ops.gemm_bwd([nn([(dzp, p['att2att.0.weight'])], dVa, True)], NN)
        dze = new(BR, E)
        ops.relu_mask_bwd(dVa, att_e, dze, keep_mask=P.m_att, scale=P.sc)
        G['att_embed.0.weight'] = tn(dze, P.x_att)
        G['att_embed.0.bias'] = csum(dze)
    if has_s:
        BM = B * P.Mw
        w_e, w_p = P.words_e3.view(BM, Wd), P.words_p3.view(BM, A)
        dzp = new(BM, A)
        ops.relu_mask_bwd(dP_w.view(BM, A), w_p, dzp)
        G['senti2att.0.weight'] = tn(dzp, w_e)
        G['senti2att.0.bias'] = csum(dzp)
        dVw = dV_w.view(BM, Wd)
        ops.gemm_bwd([nn([(dzp, p['senti2att.0.weight'])], dVw, True)], NN)
        ops.embed_relu_bwd(emb, P.sw_ids, dVw, dEmb, BM, pad_first=P.Mw, pad_id=cap.pad_id,
                           keep_mask=P.m_words, mask_scale=P.sc, skip_id=cap.pad_id)
    # fc_embed (xe / rl) and cpt2fc
    d_cpt = None
    if S.mode != 'seq2seq':
        dzf = new(B, E)
        ops.relu_mask_bwd(d_fc_e, P.fc_e, dzf, keep_mask=P.m_fc, scale=P.sc)
        if d_fc_feats is not None:
            extra = new(B, E)
            ops.relu_mask_bwd(d_fc_feats.contiguous(), cap_pre(P, 'fc'), extra)
            dzf = dzf + extra
        G['fc_embed.0.weight'] = tn(dzf, P.x_fc)
        G['fc_embed.0.bias'] = csum(dzf)
        if d_cpt_feats is not None:
            d_cpt = new(B, E)
            ops.relu_mask_bwd(d_cpt_feats.contiguous(), P.cpt, d_cpt)
    else:
        d_cpt = new(B, E)       # seq2seq: fc_e := dropout(cpt_feats) (captioner.py:250-251)
        ops.relu_mask_bwd(d_fc_e, P.cpt, d_cpt, keep_mask=P.m_cpt, scale=P.sc)
        if d_cpt_feats is not None:
            extra = new(B, E)
            ops.relu_mask_bwd(d_cpt_feats.contiguous(), cap_pre(P, 'cpt'), extra)
            d_cpt = d_cpt + extra
    if d_cpt is not None:
        G['cpt2fc.0.weight'] = tn(d_cpt, P.cmean)
        G['cpt2fc.0.bias'] = csum(d_cpt)
        dcm = new(B, Wd)
        ops.gemm_bwd([nn([(d_cpt, p['cpt2fc.0.weight'])], dcm)], NN)
        C = P.cpt_ids.shape[1]
        ops.embed_relu_bwd(emb, P.cpt_ids.view(-1), dcm, dEmb, B * C, rows_per_grad=C, scale=1.0 / C,
                           skip_id=cap.pad_id)
    G['word_embed.0.weight'] = dEmb
    ops.colsum_multi(pending_sums)
    if gs is not None:           # undo the gradient scale: x 1/S, a power of two
        torch._foreach_mul_(list(G.values()), gs[1])
    return G


def cap_pre(P, which):
    """Pre-dropout activation (the tensor the `fc_feats` / `cpt_feats` attribute exposes); equals the
    post-dropout one in sign wherever the keep-mask is 1, and its gradient ignores the mask."""
    return P.fc_pre if which == 'fc' else P.cpt_pre


class DecodeFn(torch.autograd.Function):
    """(mode, inputs, *params) -> (logp, cpt_feats[, fc_feats])."""

    @staticmethod
    def forward(ctx, cap, mode, fc, att, cpt_words, senti_words, tokens_in, senti_labels, ss_prob, masks,
                names, lazy, *params):
        with torch.no_grad():
            logp, S = _train_forward(cap, mode, fc, att, cpt_words, senti_words, tokens_in, senti_labels,
                                     ss_prob, masks, lazy)
        S.P.fc_pre = cap.fc_feats if mode != 'seq2seq' else None
        S.P.cpt_pre = cap.cpt_feats
        cap._last_sample = getattr(S, 'sample', None)
        ctx.cap, ctx.S, ctx.names = cap, S, names
        # side channel for the criteria: XELossFn / GatherLogpFn find this node as `logp.grad_fn` and append their
        # (ids, coef) pairs here in THEIR backward (which precedes this node's) instead of returning a [B,T,V] tensor
        ctx._isc_sparse = []
        ctx.set_materialize_grads(False)
        outs = [logp, cap.cpt_feats]
        if mode != 'seq2seq':
            outs.append(cap.fc_feats)
        # the attribute tensors alias internal buffers: hand autograd distinct objects
        return tuple(o if i == 0 else o.clone() for i, o in enumerate(outs))

    @staticmethod
    def backward(ctx, dlogp, d_cpt, d_fc=None):
        cap, S = ctx.cap, ctx.S
        sparse, ctx._isc_sparse = ctx._isc_sparse, []
        with torch.no_grad():
            if getattr(S, 'lazy', None) is not None or getattr(S, 'packed', None) is not None:
                # dlogp is d log p(id) [B,T]: one column per row
                if dlogp is not None:
                    coef = dlogp.contiguous() if S.lazy_live is None else (dlogp * S.lazy_live).contiguous()
                    sparse, dlogp = [(S.lazy_ids, coef)], None
            G = _backward(cap, S, dlogp.contiguous() if dlogp is not None else None, d_fc, d_cpt, sparse)
        grads = tuple(G.get(n) for n in ctx.names)
        ctx.S = None
        return (None,) * 12 + grads


def xe_with_grad(cap, mode, fc, att, cpt_words, senti_words, captions, senti_labels, ss_prob, masks, targets=None):
    names = [n for n, q in cap.named_parameters() if q.requires_grad]
    params = [q for _, q in cap.named_parameters() if q.requires_grad]
    ids = cap._ids(captions)
    tokens_in = ids[:, :-1].contiguous()
    # inside `with captioner.token_logprobs():` (the package's own training steps) the call returns log p(target) [B,T]
    # instead of the [B,T,V] log-probs - XECriterion takes either (captioner.py:427-440 reads one column per row)
    lazy = None
    if cap.__dict__.get('_token_logprobs'):
        lazy = ids[:, 1:].contiguous() if targets is None else cap._ids(targets).contiguous()
    outs = DecodeFn.apply(cap, mode, fc, att, cpt_words, senti_words, tokens_in, senti_labels, ss_prob, masks,
                          names, lazy, *params)
    logp = outs[0]
    cap.cpt_feats = outs[1]
    if mode != 'seq2seq':
        cap.fc_feats = outs[2]
    return logp


def rollout_with_grad(cap, fc, att, cpt_words, senti_words, senti_labels, T, replay, masks):
    """Sampled roll-out with REINFORCE gradients (captioner.py:290-349, sample_max=0, train mode): one unroll
    that samples each next token on the device and keeps the activations for the backward pass; the returned
    log-probs are log p(drawn token), zero after the reference's early `break`."""
    names = [n for n, q in cap.named_parameters() if q.requires_grad]
    params = [q for _, q in cap.named_parameters() if q.requires_grad]
    B = fc.shape[0]
    draws = {'T': T}
    if replay is not None:
        draws['forced'] = cap._ids(replay)
    else:
        draws['u'] = torch.rand(B, T, device=cap._dev)
    was = cap.training
    # the roll-out's [B,T,V] log-probs are never an API output (captioner.py:336 keeps log p(drawn token) only): the
    # decode node returns that column - already times `live` - from the raw logits (lazy = True; DecodeFn.backward)
    outs = DecodeFn.apply(cap, 'rl', fc, att, cpt_words, senti_words, draws, senti_labels, 0.0,
                          masks, names, True, *params)
    cap.train(was)
    lp = outs[0]
    cap.cpt_feats, cap.fc_feats = outs[1], outs[2]
    seq, seq_masks, raw, alive = cap._last_sample
    cap._last_sample = None
    return seq, lp, seq_masks


def _decode_node(logp):
    """The sparse side channel behind `logp` - (decode node, slot) - if the criterion may use it: `logp` must be that
    node's own output (a slice or a copy of it has another grad_fn), else None (dense hand-over).  A DecodeFn node has one
    log-prob output (slot None: `_isc_sparse` is a list); the merged node of two sibling unrolls (autograd_pair) has two,
    keyed by output number."""
    node = logp.grad_fn
    ch = getattr(node, '_isc_sparse', None) if node is not None else None
    if ch is None:
        return None
    if isinstance(ch, dict):
        return (node, logp.output_nr) if logp.output_nr in ch else None
    return (node, None)


def _sparse_append(chan, pair):
    node, slot = chan
    (node._isc_sparse if slot is None else node._isc_sparse[slot]).append(pair)


class GatherLogpFn(torch.autograd.Function):
    """log p(drawn token) * live (captioner.py:336, zero after the early break).  Backward: the gradient w.r.t. the
    [B,T,V] log-probs is g[b,t] * live[t] at column raw[b,t] - handed to the decode node as an (ids, coef) pair."""

    @staticmethod
    def forward(ctx, logp, raw, live, node):
        ctx.node, ctx.shape = node, logp.shape
        ctx.save_for_backward(raw, live)
        return logp.gather(2, raw.unsqueeze(2)).squeeze(2) * live

    @staticmethod
    def backward(ctx, g):
        raw, live = ctx.saved_tensors
        coef = (g * live).contiguous()
        if ctx.node is not None:
            _sparse_append(ctx.node, (raw.contiguous(), coef))
            return None, None, None, None
        d = torch.zeros(ctx.shape, dtype=coef.dtype, device=coef.device)
        d.scatter_(2, raw.unsqueeze(2), coef.unsqueeze(2))
        return d, None, None, None


class XELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, lengths_i32, node):
        out2 = torch.empty(2, dtype=torch.float32, device=pred.device)
        ops.xe_loss_fwd(pred, target, lengths_i32, out2)
        ctx.save_for_backward(target, lengths_i32, out2)
        ctx.shape, ctx.node = pred.shape, node
        return out2[0] / out2[1]

    @staticmethod
    def backward(ctx, g):
        target, lengths_i32, out2 = ctx.saved_tensors
        gout = g.reshape(1).contiguous().float()
        if ctx.node is not None:        # pred is a decode node's own output: hand (target, coef) over, no [B,T,V] tensor
            coef = torch.empty(ctx.shape[:2], dtype=torch.float32, device=target.device)
            ops.xe_loss_bwd_sparse(lengths_i32, ctx.shape[1], gout, out2, coef)
            _sparse_append(ctx.node, (target, coef))
            return None, None, None, None
        dlogp = torch.zeros(ctx.shape, dtype=torch.float32, device=target.device)
        ops.xe_loss_bwd(target, lengths_i32, gout, out2, dlogp)
        return dlogp, None, None, None


class XETokenLossFn(torch.autograd.Function):
    """XECriterion (captioner.py:427-440) on per-token log-probs tlp [B,T] = log p(target) (Captioner.token_logprobs):
    the same masked mean, the same summation order as on the [B,T,V] tensor."""

    @staticmethod
    def forward(ctx, tlp, lengths_i32):
        out2 = torch.empty(2, dtype=torch.float32, device=tlp.device)
        ops.xe_loss_tokens_fwd(tlp, lengths_i32, out2)
        ctx.save_for_backward(lengths_i32, out2)
        ctx.shape = tlp.shape
        return out2[0] / out2[1]

    @staticmethod
    def backward(ctx, g):
        lengths_i32, out2 = ctx.saved_tensors
        coef = torch.empty(ctx.shape, dtype=torch.float32, device=out2.device)
        ops.xe_loss_bwd_sparse(lengths_i32, ctx.shape[1], g.reshape(1).contiguous().float(), out2, coef)
        return coef, None


def xe_criterion_with_grad(pred, target, lengths):
    ops.require_device(pred, target)
    ln = ops.upload(lengths, torch.int32, pred.device)
    if pred.dim() == 2:                 # log p(target) [B,T] of a call inside Captioner.token_logprobs()
        return XETokenLossFn.apply(pred.contiguous(), ln)
    node = _decode_node(pred) if pred.is_contiguous() else None
    return XELossFn.apply(pred.contiguous(), target.long().contiguous(), ln, node)


class RewardLossFn(torch.autograd.Function):
    """RewardCriterion (self_critical/utils.py:169-177) as one launch each way (isc_reward_loss_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, seq_logprobs, seq_masks, reward):
        out2 = torch.empty(2, dtype=torch.float32, device=seq_logprobs.device)
        ops.reward_loss_fwd(seq_logprobs, seq_masks, reward, out2)
        ctx.save_for_backward(seq_masks, reward, out2)
        return out2[0] / out2[1]

    @staticmethod
    def backward(ctx, g):
        seq_masks, reward, out2 = ctx.saved_tensors
        d = torch.empty_like(seq_masks)
        ops.reward_loss_bwd(seq_masks, reward, g.reshape(1).contiguous().float(), out2, d)
        return d, None, None


def reward_criterion(seq_logprobs, seq_masks, reward):
    """-sum(logp * mask * reward) / sum(mask) on the device; `reward` may be a [B,T] tensor of any float dtype or
    broadcastable to it."""
    ops.require_device(seq_logprobs, seq_masks)
    lp = seq_logprobs.float().contiguous()
    mk = seq_masks.float().contiguous()
    rw = torch.as_tensor(reward, dtype=torch.float32, device=lp.device).expand_as(lp).contiguous()
    return RewardLossFn.apply(lp, mk, rw)
