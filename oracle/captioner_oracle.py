"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement of the reference caption decoder, written from the equations in
SURVEY.md section 8(a) and checked line by line against
/root/reference/models/captioner.py. It is a *functional* restatement (a dict of
tensors + free functions, no nn.Module), so it shares no structure with the
product's `Captioner`; it runs in fp32 or fp64 on the CPU and is differentiable
through torch autograd, which makes it the oracle for the backward kernels too.

Parity pin: `tests/golden/*.npz` were produced by importing the *reference itself*
(tests/golden/make_golden.py, run in the build container where /root/reference is
mounted) on the deterministic weights/inputs of `insenticap_model_amd.synth`;
`tests/test_oracle_golden.py` checks this file against every one of them.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Stochastic ops cannot be matched across devices, so they are made explicit:
  * dropout  -> optional dict of 0/1 keep-masks (scaled by 1/(1-p) here),
  * multinomial / scheduled sampling -> optional token matrices that are replayed
    ("teacher-forced replay").
"""
import math

import torch

NEG_INF = float('-inf')


def to_params(weights, dtype=torch.float32, requires_grad=False):
    """numpy dict (insenticap_model_amd.synth.make_weights) -> dict of leaf tensors."""
    p = {}
    for k, v in weights.items():
        t = torch.as_tensor(v).to(dtype).clone()
        t.requires_grad_(requires_grad)
        p[k] = t
    return p


class Ids:
    """Special token ids; captioner.py:124-129 (note: eos guard tests '<SOS>')."""

    def __init__(self, idx2word, sentiment_categories):
        self.pad = idx2word.index('<PAD>')
        self.unk = idx2word.index('<UNK>')
        self.sos = idx2word.index('<SOS>') if '<SOS>' in idx2word else self.pad
        self.eos = idx2word.index('<EOS>') if '<SOS>' in idx2word else self.pad
        self.neu = sentiment_categories.index('neutral')


def _lin(p, name, x):
    return x @ p[name + '.weight'].t() + p[name + '.bias']


def _drop(x, masks, key, p_drop):
    if masks is None or key not in masks:
        return x
    return x * masks[key].to(x.dtype) / (1.0 - p_drop)


def _embed_words(p, ids):
    # nn.Embedding + ReLU (captioner.py:133-136)
    return torch.relu(p['word_embed.0.weight'][ids])


def _lstm_cell(p, name, x, h, c):
    # nn.LSTMCell: gates = x W_ih^T + b_ih + h W_hh^T + b_hh, chunk order i,f,g,o
    g = x @ p[name + '.weight_ih'].t() + p[name + '.bias_ih'] \
        + h @ p[name + '.weight_hh'].t() + p[name + '.bias_hh']
    H = h.shape[1]
    i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def content_attention(p, h, att_e, p_att):
    """captioner.py:23-35. Returns (v_hat [B,E], alpha [B,R])."""
    h_att = _lin(p, 'attention.cont_att.h2att', h)                     # [B,A]
    e = torch.tanh(p_att + h_att.unsqueeze(1))                         # [B,R,A]
    e = (e @ p['attention.cont_att.att_alpha.weight'].t()).squeeze(-1) \
        + p['attention.cont_att.att_alpha.bias']                       # [B,R]
    alpha = torch.softmax(e, dim=-1)
    v = torch.bmm(alpha.unsqueeze(1), att_e).squeeze(1)
    return v, alpha


def senti_attention(p, h, words_e, p_words, label_e):
    """captioner.py:50-62. Returns (e_hat [B,W], alpha [B,M])."""
    hw = _lin(p, 'attention.senti_att.h2word', h)
    lw = _lin(p, 'attention.senti_att.label2word', label_e)
    e = torch.tanh(p_words + hw.unsqueeze(1) + lw.unsqueeze(1))
    e = (e @ p['attention.senti_att.word_alpha.weight'].t()).squeeze(-1) \
        + p['attention.senti_att.word_alpha.bias']
    alpha = torch.softmax(e, dim=-1)
    r = torch.bmm(alpha.unsqueeze(1), words_e).squeeze(1)
    return r, alpha


def fused_attention(p, h, att_e, p_att, words_e, p_words, label_e):
    """captioner.py:96-118. Returns (feature [B,E], dict of per-step weights)."""
    w = {}
    if att_e is None:                       # seq2seq: sentiment words only
        s, w['senti'] = senti_attention(p, h, words_e, p_words, label_e)
        return s, w
    v, w['cont'] = content_attention(p, h, att_e, p_att)
    if words_e is None:                     # xe: image regions only
        return v, w
    s, w['senti'] = senti_attention(p, h, words_e, p_words, label_e)
    z = _lin(p, 'attention.cont2att', v) + _lin(p, 'attention.senti2att', s) \
        + _lin(p, 'attention.h2att', h)
    beta = torch.sigmoid(_lin(p, 'attention.att_alpha', torch.tanh(z)))  # [B,1]
    w['gate'] = beta
    return beta * v + (1 - beta) * s, w


def init_state(p, B):
    H = p['att_lstm.weight_hh'].shape[1]
    z = p['att_lstm.weight_hh'].new_zeros
    return (z((2, B, H)), z((2, B, H)))


def step(p, it, state, fc_e, att_e=None, p_att=None, words_e=None, p_words=None,
         label_e=None, out_mask=None, p_drop=0.5):
    """captioner.py:168-186. state = (h[2,B,H], c[2,B,H]); index 0 att-LSTM, 1 lang-LSTM."""
    xt = _embed_words(p, it)
    if label_e is not None:
        xt = xt + label_e
    h, c = state
    x1 = torch.cat([h[1], fc_e, xt], dim=1)
    h_att, c_att = _lstm_cell(p, 'att_lstm', x1, h[0], c[0])
    feat, w = fused_attention(p, h_att, att_e, p_att, words_e, p_words, label_e)
    x2 = torch.cat([feat, h_att], dim=1)
    h_lang, c_lang = _lstm_cell(p, 'lang_lstm', x2, h[1], c[1])
    out = h_lang
    if out_mask is not None:
        out = out * out_mask.to(out.dtype) / (1.0 - p_drop)
    logp = torch.log_softmax(_lin(p, 'classifier', out), dim=1)
    return logp, (torch.stack([h_att, h_lang]), torch.stack([c_att, c_lang])), w


class Prologue:
    """Step-invariant tensors (captioner.py:198-214 / 247-261 / 294-315 / 357-376)."""
    fc_raw = None      # ReLU(fc_embed(fc)) before dropout  == captioner.fc_feats attr
    cpt = None         # ReLU(cpt2fc(mean ReLU(Emb[cpt])))  == captioner.cpt_feats attr
    fc_e = None        # what the att-LSTM sees
    att_e = None
    p_att = None
    words_e = None
    p_words = None
    label_e = None


def prologue(p, ids, mode, fc=None, att=None, cpt_words=None, senti_words=None,
             senti_labels=None, masks=None, p_drop=0.5):
    """mode in {'xe','seq2seq','rl','beam'}; `masks` = dropout keep-masks or None (eval)."""
    P = Prologue()
    if mode != 'seq2seq':
        B = fc.shape[0]
        P.fc_raw = torch.relu(_lin(p, 'fc_embed.0', fc))
        P.fc_e = _drop(P.fc_raw, masks, 'fc', p_drop)
        a = att.reshape(B, -1, att.shape[-1])
        a = torch.relu(_lin(p, 'att_embed.0', a))
        P.att_e = _drop(a, masks, 'att', p_drop)
        P.p_att = torch.relu(_lin(p, 'att2att.0', P.att_e))
    if cpt_words is not None:
        c = _embed_words(p, cpt_words).mean(dim=1)
        P.cpt = torch.relu(_lin(p, 'cpt2fc.0', c))
    if mode == 'seq2seq':
        P.fc_e = _drop(P.cpt, masks, 'cpt', p_drop)   # captioner.py:250-251
    if senti_words is not None:
        B = senti_words.shape[0]
        sw = torch.cat([senti_words.new_full((B, 1), ids.pad), senti_words], dim=1)
        P.words_e = _drop(_embed_words(p, sw), masks, 'words', p_drop)
        P.p_words = torch.relu(_lin(p, 'senti2att.0', P.words_e))
    if senti_labels is not None:
        le = torch.relu(p['senti_label_embed.0.weight'][senti_labels])
        P.label_e = _drop(le, masks, 'label', p_drop)
    return P


def _cat_weights(ws, key):
    xs = [w[key] for w in ws if key in w]
    return torch.cat(xs, dim=1) if xs else []


def _teacher_forced(p, P, tokens_in, masks, p_drop):
    """Unroll feeding tokens_in[:, i] at step i. Returns (logp [B,T,V], weights)."""
    B, T = tokens_in.shape
    state = init_state(p, B)
    outs, ws = [], []
    for i in range(T):
        om = masks['out'][i] if (masks is not None and 'out' in masks) else None
        logp, state, w = step(p, tokens_in[:, i], state, P.fc_e, P.att_e, P.p_att,
                              P.words_e, P.p_words, P.label_e, om, p_drop)
        outs.append(logp)
        ws.append(w)
    weights = tuple(_cat_weights(ws, k) for k in ('cont', 'senti', 'gate'))
    return torch.stack(outs, dim=1), weights


def forward_xe(p, ids, fc, att, cpt_words, captions, senti_labels, masks=None,
               fed_tokens=None, p_drop=0.5):
    """captioner.py:194-240 with ss_prob == 0, or replaying `fed_tokens` [B,T] =
    the tokens the reference actually fed after scheduled sampling."""
    P = prologue(p, ids, 'xe', fc, att, cpt_words, None, senti_labels, masks, p_drop)
    tok = captions[:, :-1] if fed_tokens is None else fed_tokens
    logp, weights = _teacher_forced(p, P, tok, masks, p_drop)
    return logp, P, weights


def forward_seq2seq(p, ids, senti_captions, cpt_words, senti_words, senti_labels,
                    masks=None, fed_tokens=None, p_drop=0.5):
    """captioner.py:242-288."""
    P = prologue(p, ids, 'seq2seq', None, None, cpt_words, senti_words, senti_labels,
                 masks, p_drop)
    tok = senti_captions[:, :-1] if fed_tokens is None else fed_tokens
    logp, weights = _teacher_forced(p, P, tok, masks, p_drop)
    return logp, P, weights


def forward_rl(p, ids, fc, att, cpt_words, senti_words, senti_labels, max_seq_len,
               sample_max=1, replay=None, masks=None, p_drop=0.5):
    """captioner.py:290-349. sample_max=1: greedy. Otherwise `replay` [B,T] holds the
    raw multinomial draws (before the `* unfinished` masking) to be replayed."""
    P = prologue(p, ids, 'rl', fc, att, cpt_words, senti_words, senti_labels, masks, p_drop)
    B = fc.shape[0]
    state = init_state(p, B)
    seq = torch.zeros((B, max_seq_len), dtype=torch.long)
    seq_logprobs = [fc.new_zeros(B) for _ in range(max_seq_len)]
    seq_masks = fc.new_zeros((B, max_seq_len))
    margins = fc.new_zeros((B, max_seq_len))
    it = torch.full((B,), ids.sos, dtype=torch.long)
    unfinished = it == ids.sos
    ws = []
    for t in range(max_seq_len):
        om = masks['out'][t] if (masks is not None and 'out' in masks) else None
        logp, state, w = step(p, it, state, P.fc_e, P.att_e, P.p_att, P.words_e,
                              P.p_words, P.label_e, om, p_drop)
        ws.append(w)
        top2 = torch.topk(logp.detach(), 2, dim=1).values
        margins[:, t] = top2[:, 0] - top2[:, 1]
        if sample_max:
            lp, it = torch.max(logp, dim=1)
        else:
            it = replay[:, t]
            lp = logp.gather(1, it.unsqueeze(1)).squeeze(1)
        seq_masks[:, t] = unfinished.to(seq_masks.dtype)
        it = it * unfinished.to(it.dtype)
        seq[:, t] = it
        seq_logprobs[t] = lp
        unfinished = unfinished & (it != ids.eos)
        if int(unfinished.sum()) == 0:
            break
    weights = tuple(_cat_weights(ws, k) for k in ('cont', 'senti', 'gate'))
    return seq, torch.stack(seq_logprobs, dim=1), seq_masks, P, weights, margins


def beam_search(p, ids, idx2word, fc_feat, att_feat, senti_words=None, senti_label=None,
                beam_size=3, decoding_constraint=1, max_seq_len=16):
    """captioner.py:351-420 for one image. Scores are python floats (fp64 sums of the
    step dtype's log-probs); candidate order uses the same stable sort semantics.
    Returns (captions, scores, id_sequences)."""
    fc = fc_feat.reshape(1, -1)
    att = att_feat.reshape(1, -1, att_feat.shape[-1])
    if senti_words is not None:
        P = prologue(p, ids, 'beam', fc, att, None, senti_words.reshape(1, -1), senti_label)
    else:
        P = prologue(p, ids, 'beam', fc, att, None, None, None)
    # each candidate: (state, score, last_word, word_ids)
    cands = [(init_state(p, 1), 0.0, ids.sos, [])]
    for t in range(max_seq_len):
        nxt = []
        all_ended = True
        for (state, score, last, words) in cands:
            if t > 0 and last == ids.eos:
                nxt.append((state, score, last, words))
                continue
            all_ended = False
            it = torch.tensor([last], dtype=torch.long)
            logp, st, _ = step(p, it, state, P.fc_e, P.att_e, P.p_att, P.words_e,
                               P.p_words, P.label_e)
            logp = logp.detach().squeeze(0).clone()
            if ids.pad != ids.eos:
                logp[ids.pad] = NEG_INF
                logp[ids.sos] = NEG_INF
                logp[ids.unk] = NEG_INF
            if decoding_constraint:
                logp[last] = NEG_INF
            vals, idx = torch.sort(logp, descending=True)
            for k in range(beam_size):
                nxt.append((st, score + float(vals[k]), int(idx[k]), words + [int(idx[k])]))
        cands = sorted(nxt, key=lambda c: c[1], reverse=True)[:beam_size]
        if all_ended:
            break
    caps = [' '.join(idx2word[i] for i in c[3] if i != ids.eos) for c in cands]
    return caps, [c[1] for c in cands], [c[3] for c in cands]


def xe_criterion(pred, target, lengths):
    """captioner.py:427-440: masked NLL, global token mean."""
    max_len = max(lengths)
    mask = pred.new_zeros(len(lengths), max_len)
    for i, l in enumerate(lengths):
        mask[i, :l] = 1
    nll = -pred.gather(2, target.unsqueeze(2)).squeeze(2) * mask
    return nll.sum() / mask.sum()


def reward_criterion(seq_logprobs, seq_masks, reward):
    """self_critical/utils.py:169-177."""
    return (-seq_logprobs * seq_masks * reward).sum() / seq_masks.sum()


def domain_align_loss(cpt_feats, fc_feats):
    """nn.MSELoss()(cpt_feats, fc_feats.detach()); train_xe.py:163, decoder.py:89."""
    return ((cpt_feats - fc_feats.detach()) ** 2).mean()


def clamp_adam_step(params, grads, m, v, step_no, lr, clip=0.1, b1=0.9, b2=0.999, eps=1e-8):
    """clip_gradient (train_xe.py:19-23) followed by torch.optim.Adam defaults
    (captioner.py:422-423), restated: returns nothing, updates in place."""
    with torch.no_grad():
        for k in params:
            g = grads[k].clamp(-clip, clip)
            m[k].mul_(b1).add_(g, alpha=1 - b1)
            v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** step_no
            bc2 = 1 - b2 ** step_no
            denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
            params[k].addcdiv_(m[k], denom, value=-lr / bc1)
