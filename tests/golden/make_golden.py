#!/usr/bin/env python3
"""Golden-vector generator. Runs ONLY in the build container, where the upstream
reference is mounted read-only at /root/reference; nothing here is needed on the
GPU box. It imports the reference *unmodified*, loads the deterministic weights of
`insenticap_model_amd.synth` through `load_state_dict`, runs it on the seeded
synthetic inputs and writes small `.npz` fixtures next to this file.

Only data (inputs are re-derivable from seeds; expected outputs are stored) is
committed - no reference source, bytecode or pickled module.

    python tests/golden/make_golden.py            # all cases
    python tests/golden/make_golden.py tiny cfg1  # selected cases

Observation hooks used (the reference code itself is not edited):
  * a forward hook on `captioner.drop` records each dropout keep-mask in call order;
  * `torch.multinomial` is wrapped to record the raw draws;
  * `captioner.forward_step` is wrapped to record fed tokens and top-2 margins.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('INSENTICAP_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from insenticap_model_amd import synth  # noqa: E402
from models.captioner import Captioner, XECriterion  # noqa: E402  (reference)
from self_critical.utils import RewardCriterion  # noqa: E402      (reference)

torch.set_num_threads(8)


def build_reference(V, settings, seed, dropout_p=None):
    st = dict(settings)
    if dropout_p is not None:
        st['dropout_p'] = dropout_p
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
    w = synth.make_weights(V, settings, seed=seed)
    missing = cap.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(cap.state_dict().keys()) == list(w.keys()), 'state_dict order differs'
    return cap


def T(d, *keys):
    return [torch.from_numpy(np.asarray(d[k])) for k in keys]


class StepSpy:
    """Wraps captioner.forward_step: records fed token ids and top-2 margins."""

    def __init__(self, cap):
        self.cap = cap
        self.orig = cap.forward_step
        self.fed, self.margins = [], []
        cap.forward_step = self

    def __call__(self, it, *a, **k):
        logp, state = self.orig(it, *a, **k)
        self.fed.append(it.detach().clone())
        top2 = torch.topk(logp.detach(), 2, dim=1).values
        self.margins.append((top2[:, 0] - top2[:, 1]).clone())
        return logp, state

    def close(self):
        self.cap.forward_step = self.orig

    def fed_matrix(self):
        return torch.stack(self.fed, dim=1).numpy()

    def margin_matrix(self):
        return torch.stack(self.margins, dim=1).numpy()


class DropSpy:
    def __init__(self, cap):
        self.masks = []
        self.h = cap.drop.register_forward_hook(self._hook)

    def _hook(self, mod, inp, out):
        if mod.training:
            self.masks.append((out.detach() != 0).numpy())

    def close(self):
        self.h.remove()


class MultinomialSpy:
    def __init__(self):
        self.draws = []
        self.orig = torch.multinomial
        torch.multinomial = self

    def __call__(self, *a, **k):
        r = self.orig(*a, **k)
        self.draws.append(r.detach().clone().view(-1))
        return r

    def close(self):
        torch.multinomial = self.orig


def grads_of(cap):
    return {k: (p.grad.detach().numpy().copy() if p.grad is not None else None)
            for k, p in cap.named_parameters()}


def clamp_adam_reference(cap, lr, clip=0.1):
    """train_xe.py:189-192 order: backward done by caller; clamp; Adam.step."""
    optim, _, _ = cap.get_optim_criterion(lr)
    for group in optim.param_groups:
        for prm in group['params']:
            if prm.grad is not None:
                prm.grad.data.clamp_(-clip, clip)
    optim.step()
    return {k: v.detach().numpy().copy() for k, v in cap.state_dict().items()}


def grad_digest(g):
    """Small per-tensor fingerprint for big tensors: sum, abs-sum, l2, 64 strided samples."""
    flat = g.reshape(-1).astype(np.float64)
    n = flat.size
    idx = (np.arange(64, dtype=np.int64) * 2654435761 % n)
    return np.concatenate([[flat.sum(), np.abs(flat).sum(), np.sqrt((flat ** 2).sum())],
                           flat[idx]])


def xe_train_iteration(cap, d, s2s, lr, out, prefix, full_grads):
    """One train_xe.py inner step (train_xe.py:155-192) in eval-mode dropout with grad on."""
    xe_crit, da_crit = XECriterion(), torch.nn.MSELoss()
    fc, att, cpt, caps, lab = T(d, 'fc_feats', 'att_feats', 'cpt_words', 'captions', 'senti_labels')
    cap.zero_grad()
    pred = cap(fc, att, cpt, caps, lab, 0.0, mode='xe')
    xe_loss = xe_crit(pred, caps[:, 1:], d['lengths'])
    da_loss = da_crit(cap.cpt_feats, cap.fc_feats.detach())
    out[prefix + 'xe_logp'] = pred.detach().numpy() if full_grads else pred.detach().numpy()[:, :, :32]
    out[prefix + 'xe_logp_tgt'] = pred.detach().gather(2, caps[:, 1:].unsqueeze(2)).squeeze(2).numpy()
    out[prefix + 'xe_fc_feats'] = cap.fc_feats.detach().numpy()
    out[prefix + 'xe_cpt_feats'] = cap.cpt_feats.detach().numpy()
    out[prefix + 'xe_cont_weights'] = cap.cont_weights.detach().numpy()
    scaps, scpt, ssw, slab = T(s2s, 'captions', 'cpt_words', 'senti_words', 'senti_labels')
    pred2 = cap(scaps, scpt, ssw, slab, 0.0, mode='seq2seq')
    s2s_loss = xe_crit(pred2, scaps[:, 1:], s2s['lengths'])
    out[prefix + 's2s_logp'] = pred2.detach().numpy() if full_grads else pred2.detach().numpy()[:, :, :32]
    out[prefix + 's2s_logp_tgt'] = pred2.detach().gather(2, scaps[:, 1:].unsqueeze(2)).squeeze(2).numpy()
    out[prefix + 's2s_senti_weights'] = cap.senti_weights.detach().numpy()
    all_loss = xe_loss + da_loss + s2s_loss
    all_loss.backward()
    out[prefix + 'losses'] = np.array([float(xe_loss), float(da_loss), float(s2s_loss)], np.float64)
    for k, g in grads_of(cap).items():
        if g is None:   # gate-fusion weights are unused in xe/seq2seq (captioner.py:101-103)
            continue
        if full_grads:
            out[prefix + 'grad/' + k] = g
        else:
            out[prefix + 'gdig/' + k] = grad_digest(g)
    newp = clamp_adam_reference(cap, lr)
    for k, v in newp.items():
        if full_grads:
            out[prefix + 'adam/' + k] = v
        else:
            out[prefix + 'adig/' + k] = grad_digest(v)


def rollout_cases(cap, d, Tlen, out, prefix):
    fc, att, cpt, sw, lab = T(d, 'fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')
    cap.eval()
    spy = StepSpy(cap)
    with torch.no_grad():
        seq, lp, mk = cap(fc, att, cpt, sw, lab, Tlen, 1, mode='rl')
    spy.close()
    out[prefix + 'greedy_seq'] = seq.numpy()
    out[prefix + 'greedy_logprobs'] = lp.numpy()
    out[prefix + 'greedy_masks'] = mk.numpy()
    out[prefix + 'greedy_margins'] = spy.margin_matrix()
    out[prefix + 'greedy_cont_weights'] = cap.cont_weights.numpy()
    out[prefix + 'greedy_senti_weights'] = cap.senti_weights.numpy()
    out[prefix + 'greedy_gate_weights'] = cap.cont_senti_weights.numpy()
    out[prefix + 'greedy_fc_feats'] = cap.fc_feats.numpy()
    out[prefix + 'greedy_cpt_feats'] = cap.cpt_feats.numpy()

    # sampled rollout (eval-mode dropout, grad on) + REINFORCE loss + grads
    torch.manual_seed(1234)
    ms = MultinomialSpy()
    cap.zero_grad()
    seq, lp, mk = cap(fc, att, cpt, sw, lab, Tlen, 0, mode='rl')
    ms.close()
    draws = torch.stack(ms.draws, dim=1).numpy()
    out[prefix + 'sample_draws'] = draws
    out[prefix + 'sample_seq'] = seq.numpy()
    out[prefix + 'sample_logprobs'] = lp.detach().numpy()
    out[prefix + 'sample_masks'] = mk.numpy()
    rng = np.random.default_rng(77)
    reward = np.repeat(rng.normal(size=(seq.shape[0], 1)), seq.shape[1], 1).astype(np.float32)
    out[prefix + 'sample_reward'] = reward
    loss = RewardCriterion()(lp, mk, torch.from_numpy(reward))
    loss.backward()
    out[prefix + 'sample_rl_loss'] = np.array([float(loss)])
    return grads_of(cap)


def beam_cases(cap, d, Tlen, out, prefix, n_images, beams):
    fc, att, sw, lab = T(d, 'fc_feats', 'att_feats', 'senti_words', 'senti_labels')
    for b in beams:
        for senti in (0, 1):
            caps_all, scores_all = [], []
            for i in range(n_images):
                with torch.no_grad():
                    if senti:
                        c, s = cap.sample(fc[i], att[i], sw[i], lab[i:i + 1], b, 1, Tlen)
                    else:
                        c, s = cap.sample(fc[i], att[i], None, None, b, 1, Tlen)
                caps_all.append(list(c) + [''] * (b - len(c)))
                scores_all.append(list(s) + [np.nan] * (b - len(s)))
            out[prefix + 'beam%d_senti%d_caps' % (b, senti)] = np.array(caps_all)
            out[prefix + 'beam%d_senti%d_scores' % (b, senti)] = np.array(scores_all, np.float64)
    # decoding_constraint off, one setting
    caps_all, scores_all = [], []
    for i in range(n_images):
        with torch.no_grad():
            c, s = cap.sample(fc[i], att[i], sw[i], lab[i:i + 1], beams[0], 0, Tlen)
        caps_all.append(list(c))
        scores_all.append(list(s))
    out[prefix + 'beam%d_nocons_caps' % beams[0]] = np.array(caps_all)
    out[prefix + 'beam%d_nocons_scores' % beams[0]] = np.array(scores_all, np.float64)


def case_tiny():
    """Tiny dims, full tensors incl. every gradient and post-Adam parameter."""
    V, st, seed = 64, synth.TINY_SETTINGS, 1
    B, R, Tlen = 6, 6, 8
    out = {}
    d = synth.make_inputs(B, V, st, regions=R, seq_len=Tlen, seed=11)
    s2s = synth.make_inputs(4, V, st, regions=R, seq_len=Tlen, seed=12)
    cap = build_reference(V, st, seed)
    cap.eval()
    xe_train_iteration(cap, d, s2s, 4e-4, out, 'it/', full_grads=True)

    cap = build_reference(V, st, seed)
    g = rollout_cases(cap, d, Tlen, out, 'rl/')
    for k, v in g.items():
        if v is not None:
            out['rl/grad/' + k] = v
    # rows 3..5 all emit <EOS> before T: pins the early `break` (captioner.py:343-344),
    # after which trailing seq / seq_logprobs / seq_masks columns stay zero
    de = {k: (v[3:6] if isinstance(v, np.ndarray) else v[3:6]) for k, v in d.items()}
    cap = build_reference(V, st, seed)
    rollout_cases(cap, de, Tlen, out, 'early/')
    assert out['early/greedy_masks'][:, -1].sum() == 0, 'early case does not end early'
    cap = build_reference(V, st, seed)
    beam_cases(cap, d, Tlen, out, 'beam/', n_images=B, beams=(3, 5))

    # 4-D grid input [B,h,w,F] must behave like [B,h*w,F] (captioner.py:208)
    dg = synth.make_inputs(B, V, st, regions=R, seq_len=Tlen, seed=11, grid=(2, 3))
    cap.eval()
    fc, att, cpt, caps, lab = T(dg, 'fc_feats', 'att_feats', 'cpt_words', 'captions', 'senti_labels')
    with torch.no_grad():
        out['grid/xe_logp'] = cap(fc, att, cpt, caps, lab, 0.0, mode='xe').numpy()

    # train mode with dropout p=0.5: masks recorded and stored (bool)
    cap = build_reference(V, st, seed)
    cap.train()
    torch.manual_seed(99)
    ds = DropSpy(cap)
    fc, att, cpt, caps, lab = T(d, 'fc_feats', 'att_feats', 'cpt_words', 'captions', 'senti_labels')
    cap.zero_grad()
    pred = cap(fc, att, cpt, caps, lab, 0.0, mode='xe')
    loss = XECriterion()(pred, caps[:, 1:], d['lengths'])
    loss.backward()
    ds.close()
    names = ['fc', 'att', 'label'] + ['out%d' % i for i in range(Tlen)]
    assert len(ds.masks) == len(names), len(ds.masks)
    for n, m in zip(names, ds.masks):
        out['drop/mask_' + n] = m
    out['drop/xe_logp'] = pred.detach().numpy()
    out['drop/loss'] = np.array([float(loss)])
    for k, v in grads_of(cap).items():
        if v is not None:
            out['drop/grad/' + k] = v

    # seq2seq in train mode with dropout: masks order cpt, words, label, out*
    cap = build_reference(V, st, seed)
    cap.train()
    torch.manual_seed(98)
    ds = DropSpy(cap)
    scaps, scpt, ssw, slab = T(s2s, 'captions', 'cpt_words', 'senti_words', 'senti_labels')
    pred = cap(scaps, scpt, ssw, slab, 0.0, mode='seq2seq')
    ds.close()
    names = ['cpt', 'words', 'label'] + ['out%d' % i for i in range(Tlen)]
    assert len(ds.masks) == len(names), len(ds.masks)
    for n, m in zip(names, ds.masks):
        out['drop_s2s/mask_' + n] = m
    out['drop_s2s/logp'] = pred.detach().numpy()

    # scheduled sampling, dropout disabled (p=0) so train mode is deterministic given draws
    cap = build_reference(V, st, seed, dropout_p=0.0)
    cap.train()
    torch.manual_seed(5)
    spy = StepSpy(cap)
    pred = cap(fc, att, cpt, caps, lab, 0.5, mode='xe')
    spy.close()
    out['ss/fed_tokens'] = spy.fed_matrix()
    out['ss/xe_logp'] = pred.detach().numpy()
    assert (out['ss/fed_tokens'] != caps[:, :-1].numpy()).any(), 'ss never triggered'
    np.savez_compressed(os.path.join(HERE, 'tiny.npz'), **out)
    print('tiny: %d arrays' % len(out))


def case_cfg1():
    """BASELINE.json configs[0]: B=4, 36x2048 feats, V=10k, T=20 (digests for big tensors)."""
    V, st, seed = 10000, synth.DEFAULT_SETTINGS, 0
    B, R, Tlen = 4, 36, 20
    out = {}
    d = synth.make_inputs(B, V, st, regions=R, seq_len=Tlen, seed=1)
    s2s = synth.make_inputs(5, V, st, regions=R, seq_len=Tlen, seed=2)
    cap = build_reference(V, st, seed)
    rollout_g = rollout_cases(cap, d, Tlen, out, 'rl/')
    for k, v in rollout_g.items():
        if v is not None:
            out['rl/gdig/' + k] = grad_digest(v)
    beam_cases(cap, d, Tlen, out, 'beam/', n_images=2, beams=(5,))
    cap = build_reference(V, st, seed)
    cap.eval()
    xe_train_iteration(cap, d, s2s, 4e-4, out, 'it/', full_grads=False)
    np.savez_compressed(os.path.join(HERE, 'cfg1.npz'), **out)
    print('cfg1: %d arrays' % len(out))


def case_b128():
    """BASELINE.json configs[1] shape: B=128 XE forward+backward, digests only; plus
    greedy at B=128 (token ids, logprobs, margins)."""
    V, st, seed = 10000, synth.DEFAULT_SETTINGS, 0
    B, R, Tlen = 128, 36, 20
    out = {}
    d = synth.make_inputs(B, V, st, regions=R, seq_len=Tlen, seed=21)
    s2s = synth.make_inputs(80, V, st, regions=R, seq_len=Tlen, seed=22)
    cap = build_reference(V, st, seed)
    fc, att, cpt, sw, lab = T(d, 'fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')
    cap.eval()
    spy = StepSpy(cap)
    with torch.no_grad():
        seq, lp, mk = cap(fc, att, cpt, sw, lab, Tlen, 1, mode='rl')
    spy.close()
    out['rl/greedy_seq'] = seq.numpy()
    out['rl/greedy_logprobs'] = lp.numpy()
    out['rl/greedy_masks'] = mk.numpy()
    out['rl/greedy_margins'] = spy.margin_matrix()
    xe_crit, da_crit = XECriterion(), torch.nn.MSELoss()
    caps = torch.from_numpy(d['captions'])
    cap.zero_grad()
    pred = cap(fc, att, cpt, caps, lab, 0.0, mode='xe')
    xe_loss = xe_crit(pred, caps[:, 1:], d['lengths'])
    da_loss = da_crit(cap.cpt_feats, cap.fc_feats.detach())
    scaps, scpt, ssw, slab = T(s2s, 'captions', 'cpt_words', 'senti_words', 'senti_labels')
    pred2 = cap(scaps, scpt, ssw, slab, 0.0, mode='seq2seq')
    s2s_loss = xe_crit(pred2, scaps[:, 1:], s2s['lengths'])
    (xe_loss + da_loss + s2s_loss).backward()
    out['it/losses'] = np.array([float(xe_loss), float(da_loss), float(s2s_loss)], np.float64)
    out['it/xe_logp_tgt'] = pred.detach().gather(2, caps[:, 1:].unsqueeze(2)).squeeze(2).numpy()
    out['it/s2s_logp_tgt'] = pred2.detach().gather(2, scaps[:, 1:].unsqueeze(2)).squeeze(2).numpy()
    for k, g in grads_of(cap).items():
        if g is not None:
            out['it/gdig/' + k] = grad_digest(g)
    newp = clamp_adam_reference(cap, 4e-4)
    for k, v in newp.items():
        out['it/adig/' + k] = grad_digest(v)
    np.savez_compressed(os.path.join(HERE, 'b128.npz'), **out)
    print('b128: %d arrays' % len(out))


def case_cider():
    """CIDEr-D self-critical reward (self_critical/utils.py:38-83) on synthetic captions."""
    from self_critical.utils import get_ciderd_scorer, get_self_critical_reward
    out = {}
    for name, (n_img, V, B, Tn, seed) in {'small': (40, 60, 16, 12, 7), 'cfg5': (600, 10000, 256, 20, 8)}.items():
        split, fns, gt, sample, greedy = synth.make_cider_data(n_img, V, B, seq_len=Tn, seed=seed)
        scorer = get_ciderd_scorer(split, 1, 2)
        rew = get_self_critical_reward(torch.from_numpy(sample), torch.from_numpy(greedy), fns, gt, 1, 2, scorer)
        assert rew.dtype == np.float64 and rew.shape == (B, Tn)
        out[name + '/reward'] = rew[:, 0].copy()
        # raw scores of the 2B hypotheses, through the scorer's own entry point
        res = [{'image_id': fn, 'caption': [' '.join(str(w) for w in _words(row))]}
               for rows in (sample, greedy) for fn, row in zip(fns, rows)]
        gts = {fn: [' '.join(str(w) for w in _words(c)) for c in gt[fn]] for fn in fns}
        _, scores = scorer.compute_score(gts, res)
        out[name + '/scores'] = np.asarray(scores, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'cider.npz'), **out)
    print('cider: %d arrays' % len(out))


def _words(arr, sos=1, eos=2):
    arr = [int(x) for x in arr]
    if arr[0] == sos:
        arr = arr[1:]
    o = []
    for w in arr:
        if w == eos:
            break
        o.append(w)
    return o + [eos]


def case_detector():
    """Helper nets and Detector.forward (models/decoder.py:52-180) in eval mode on tiny dims: the loss
    dictionary of two 'fact' batches, with the multinomial draws of the sampled roll-outs recorded."""
    from models.decoder import Detector
    V, Tn, B = 64, 8, 4
    st = dict(synth.TINY_SETTINGS, **synth.HELPER_SETTINGS)
    idx2word = synth.make_idx2word(V)
    det = Detector(idx2word, Tn, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
    det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=1).items()})
    for name, mod, seed in (('senti_detector', det.senti_detector, 51), ('sent_senti_cls', det.sent_senti_cls, 52)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_module_weights(shapes, seed).items()})
    batches, split = synth.make_rl_batches(2, B, V, st, seq_len=Tn)
    det.set_ciderd_scorer(split)
    tens = lambda b: (b[0], torch.from_numpy(b[1]), torch.from_numpy(b[2]), (torch.from_numpy(b[3][0]), b[3][1]),
                      torch.from_numpy(b[4]), torch.from_numpy(b[5]), b[6])
    out = {}
    # helper nets alone
    for i, b in enumerate(batches):
        labels, maps, _, scores = det.senti_detector.sample(torch.from_numpy(b[2]), 0.7)
        out['senti_det/labels%d' % i] = labels.numpy()
        out['senti_det/scores%d' % i] = scores.detach().numpy()
        out['senti_det/maps%d' % i] = maps.detach().numpy()
        logits0, _ = det.senti_detector(torch.from_numpy(b[2]))
        out['senti_det/logits%d' % i] = logits0.detach().numpy()
        det.sent_senti_cls.eval()
        with torch.no_grad():
            pred, w = det.sent_senti_cls(torch.from_numpy(b[3][0])[:, 1:], b[3][1])
        out['sent_cls/pred%d' % i] = pred.numpy()
        out['sent_cls/weights%d' % i] = w.numpy()
    # Detector.forward, eval mode
    ms = MultinomialSpy()
    bounds = []
    orig = det.captioner.forward_rl

    def spy_rl(*a, **k):
        n0 = len(ms.draws)
        r = orig(*a, **k)
        if not k.get('sample_max', a[-1] if len(a) >= 7 else 1):
            bounds.append((n0, len(ms.draws)))
        return r
    det.captioner.forward_rl = spy_rl
    torch.manual_seed(2024)
    losses = det(([tens(b) for b in batches],), 'fact', False)
    ms.close()
    det.captioner.forward_rl = orig
    for i, (a, b) in enumerate(bounds):
        dr = torch.stack(ms.draws[a:b], dim=1).numpy()
        out['det/draws%d' % i] = np.pad(dr, ((0, 0), (0, Tn - dr.shape[1])))
        out['det/steps%d' % i] = np.array([dr.shape[1]])
    assert len(bounds) == 2, bounds
    for k, v in losses.items():
        out['det/loss_' + k] = np.array([v], dtype=np.float64)
    print({k: round(v, 5) for k, v in losses.items()})
    # Detector.sample (models/decoder.py:182-192): beam search + detected sentiment, one image at a time
    caps_all, sentis_all = [], []
    for bi, b in enumerate(batches):
        for i in range(B):
            caps, det_sentis = det.sample(torch.from_numpy(b[1][i]), torch.from_numpy(b[2][i]),
                                          torch.from_numpy(b[5][i]), beam_size=3, decoding_constraint=1)
            caps_all.append(list(caps) + [''] * (3 - len(caps)))
            sentis_all.append(det_sentis[0])
    out['det/sample_caps'] = np.array(caps_all)
    out['det/sample_sentis'] = np.array(sentis_all)
    np.savez_compressed(os.path.join(HERE, 'detector.npz'), **out)
    print('detector: %d arrays' % len(out))


def case_det_train():
    """Detector.forward(data, 'fact', training=True) (models/decoder.py:52-180 incl. the update at :161-167) on tiny
    dims with dropout_p = 0: two iterations (two fact batches, one seq2seq batch served twice).  Every stochastic draw
    is recorded so the build can replay it: the raw multinomial draws of each sampled roll-out, and the tokens the XE
    (ss_prob 0.5) and seq2seq (ss_prob 0.25) unrolls actually fed under scheduled sampling.  Stored: the 7-key loss
    dictionary, every captioner parameter after the two clamp + Adam steps, and the (clamped) gradient of iteration 2."""
    from models.decoder import Detector
    V, Tn, B = 64, 8, 4
    st = dict(synth.TINY_SETTINGS, **synth.HELPER_SETTINGS)
    st['dropout_p'] = 0.0
    idx2word = synth.make_idx2word(V)
    det = Detector(idx2word, Tn, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-4}, st)
    det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=1).items()})
    for name, mod, seed in (('senti_detector', det.senti_detector, 51), ('sent_senti_cls', det.sent_senti_cls, 52)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_module_weights(shapes, seed).items()})
    batches, split = synth.make_rl_batches(2, B, V, st, seq_len=Tn)
    det.set_ciderd_scorer(split)
    tens = lambda b: (b[0], torch.from_numpy(b[1]), torch.from_numpy(b[2]), (torch.from_numpy(b[3][0]), b[3][1]),
                      torch.from_numpy(b[4]), torch.from_numpy(b[5]), b[6])
    s = synth.make_inputs(3, V, st, regions=6, seq_len=Tn, seed=77)
    t = torch.from_numpy
    scs = [((t(s['captions']), s['lengths']), t(s['cpt_words']), t(s['senti_words']), t(s['senti_labels']))]
    cap = det.captioner
    ms, spy = MultinomialSpy(), StepSpy(cap)
    rl_bounds, xe_bounds, s2s_bounds = [], [], []
    o_rl, o_xe, o_s2s = cap.forward_rl, cap.forward_xe, cap.forward_seq2seq

    def spy_rl(*a, **k):
        n0 = len(ms.draws)
        r = o_rl(*a, **k)
        if not k.get('sample_max', a[-1] if len(a) >= 7 else 1):
            rl_bounds.append((n0, len(ms.draws)))
        return r

    def spy_xe(*a, **k):
        n0 = len(spy.fed)
        r = o_xe(*a, **k)
        xe_bounds.append((n0, len(spy.fed)))
        return r

    def spy_s2s(*a, **k):
        n0 = len(spy.fed)
        r = o_s2s(*a, **k)
        s2s_bounds.append((n0, len(spy.fed)))
        return r
    cap.forward_rl, cap.forward_xe, cap.forward_seq2seq = spy_rl, spy_xe, spy_s2s
    torch.manual_seed(4242)
    losses = det(([tens(b) for b in batches], scs), 'fact', True)
    ms.close()
    spy.close()
    cap.forward_rl, cap.forward_xe, cap.forward_seq2seq = o_rl, o_xe, o_s2s
    assert len(rl_bounds) == len(xe_bounds) == len(s2s_bounds) == 2, (rl_bounds, xe_bounds, s2s_bounds)
    out = {}
    sampled = 0
    for i in range(2):
        a, b = rl_bounds[i]
        dr = torch.stack(ms.draws[a:b], dim=1).numpy()
        out['dt/draws%d' % i] = np.pad(dr, ((0, 0), (0, Tn - dr.shape[1])))
        out['dt/steps%d' % i] = np.array([dr.shape[1]])
        for name, (a, b), src in (('xe', xe_bounds[i], batches[i][3][0]), ('s2s', s2s_bounds[i], s['captions'])):
            fed = torch.stack(spy.fed[a:b], dim=1).numpy()
            out['dt/fed_%s%d' % (name, i)] = fed
            sampled += int((fed != np.asarray(src)[:, :fed.shape[1]]).sum())
    assert sampled > 0, 'scheduled sampling never replaced a token'
    for k, v in losses.items():
        out['dt/loss_' + k] = np.array([v], dtype=np.float64)
    assert set(losses) == {'da_loss', 'fact_reward', 'cls_reward', 'all_rewards', 'cap_loss', 'xe_loss', 'seq2seq_loss'}
    for k, v in cap.state_dict().items():
        out['dt/after/' + k] = v.detach().numpy().copy()
    for k, q in cap.named_parameters():          # what .grad holds after the call: iteration 2's gradient, clamped
        if q.grad is not None:
            out['dt/grad2/' + k] = q.grad.detach().numpy().copy()
    # the FIRST iteration alone (a fresh Detector, the same weights and seed: the same draws as iteration 1 above): the
    # CPU oracle checks its loss dictionary without having to replay an optimiser step in between
    det1 = Detector(idx2word, Tn, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-4}, st)
    det1.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=1).items()})
    for name, mod, seed in (('senti_detector', det1.senti_detector, 51), ('sent_senti_cls', det1.sent_senti_cls, 52)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_module_weights(shapes, seed).items()})
    det1.set_ciderd_scorer(split)
    rec = {}
    o_rl1 = det1.captioner.forward_rl

    def spy_rl1(*a, **k):
        r = o_rl1(*a, **k)
        key = 'greedy' if k.get('sample_max', a[-1] if len(a) >= 7 else 1) else 'sample'
        rec[key] = [x.detach().numpy().copy() for x in r]
        return r
    det1.captioner.forward_rl = spy_rl1
    torch.manual_seed(4242)
    losses1 = det1(([tens(batches[0])], scs), 'fact', True)
    det1.captioner.forward_rl = o_rl1
    for k, v in losses1.items():
        out['dt1/loss_' + k] = np.array([v], dtype=np.float64)
    out['dt1/sample_seq'], out['dt1/sample_logprobs'], out['dt1/sample_masks'] = rec['sample']
    out['dt1/greedy_seq'] = rec['greedy'][0]
    print({k: round(v, 5) for k, v in losses.items()}, 'tokens replaced by scheduled sampling:', sampled)
    print('iteration 1 alone:', {k: round(v, 5) for k, v in losses1.items()})
    np.savez_compressed(os.path.join(HERE, 'det_train.npz'), **out)
    print('det_train: %d arrays' % len(out))


def case_checkpoint():
    """A checkpoint file exactly as train_xe.py:241-254 writes it (tiny model, after one training step with
    the reference's Adam), plus the parameters the reference reaches after a SECOND step from that state:
    the build must load the file (model + optimizer) and reproduce step two."""
    V, st, seed = 64, synth.TINY_SETTINGS, 1
    d = synth.make_inputs(6, V, st, regions=6, seq_len=8, seed=11)
    s2s = synth.make_inputs(4, V, st, regions=6, seq_len=8, seed=12)
    cap = build_reference(V, st, seed)
    cap.eval()
    optim, xe_crit, da_crit = cap.get_optim_criterion(4e-4)

    def step():
        fc, att, cpt, caps, lab = T(d, 'fc_feats', 'att_feats', 'cpt_words', 'captions', 'senti_labels')
        pred = cap(fc, att, cpt, caps, lab, 0.0, mode='xe')
        loss = xe_crit(pred, caps[:, 1:], d['lengths']) + da_crit(cap.cpt_feats, cap.fc_feats.detach())
        scaps, scpt, ssw, slab = T(s2s, 'captions', 'cpt_words', 'senti_words', 'senti_labels')
        pred2 = cap(scaps, scpt, ssw, slab, 0.0, mode='seq2seq')
        loss = loss + xe_crit(pred2, scaps[:, 1:], s2s['lengths'])
        optim.zero_grad()
        loss.backward()
        for group in optim.param_groups:
            for prm in group['params']:
                if prm.grad is not None:
                    prm.grad.data.clamp_(-0.1, 0.1)
        optim.step()
        return float(loss.detach())
    l1 = step()
    chk = {'epoch': 7, 'model': cap.state_dict(), 'optimizer': optim.state_dict(), 'settings': dict(st),
           'idx2word': synth.make_idx2word(V), 'sentiment_categories': list(synth.SENTIMENT_CATEGORIES),
           'dataset_name': 'coco', 'corpus_type': 'part'}
    torch.save(chk, os.path.join(HERE, 'ref_xe_checkpoint_tiny.pth'))
    l2 = step()
    out = {'loss1': np.array([l1]), 'loss2': np.array([l2])}
    for k, v in cap.state_dict().items():
        out['after2/' + k] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'checkpoint.npz'), **out)
    print('checkpoint: %d arrays' % len(out))


def case_beam64():
    """BASELINE.json configs[2]: beam 5 over 64 images, sentiment words + label supplied, decoding_constraint=1,
    T=20 - all 64 images through the reference's one-image `Captioner.sample` (captioner.py:351-420)."""
    V, st, seed = 10000, synth.DEFAULT_SETTINGS, 0
    n, R, Tlen = 64, 36, 20
    d = synth.make_inputs(n, V, st, regions=R, seq_len=Tlen, seed=321)
    cap = build_reference(V, st, seed)
    cap.eval()
    out = {}
    beam_cases_one(cap, d, Tlen, out, 'beam/', n_images=n, beam=5)
    np.savez_compressed(os.path.join(HERE, 'beam64.npz'), **out)
    print('beam64: %d arrays' % len(out))


def beam_cases_one(cap, d, Tlen, out, prefix, n_images, beam):
    fc, att, sw, lab = T(d, 'fc_feats', 'att_feats', 'senti_words', 'senti_labels')
    caps_all, scores_all = [], []
    for i in range(n_images):
        with torch.no_grad():
            c, s = cap.sample(fc[i], att[i], sw[i], lab[i:i + 1], beam, 1, Tlen)
        caps_all.append(list(c) + [''] * (beam - len(c)))
        scores_all.append(list(s) + [np.nan] * (beam - len(s)))
    out[prefix + 'beam%d_senti1_caps' % beam] = np.array(caps_all)
    out[prefix + 'beam%d_senti1_scores' % beam] = np.array(scores_all, np.float64)


def case_det512():
    """BASELINE.json configs[4] at full size on ONE batch: Detector.forward(data, 'fact', training=False) of
    models/decoder.py:52-180 with B=512, V=10k, T=20, 6x6x2048 grid - sampled roll-out (draws recorded), greedy
    roll-out, CIDEr-D + classifier rewards, RL / XE / domain-align losses.  Also records the per-row quantities the
    dictionary is made of, so that a mismatch can be traced to its row."""
    from models.decoder import Detector
    from self_critical import utils as ref_utils
    V, Tn, B = 10000, 20, 512
    st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
    idx2word = synth.make_idx2word(V)
    det = Detector(idx2word, Tn, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
    det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
    for name, mod, seed in (('senti_detector', det.senti_detector, 51), ('sent_senti_cls', det.sent_senti_cls, 52)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_module_weights(shapes, seed).items()})
    batches, split = synth.make_rl_batches(1, B, V, st, grid=(6, 6), seq_len=Tn, seed=60)
    det.set_ciderd_scorer(split)
    b = batches[0]
    item = (b[0], torch.from_numpy(b[1]), torch.from_numpy(b[2]), (torch.from_numpy(b[3][0]), b[3][1]),
            torch.from_numpy(b[4]), torch.from_numpy(b[5]), b[6])
    out = {}
    ms = MultinomialSpy()
    spy = StepSpy(det.captioner)
    rec = {}
    orig_rl, orig_scr, orig_cls = det.captioner.forward_rl, ref_utils.get_self_critical_reward, ref_utils.get_cls_reward
    import models.decoder as ref_decoder

    def spy_rl(*a, **k):
        n0, s0 = len(ms.draws), len(spy.margins)
        r = orig_rl(*a, **k)
        greedy = bool(k.get('sample_max', a[-1] if len(a) >= 7 else 1))
        tag = 'greedy' if greedy else 'sample'
        rec[tag + '_seq'], rec[tag + '_logprobs'], rec[tag + '_masks'] = [x.detach().numpy().copy() for x in r]
        m = torch.stack(spy.margins[s0:], dim=1).numpy()
        rec[tag + '_margins'] = np.pad(m, ((0, 0), (0, Tn - m.shape[1])))
        if not greedy:
            dr = torch.stack(ms.draws[n0:], dim=1).numpy()
            rec['draws'] = np.pad(dr, ((0, 0), (0, Tn - dr.shape[1])))
        return r

    def spy_scr(*a, **k):
        r = orig_scr(*a, **k)
        rec['fact_reward_rows'] = np.asarray(r)[:, 0].copy()
        return r

    def spy_cls(*a, **k):
        r = orig_cls(*a, **k)
        rec['cls_reward'] = np.asarray(r).copy()
        return r
    orig_sd = det.senti_detector.sample

    def spy_sd(*a, **k):
        r = orig_sd(*a, **k)
        rec['senti_labels'], rec['senti_scores'] = r[0].numpy().copy(), r[3].detach().numpy().copy()
        return r
    det.senti_detector.sample = spy_sd
    det.captioner.forward_rl = spy_rl
    ref_decoder.get_self_critical_reward = spy_scr
    ref_decoder.get_cls_reward = spy_cls
    torch.manual_seed(515)
    losses = det(([item],), 'fact', False)
    ms.close()
    spy.close()
    det.captioner.forward_rl = orig_rl
    det.senti_detector.sample = orig_sd
    ref_decoder.get_self_critical_reward, ref_decoder.get_cls_reward = orig_scr, orig_cls
    for k, v in rec.items():
        out['det/' + k] = v
    for k, v in losses.items():
        out['det/loss_' + k] = np.array([v], dtype=np.float64)
    live = rec['greedy_masks'] > 0
    print({k: round(v, 5) for k, v in losses.items()}, 'min greedy margin on live steps',
          float(rec['greedy_margins'][live].min()))
    np.savez_compressed(os.path.join(HERE, 'det512.npz'), **out)
    print('det512: %d arrays' % len(out))


def _ref_detector(V, Tn, st, lr, weight_seed):
    """The reference Detector with the build's deterministic weights (captioner + both helper nets)."""
    from models.decoder import Detector
    det = Detector(synth.make_idx2word(V), Tn, synth.SENTIMENT_CATEGORIES, {'cap_lr': lr}, st)
    det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=weight_seed).items()})
    for name, mod, seed in (('senti_detector', det.senti_detector, 51), ('sent_senti_cls', det.sent_senti_cls, 52)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_module_weights(shapes, seed).items()})
    return det


class _CallSpy:
    """Wraps forward_rl / forward_xe / forward_seq2seq of a reference captioner: index ranges of the multinomial draws
    of every sampled roll-out and of the tokens every teacher-forced unroll fed."""

    def __init__(self, cap):
        self.cap, self.ms, self.spy = cap, MultinomialSpy(), StepSpy(cap)
        self.rl, self.xe, self.s2s = [], [], []
        self.orig = (cap.forward_rl, cap.forward_xe, cap.forward_seq2seq)
        o_rl, o_xe, o_s2s = self.orig

        self.greedy = []          # (seq, masks) of every greedy roll-out (sample_max = 1), as the reference returned them

        def spy_rl(*a, **k):
            n0 = len(self.ms.draws)
            r = o_rl(*a, **k)
            if not k.get('sample_max', a[-1] if len(a) >= 7 else 1):
                self.rl.append((n0, len(self.ms.draws)))
            else:
                self.greedy.append((r[0].detach().numpy().copy(), r[2].detach().numpy().copy()))
            return r

        def spy_xe(*a, **k):
            n0 = len(self.spy.fed)
            r = o_xe(*a, **k)
            self.xe.append((n0, len(self.spy.fed)))
            return r

        def spy_s2s(*a, **k):
            n0 = len(self.spy.fed)
            r = o_s2s(*a, **k)
            self.s2s.append((n0, len(self.spy.fed)))
            return r
        cap.forward_rl, cap.forward_xe, cap.forward_seq2seq = spy_rl, spy_xe, spy_s2s

    def close(self):
        self.ms.close()
        self.spy.close()
        self.cap.forward_rl, self.cap.forward_xe, self.cap.forward_seq2seq = self.orig

    def draws(self, i, Tn):
        a, b = self.rl[i]
        dr = torch.stack(self.ms.draws[a:b], dim=1).numpy()
        return np.pad(dr, ((0, 0), (0, Tn - dr.shape[1]))), dr.shape[1]

    def fed(self, bounds, i):
        a, b = bounds[i]
        return torch.stack(self.spy.fed[a:b], dim=1).numpy()


def _senti_items(batches, seed):
    """rl_senti collate layout (dataloader.py:93-109): (fns, fc, att, cpts, sentis, senti_labels)."""
    rng = np.random.default_rng(seed)
    items = []
    for b in batches:
        labels = rng.integers(0, len(synth.SENTIMENT_CATEGORIES), size=len(b[0])).astype(np.int64)
        items.append((b[0], torch.from_numpy(b[1]), torch.from_numpy(b[2]), torch.from_numpy(b[4]), torch.from_numpy(b[5]),
                      torch.from_numpy(labels)))
    return items


def case_det_senti():
    """Detector.forward((senti_loader, scs_loader), 'senti', training) (train_rl.py:232-235; models/decoder.py:69-71,
    100-101: sentiment labels from the batch when training, from the image detector otherwise; no CIDEr reward, no XE
    term) on tiny dims with dropout_p = 0: two training iterations (draws of the sampled roll-outs and the tokens the
    seq2seq unroll fed under scheduled sampling recorded; loss dictionary, iteration 2's clamped gradient, every parameter
    after the two steps) and the same two batches with training = False on fresh weights."""
    V, Tn, B = 64, 8, 4
    st = dict(synth.TINY_SETTINGS, **synth.HELPER_SETTINGS)
    st['dropout_p'] = 0.0
    batches, _ = synth.make_rl_batches(2, B, V, st, seq_len=Tn, seed=90)
    s = synth.make_inputs(3, V, st, regions=6, seq_len=Tn, seed=78)
    t = torch.from_numpy
    scs = [((t(s['captions']), s['lengths']), t(s['cpt_words']), t(s['senti_words']), t(s['senti_labels']))]
    out = {}
    for b_i, it in enumerate(_senti_items(batches, 91)):
        out['ds/labels%d' % b_i] = it[5].numpy()
    # training
    det = _ref_detector(V, Tn, st, 4e-4, 1)
    cs = _CallSpy(det.captioner)
    torch.manual_seed(777)
    losses = det((_senti_items(batches, 91), scs), 'senti', True)
    cs.close()
    assert len(cs.rl) == 2 and len(cs.s2s) == 2 and len(cs.xe) == 0, (cs.rl, cs.xe, cs.s2s)
    for i in range(2):
        out['ds/draws%d' % i], steps = cs.draws(i, Tn)
        out['ds/steps%d' % i] = np.array([steps])
        out['ds/fed_s2s%d' % i] = cs.fed(cs.s2s, i)
    assert set(losses) == {'da_loss', 'cls_reward', 'all_rewards', 'cap_loss', 'seq2seq_loss'}, set(losses)
    for k, v in losses.items():
        out['ds/loss_' + k] = np.array([v], dtype=np.float64)
    for k, v in det.captioner.state_dict().items():
        out['ds/after/' + k] = v.detach().numpy().copy()
    for k, q in det.captioner.named_parameters():
        if q.grad is not None:
            out['ds/grad2/' + k] = q.grad.detach().numpy().copy()
    print('senti, training:', {k: round(v, 5) for k, v in losses.items()})
    # evaluation (labels from the image detector)
    det = _ref_detector(V, Tn, st, 4e-4, 1)
    cs = _CallSpy(det.captioner)
    torch.manual_seed(778)
    losses = det((_senti_items(batches, 91),), 'senti', False)
    cs.close()
    assert len(cs.rl) == 2 and not cs.xe and not cs.s2s
    for i in range(2):
        out['dse/draws%d' % i], steps = cs.draws(i, Tn)
        out['dse/steps%d' % i] = np.array([steps])
    assert set(losses) == {'da_loss', 'cls_reward', 'all_rewards', 'cap_loss'}, set(losses)
    for k, v in losses.items():
        out['dse/loss_' + k] = np.array([v], dtype=np.float64)
    print('senti, eval:', {k: round(v, 5) for k, v in losses.items()})
    np.savez_compressed(os.path.join(HERE, 'det_senti.npz'), **out)
    print('det_senti: %d arrays' % len(out))


def case_det512_train():
    """BASELINE.json configs[4] at full size, TRAINING: TWO iterations of Detector.forward(data, 'fact', True)
    (models/decoder.py:52-180 incl. the update at :161-167), one call per iteration, on two different fact batches with
    B = 512, V = 10k, T = 20, a 6 x 6 x 2048 grid, an 80-caption seq2seq batch, dropout_p = 0, lr 4e-5 (opts.py:41-42).
    Recorded for replay, per iteration: the multinomial draws of the sampled roll-out, the tokens the XE (ss_prob 0.5)
    and seq2seq (ss_prob 0.25) unrolls fed, and the greedy baseline's token and mask matrices (so that a test can hand
    the REINFORCE term the reference's own reward: a greedy near-tie flipped by fp32 reassociation changes CIDEr-D of
    that row by up to its whole range).  Stored per iteration: the 7-key loss dictionary, a digest (sum, abs-sum, l2, 64
    strided samples) of every clamped gradient and of every parameter after the step.  Keys of iteration 1 carry no
    suffix (d5t/grad/..., d5t/after/...), those of iteration 2 the suffix 2 (d5t/grad2/..., d5t/after2/...)."""
    V, Tn, B, Bs = 10000, 20, 512, 80
    st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
    st['dropout_p'] = 0.0
    det = _ref_detector(V, Tn, st, 4e-5, 0)
    # (iteration 1 = the batch of rounds 3-4: make_rl_batches(1, ..., seed=60) - its document frequencies come from ITS
    # references alone, as before; iteration 2 = another batch of the same generator with its own references)
    batches, split = synth.make_rl_batches(1, B, V, st, grid=(6, 6), seq_len=Tn, seed=60)
    batches2, split2 = synth.make_rl_batches(1, B, V, st, grid=(6, 6), seq_len=Tn, seed=61)
    s = synth.make_inputs(Bs, V, st, regions=6, seq_len=Tn, seed=79)
    t = torch.from_numpy
    scs = [((t(s['captions']), s['lengths']), t(s['cpt_words']), t(s['senti_words']), t(s['senti_labels']))]
    cs = _CallSpy(det.captioner)
    torch.manual_seed(5150)
    import time
    out = {}
    for it, (bb, sp) in enumerate(((batches, split), (batches2, split2))):
        sfx = '' if it == 0 else '2'
        det.set_ciderd_scorer(sp)
        b = bb[0]
        item = (b[0], torch.from_numpy(b[1]), torch.from_numpy(b[2]), (torch.from_numpy(b[3][0]), b[3][1]),
                torch.from_numpy(b[4]), torch.from_numpy(b[5]), b[6])
        t0 = time.time()
        losses = det(([item], scs), 'fact', True)
        print('reference iteration %d: %.1f s' % (it + 1, time.time() - t0))
        assert len(cs.rl) == it + 1 and len(cs.xe) == it + 1 and len(cs.s2s) == it + 1 and len(cs.greedy) == it + 1
        out['d5t/draws' + sfx], steps = cs.draws(it, Tn)
        out['d5t/steps' + sfx] = np.array([steps])
        out['d5t/fed_xe' + sfx] = cs.fed(cs.xe, it)
        out['d5t/fed_s2s' + sfx] = cs.fed(cs.s2s, it)
        gseq, gmk = cs.greedy[it]
        out['d5t/greedy_seq' + sfx] = np.pad(gseq, ((0, 0), (0, Tn - gseq.shape[1]))).astype(np.int64)
        out['d5t/greedy_masks' + sfx] = np.pad(gmk, ((0, 0), (0, Tn - gmk.shape[1]))).astype(np.float32)
        assert set(losses) == {'da_loss', 'fact_reward', 'cls_reward', 'all_rewards', 'cap_loss', 'xe_loss', 'seq2seq_loss'}
        for k, v in losses.items():
            out['d5t/loss%s_%s' % (sfx, k)] = np.array([v], dtype=np.float64)
        for k, q in det.captioner.named_parameters():
            if q.grad is not None:
                out['d5t/grad%s/%s' % (sfx, k)] = grad_digest(q.grad.detach().numpy())
                out['d5t/gmax%s/%s' % (sfx, k)] = np.array([float(q.grad.detach().abs().max())])
        for k, v in det.captioner.state_dict().items():
            out['d5t/after%s/%s' % (sfx, k)] = grad_digest(v.detach().numpy())
        print({k: round(v, 5) for k, v in losses.items()})
    cs.close()
    np.savez_compressed(os.path.join(HERE, 'det512_train.npz'), **out)
    print('det512_train: %d arrays' % len(out))


def case_collate():
    """The four collate functions the decoder path consumes (dataloader.py:11-109: caption, scs, rl_fact, rl_senti).
    `dataloader.py` imports h5py at module level, which this image does not have; only its Dataset classes
    (dataloader.py:171-178 ...) ever touch it - the collates are pure list / torch code.  The module is therefore
    imported with an EMPTY module object registered as `h5py` (no attribute of it is ever read on this path); the
    h5-backed Dataset classes are NOT exercised and stay unpinned.  Inputs are written next to the outputs so the
    test needs no generator."""
    import random
    import types
    sys.modules.setdefault('h5py', types.ModuleType('h5py'))
    import dataloader as ref_dl
    rng = np.random.default_rng(2718)
    out = {}

    def ragged(n, lo, hi, V=50):
        return [[int(x) for x in rng.integers(4, V, size=int(rng.integers(lo, hi + 1)))] for _ in range(n)]
    n_img, F = 7, 8
    fns = ['img%02d' % i for i in range(n_img)]
    fcs = rng.random((n_img, F), dtype=np.float32)
    atts = rng.random((n_img, 2, 3, F), dtype=np.float32)
    caps5 = [ragged(5, 3, 14) for _ in range(n_img)]           # lengths straddle max_seq_len=9 -> truncation + ties
    cpts = ragged(n_img, 2, 8)                                  # shorter and longer than num_concepts=5
    sentis = ragged(n_img, 3, 13)                               # shorter and longer than num_sentiments=10
    labels = [int(x) for x in rng.integers(0, 3, size=n_img)]
    kw = dict(pad_index=0, max_seq_len=9, num_concepts=5, num_sentiments=10)

    def dump_ragged(key, rows):
        out[key + '/flat'] = np.asarray([x for r in rows for x in r], dtype=np.int64)
        out[key + '/len'] = np.asarray([len(r) for r in rows], dtype=np.int64)
    out['in/fns'] = np.array(fns)
    out['in/fc'], out['in/att'] = fcs, atts
    dump_ragged('in/caps', [c for img in caps5 for c in img])
    dump_ragged('in/cpts', cpts)
    dump_ragged('in/sentis', sentis)
    out['in/labels'] = np.asarray(labels, dtype=np.int64)

    # caption (dataloader.py:11-34)
    f = ref_dl.create_collate_fn('caption', **kw)
    r = f([(fns[i], fcs[i], atts[i], caps5[i], cpts[i]) for i in range(n_img)])
    out['caption/fns'] = np.array(r[0])
    out['caption/fc'], out['caption/att'] = r[1].numpy(), r[2].numpy()
    out['caption/caps'], out['caption/lengths'] = r[3][0].numpy(), np.asarray(r[3][1], dtype=np.int64)
    out['caption/cpts'] = r[4].numpy()
    # scs (dataloader.py:36-58): rows of (cap, cpts, sentis, senti_id)
    f = ref_dl.create_collate_fn('senti_corpus_with_sentis', **kw)
    scs_rows = [(caps5[i][0], cpts[i], sentis[i], labels[i]) for i in range(n_img)]
    r = f(list(scs_rows))
    out['scs/caps'], out['scs/lengths'] = r[0][0].numpy(), np.asarray(r[0][1], dtype=np.int64)
    out['scs/cpts'], out['scs/sentis'], out['scs/labels'] = r[1].numpy(), r[2].numpy(), r[3].numpy()
    # rl_fact (dataloader.py:60-91): draws one caption per image with random.sample
    f = ref_dl.create_collate_fn('rl_fact', **kw)
    random.seed(31337)
    r = f([(fns[i], caps5[i], fcs[i], atts[i], cpts[i], sentis[i]) for i in range(n_img)])
    out['rl_fact/fns'] = np.array(r[0])
    out['rl_fact/fc'], out['rl_fact/att'] = r[1].numpy(), r[2].numpy()
    out['rl_fact/caps'], out['rl_fact/lengths'] = r[3][0].numpy(), np.asarray(r[3][1], dtype=np.int64)
    out['rl_fact/cpts'], out['rl_fact/sentis'] = r[4].numpy(), r[5].numpy()
    gt_rows = [c for fn in fns for c in r[6][fn]]
    dump_ragged('rl_fact/gt', gt_rows)
    out['rl_fact/gt_count'] = np.asarray([len(r[6][fn]) for fn in fns], dtype=np.int64)
    # rl_senti (dataloader.py:93-109)
    f = ref_dl.create_collate_fn('rl_senti', **kw)
    r = f([(fns[i], fcs[i], atts[i], cpts[i], sentis[i], labels[i]) for i in range(n_img)])
    out['rl_senti/fns'] = np.array(r[0])
    out['rl_senti/fc'], out['rl_senti/att'] = r[1].numpy(), r[2].numpy()
    out['rl_senti/cpts'], out['rl_senti/sentis'], out['rl_senti/labels'] = r[3].numpy(), r[4].numpy(), r[5].numpy()
    np.savez_compressed(os.path.join(HERE, 'collate.npz'), **out)
    print('collate: %d arrays' % len(out))


CASES = {'tiny': case_tiny, 'cfg1': case_cfg1, 'b128': case_b128, 'cider': case_cider, 'detector': case_detector,
         'checkpoint': case_checkpoint, 'beam64': case_beam64, 'det512': case_det512,
         'collate': case_collate, 'det_train': case_det_train, 'det_senti': case_det_senti,
         'det512_train': case_det512_train}

if __name__ == '__main__':
    which = sys.argv[1:] or list(CASES)
    for name in which:
        CASES[name]()
