"""Pins oracle/captioner_oracle.py against every golden vector produced from the
reference itself (tests/golden/make_golden.py). CPU only."""
import numpy as np
import pytest
import torch

from conftest import assert_digest_close, case_setup, digest, trusted_prefix
from insenticap_model_amd import synth
from oracle import captioner_oracle as O

torch.set_num_threads(8)


def tt(d, k):
    return torch.from_numpy(np.asarray(d[k]))


def ids_for(V):
    return O.Ids(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES)


def run_iteration(name):
    c, st, w, d, s2s = case_setup(name)
    ids = ids_for(c['V'])
    p = O.to_params(w, torch.float32, True)
    logp, P, ws = O.forward_xe(p, ids, tt(d, 'fc_feats'), tt(d, 'att_feats'), tt(d, 'cpt_words'),
                               tt(d, 'captions'), tt(d, 'senti_labels'))
    xe = O.xe_criterion(logp, tt(d, 'captions')[:, 1:], d['lengths'])
    da = O.domain_align_loss(P.cpt, P.fc_raw)
    logp2, P2, ws2 = O.forward_seq2seq(p, ids, tt(s2s, 'captions'), tt(s2s, 'cpt_words'),
                                       tt(s2s, 'senti_words'), tt(s2s, 'senti_labels'))
    l2 = O.xe_criterion(logp2, tt(s2s, 'captions')[:, 1:], s2s['lengths'])
    (xe + da + l2).backward()
    return c, d, s2s, p, logp, P, ws, logp2, ws2, (float(xe.detach()), float(da.detach()), float(l2.detach()))


def test_tiny_iteration_full_tensors(golden):
    g = golden('tiny')
    c, d, s2s, p, logp, P, ws, logp2, ws2, losses = run_iteration('tiny')
    np.testing.assert_allclose(logp.detach().numpy(), g['it/xe_logp'], atol=2e-5)
    np.testing.assert_allclose(logp2.detach().numpy(), g['it/s2s_logp'], atol=2e-5)
    np.testing.assert_allclose(P.fc_raw.detach().numpy(), g['it/xe_fc_feats'], atol=1e-6)
    np.testing.assert_allclose(P.cpt.detach().numpy(), g['it/xe_cpt_feats'], atol=1e-6)
    np.testing.assert_allclose(ws[0].detach().numpy(), g['it/xe_cont_weights'], atol=1e-6)
    np.testing.assert_allclose(ws2[1].detach().numpy(), g['it/s2s_senti_weights'], atol=1e-6)
    np.testing.assert_allclose(losses, g['it/losses'], rtol=1e-5)
    n = 0
    for k in p:
        if ('it/grad/' + k) not in g.files:
            assert p[k].grad is None or float(p[k].grad.abs().max()) == 0.0, k
            continue
        ref = g['it/grad/' + k]
        np.testing.assert_allclose(p[k].grad.numpy(), ref, atol=1e-4 * np.abs(ref).max() + 1e-7, err_msg=k)
        n += 1
    assert n == 32  # 40 tensors minus the 8 gate-fusion ones (4 layers x w,b) unused in xe/seq2seq
    m = {k: torch.zeros_like(v) for k, v in p.items() if v.grad is not None}
    v = {k: torch.zeros_like(x) for k, x in m.items()}
    O.clamp_adam_step({k: p[k] for k in m}, {k: p[k].grad for k in m}, m, v, 1, 4e-4)
    for k in p:
        # Adam's first step is lr*g/(|g|+eps): where |g| is within a few orders of eps=1e-8 the
        # update magnifies fp32 noise in g, so those elements are only bounded by one lr step.
        gref = np.abs(g['it/grad/' + k]) if ('it/grad/' + k) in g.files else None
        got, ref = p[k].detach().numpy(), g['it/adam/' + k]
        if gref is None:
            np.testing.assert_array_equal(got, ref, err_msg=k)      # never touched by the optimiser
            continue
        big = gref > 1e-4
        np.testing.assert_allclose(got[big], ref[big], atol=2e-6, err_msg=k)
        np.testing.assert_allclose(got[~big], ref[~big], atol=8.2e-4, err_msg=k)   # sign flips: up to 2*lr


@pytest.mark.parametrize('prefix', ['rl/', 'early/'])
def test_tiny_rollouts(golden, prefix):
    g = golden('tiny')
    c, st, w, d, _ = case_setup('tiny')
    if prefix == 'early/':
        d = {k: v[3:6] for k, v in d.items()}
    ids = ids_for(c['V'])
    p = O.to_params(w, torch.float32, True)
    a = (tt(d, 'fc_feats'), tt(d, 'att_feats'), tt(d, 'cpt_words'), tt(d, 'senti_words'), tt(d, 'senti_labels'))
    with torch.no_grad():
        seq, lp, mk, P, ws, mar = O.forward_rl(p, ids, *a, c['T'], 1)
    assert (seq.numpy() == g[prefix + 'greedy_seq']).all()
    assert (mk.numpy() == g[prefix + 'greedy_masks']).all()
    np.testing.assert_allclose(lp.numpy(), g[prefix + 'greedy_logprobs'], atol=2e-5)
    steps = g[prefix + 'greedy_margins'].shape[1]
    np.testing.assert_allclose(mar.numpy()[:, :steps], g[prefix + 'greedy_margins'], atol=5e-5)
    for i, key in enumerate(['cont', 'senti', 'gate']):
        np.testing.assert_allclose(ws[i].numpy(), g[prefix + 'greedy_%s_weights' % key], atol=2e-6)
    np.testing.assert_allclose(P.fc_raw.numpy(), g[prefix + 'greedy_fc_feats'], atol=1e-6)
    np.testing.assert_allclose(P.cpt.numpy(), g[prefix + 'greedy_cpt_feats'], atol=1e-6)
    # sampled rollout replayed from the reference's raw multinomial draws
    draws = torch.from_numpy(g[prefix + 'sample_draws'])
    seq, lp, mk, *_ = O.forward_rl(p, ids, *a, c['T'], 0, replay=draws)
    assert (seq.numpy() == g[prefix + 'sample_seq']).all()
    assert (mk.numpy() == g[prefix + 'sample_masks']).all()
    np.testing.assert_allclose(lp.detach().numpy(), g[prefix + 'sample_logprobs'], atol=2e-5)
    loss = O.reward_criterion(lp, mk, torch.from_numpy(g[prefix + 'sample_reward']))
    np.testing.assert_allclose(float(loss.detach()), g[prefix + 'sample_rl_loss'][0], rtol=2e-5)
    if prefix == 'rl/':
        loss.backward()
        for k in p:
            key = 'rl/grad/' + k
            if key in g.files:
                ref = g[key]
                np.testing.assert_allclose(p[k].grad.numpy(), ref, atol=1e-4 * np.abs(ref).max() + 1e-7,
                                           err_msg=k)


def test_tiny_beam(golden):
    g = golden('tiny')
    c, st, w, d, _ = case_setup('tiny')
    ids = ids_for(c['V'])
    i2w = synth.make_idx2word(c['V'])
    p = O.to_params(w)
    with torch.no_grad():
        for b in (3, 5):
            for s in (0, 1):
                for i in range(c['B']):
                    caps, sc, _ = O.beam_search(
                        p, ids, i2w, tt(d, 'fc_feats')[i], tt(d, 'att_feats')[i],
                        tt(d, 'senti_words')[i] if s else None,
                        tt(d, 'senti_labels')[i:i + 1] if s else None, b, 1, c['T'])
                    assert caps == list(g['beam/beam%d_senti%d_caps' % (b, s)][i])
                    np.testing.assert_allclose(sc, g['beam/beam%d_senti%d_scores' % (b, s)][i], atol=1e-4)
        for i in range(c['B']):
            caps, sc, _ = O.beam_search(p, ids, i2w, tt(d, 'fc_feats')[i], tt(d, 'att_feats')[i],
                                        tt(d, 'senti_words')[i], tt(d, 'senti_labels')[i:i + 1], 3, 0, c['T'])
            assert caps == list(g['beam/beam3_nocons_caps'][i])
            np.testing.assert_allclose(sc, g['beam/beam3_nocons_scores'][i], atol=1e-4)


def test_tiny_grid_input_equals_flat(golden):
    g = golden('tiny')
    c, st, w, d, _ = case_setup('tiny')
    dg = synth.make_inputs(c['B'], c['V'], st, regions=c['R'], seq_len=c['T'], seed=c['in_seed'], grid=(2, 3))
    p = O.to_params(w)
    with torch.no_grad():
        logp, _, _ = O.forward_xe(p, ids_for(c['V']), tt(dg, 'fc_feats'), tt(dg, 'att_feats'),
                                  tt(dg, 'cpt_words'), tt(dg, 'captions'), tt(dg, 'senti_labels'))
    np.testing.assert_allclose(logp.numpy(), g['grid/xe_logp'], atol=2e-5)


def test_tiny_dropout_masks_and_grads(golden):
    g = golden('tiny')
    c, st, w, d, s2s = case_setup('tiny')
    ids = ids_for(c['V'])
    p = O.to_params(w, torch.float32, True)
    masks = {k: torch.from_numpy(g['drop/mask_' + k]) for k in ('fc', 'att', 'label')}
    masks['out'] = [torch.from_numpy(g['drop/mask_out%d' % i]) for i in range(c['T'])]
    logp, P, _ = O.forward_xe(p, ids, tt(d, 'fc_feats'), tt(d, 'att_feats'), tt(d, 'cpt_words'),
                              tt(d, 'captions'), tt(d, 'senti_labels'), masks=masks)
    np.testing.assert_allclose(logp.detach().numpy(), g['drop/xe_logp'], atol=5e-5)
    loss = O.xe_criterion(logp, tt(d, 'captions')[:, 1:], d['lengths'])
    np.testing.assert_allclose(float(loss.detach()), g['drop/loss'][0], rtol=1e-5)
    loss.backward()
    for k in p:
        key = 'drop/grad/' + k
        if key in g.files:
            ref = g[key]
            np.testing.assert_allclose(p[k].grad.numpy(), ref, atol=1e-4 * np.abs(ref).max() + 1e-7, err_msg=k)
    p = O.to_params(w)
    masks = {k: torch.from_numpy(g['drop_s2s/mask_' + k]) for k in ('cpt', 'words', 'label')}
    masks['out'] = [torch.from_numpy(g['drop_s2s/mask_out%d' % i]) for i in range(c['T'])]
    with torch.no_grad():
        logp, _, _ = O.forward_seq2seq(p, ids, tt(s2s, 'captions'), tt(s2s, 'cpt_words'),
                                       tt(s2s, 'senti_words'), tt(s2s, 'senti_labels'), masks=masks)
    np.testing.assert_allclose(logp.numpy(), g['drop_s2s/logp'], atol=5e-5)


def test_tiny_scheduled_sampling_replay(golden):
    g = golden('tiny')
    c, st, w, d, _ = case_setup('tiny')
    p = O.to_params(w)
    with torch.no_grad():
        logp, _, _ = O.forward_xe(p, ids_for(c['V']), tt(d, 'fc_feats'), tt(d, 'att_feats'),
                                  tt(d, 'cpt_words'), tt(d, 'captions'), tt(d, 'senti_labels'),
                                  fed_tokens=torch.from_numpy(g['ss/fed_tokens']))
    np.testing.assert_allclose(logp.numpy(), g['ss/xe_logp'], atol=2e-5)


def test_cfg1_rollout_and_beam(golden):
    """BASELINE.json configs[0]: B=4, 36x2048, V=10k, T=20."""
    g = golden('cfg1')
    c, st, w, d, _ = case_setup('cfg1')
    ids = ids_for(c['V'])
    p = O.to_params(w)
    a = (tt(d, 'fc_feats'), tt(d, 'att_feats'), tt(d, 'cpt_words'), tt(d, 'senti_words'), tt(d, 'senti_labels'))
    with torch.no_grad():
        seq, lp, mk, P, ws, mar = O.forward_rl(p, ids, *a, c['T'], 1)
    gm = g['rl/greedy_margins']
    n = trusted_prefix(np.pad(gm, ((0, 0), (0, c['T'] - gm.shape[1]))), g['rl/greedy_masks'], 2e-3)
    for b in range(c['B']):
        assert (seq.numpy()[b, :n[b]] == g['rl/greedy_seq'][b, :n[b]]).all()
        np.testing.assert_allclose(lp.numpy()[b, :n[b]], g['rl/greedy_logprobs'][b, :n[b]], atol=1e-4)
    assert n.min() >= 10  # the fixture must actually pin something
    i2w = synth.make_idx2word(c['V'])
    with torch.no_grad():
        for i in range(2):
            caps, sc, _ = O.beam_search(p, ids, i2w, a[0][i], a[1][i], a[3][i], a[4][i:i + 1], 5, 1, c['T'])
            assert caps == list(g['beam/beam5_senti1_caps'][i])
            np.testing.assert_allclose(sc, g['beam/beam5_senti1_scores'][i], atol=1e-3)


def test_cfg1_iteration_digests(golden):
    g = golden('cfg1')
    c, d, s2s, p, logp, P, ws, logp2, ws2, losses = run_iteration('cfg1')
    np.testing.assert_allclose(losses, g['it/losses'], rtol=2e-5)
    np.testing.assert_allclose(logp.detach().numpy()[:, :, :32], g['it/xe_logp'], atol=1e-4)
    tgt = logp.detach().gather(2, tt(d, 'captions')[:, 1:].unsqueeze(2)).squeeze(2).numpy()
    np.testing.assert_allclose(tgt, g['it/xe_logp_tgt'], atol=1e-4)
    for k in p:
        key = 'it/gdig/' + k
        if key not in g.files:
            continue
        assert_digest_close(digest(p[k].grad.numpy()), g[key], k)


def test_beam5_all_64_images_oracle_vs_reference(golden):
    """BASELINE config 2 (beam 5 x 64 images, sentiment words on): the oracle's beam search against the reference's own
    `Captioner.sample` on every image (tests/golden/beam64.npz) - the oracle is what bench-size GPU tests lean on."""
    g = golden('beam64')
    c, st, w, _, _ = case_setup('cfg1')
    idx2word = synth.make_idx2word(c['V'])
    p, ids = O.to_params(w), ids_for(c['V'])
    d = synth.make_inputs(64, c['V'], st, regions=36, seq_len=20, seed=321)
    for i in range(0, 64, 3):               # 22 images here keep the CPU suite short; the GPU test covers all 64
        caps, scores, _ = O.beam_search(p, ids, idx2word, tt(d, 'fc_feats')[i], tt(d, 'att_feats')[i],
                                        tt(d, 'senti_words')[i], tt(d, 'senti_labels')[i:i + 1], 5, 1, 20)
        assert list(caps) == [str(x) for x in g['beam/beam5_senti1_caps'][i]], i
        np.testing.assert_allclose(scores, g['beam/beam5_senti1_scores'][i], atol=1e-4)


def test_oracle_reproduces_the_references_first_rl_training_iteration(golden):
    """models/decoder.py:52-167 for ONE 'fact' batch in training mode (tests/golden/det_train.npz, `dt1/*`: a fresh
    reference Detector, dropout 0): with the reference's multinomial draws and scheduled-sampling tokens replayed, the
    oracle (captioner restatement + CIDEr-D restatement + the frozen helper nets) reproduces the sampled and greedy
    roll-outs and all seven entries of the loss dictionary.  (The GPU test replays the same fixture through the product.)"""
    from insenticap_model_amd.helper_nets import SentenceSentimentClassifier, SentimentDetector
    from insenticap_model_amd.rewards import get_cls_reward
    from oracle.ciderd_oracle import CiderDOracle
    from test_detector import load_helper
    g = golden('det_train')
    V, Tn, B = 64, 8, 4
    st = dict(synth.TINY_SETTINGS, **synth.HELPER_SETTINGS)
    st['dropout_p'] = 0.0
    idx2word = synth.make_idx2word(V)
    ids = ids_for(V)
    p = O.to_params(synth.make_weights(V, st, seed=1))
    senti_det = load_helper(SentimentDetector(synth.SENTIMENT_CATEGORIES, st), 51)
    sent_cls = load_helper(SentenceSentimentClassifier(idx2word, synth.SENTIMENT_CATEGORIES, st), 52)
    batches, split = synth.make_rl_batches(2, B, V, st, seq_len=Tn)
    fns, fc, att, (caps, lengths), cpts, sentis, gts = batches[0]
    t = torch.from_numpy
    fc, att, caps, cpts, sentis = t(fc), t(att), t(caps), t(cpts), t(sentis)
    n_data = 2                                    # the reference divides by len(data) = len((fact, seq2seq)) = 2
    with torch.no_grad():
        labels = senti_det.sample(att, 0.7)[0]
        att3 = att.reshape(B, -1, att.shape[-1])
        draws = t(g['dt/draws0'])
        seq, lp, mk, P, _, _ = O.forward_rl(p, ids, fc, att3, cpts, sentis, labels, Tn, 0, replay=draws)
        gseq, _, gmk, _, _, _ = O.forward_rl(p, ids, fc, att3, cpts, sentis, labels, Tn, 1)
    n_steps = g['dt1/sample_seq'].shape[1]
    assert (seq.numpy()[:, :n_steps] == g['dt1/sample_seq']).all() and (gseq.numpy()[:, :g['dt1/greedy_seq'].shape[1]] == g['dt1/greedy_seq']).all()
    np.testing.assert_allclose(lp.numpy()[:, :n_steps], g['dt1/sample_logprobs'], atol=2e-5)
    np.testing.assert_allclose(mk.numpy()[:, :n_steps], g['dt1/sample_masks'])
    got = {'da_loss': float(O.domain_align_loss(P.cpt, P.fc_raw)) / n_data}
    # CIDEr-D reward: document frequencies over every image of the split, references of this batch
    allcaps = {}
    for v in split.values():
        allcaps.update(v)
    cd = CiderDOracle(list(allcaps.values()), ids.sos, ids.eos)
    fact = np.array(cd.self_critical_reward(seq.numpy(), gseq.numpy(), [gts[fn] for fn in fns]))
    got['fact_reward'] = float(fact.mean()) / n_data
    cls = get_cls_reward(seq, mk, gseq, gmk, labels, sent_cls, on_device=True)
    got['cls_reward'] = float(cls.mean(-1).mean(-1)) / n_data
    rewards = torch.from_numpy(np.repeat(fact[:, None], Tn, 1)).float() + 0.4 * cls
    got['all_rewards'] = float(rewards.mean(-1).mean(-1)) / n_data
    got['cap_loss'] = float(O.reward_criterion(lp, mk, rewards)) / n_data
    with torch.no_grad():
        xl = sent_cls(caps[:, 1:], lengths)[0].softmax(dim=-1).argmax(dim=-1)
        logp, _, _ = O.forward_xe(p, ids, fc, att3, cpts, caps, xl, fed_tokens=t(g['dt/fed_xe0']))
        got['xe_loss'] = float(O.xe_criterion(logp, caps[:, 1:], list(lengths))) / n_data
        s = synth.make_inputs(3, V, st, regions=6, seq_len=Tn, seed=77)
        logp2, _, _ = O.forward_seq2seq(p, ids, t(s['captions']), t(s['cpt_words']), t(s['senti_words']),
                                        t(s['senti_labels']), fed_tokens=t(g['dt/fed_s2s0']))
        got['seq2seq_loss'] = float(O.xe_criterion(logp2, t(s['captions'])[:, 1:], list(s['lengths']))) / n_data
    for k, v in got.items():
        np.testing.assert_allclose(v, g['dt1/loss_' + k][0], rtol=2e-4, atol=2e-6, err_msg=k)


def test_oracle_reproduces_the_references_senti_branch_in_evaluation(golden):
    """models/decoder.py:52-167 with data_type 'senti', training False (tests/golden/det_senti.npz, `dse/*`): labels from
    the image sentiment detector, no CIDEr reward (`fact_reward = 0`), no XE and no seq2seq term - both batches of the
    fixture; with the reference's multinomial draws replayed the oracle reproduces the four entries of the dictionary.
    (The GPU test replays training and evaluation through the product.)"""
    from insenticap_model_amd.helper_nets import SentenceSentimentClassifier, SentimentDetector
    from insenticap_model_amd.rewards import get_cls_reward
    from test_detector import load_helper
    g = golden('det_senti')
    V, Tn, B = 64, 8, 4
    st = dict(synth.TINY_SETTINGS, **synth.HELPER_SETTINGS)
    st['dropout_p'] = 0.0
    idx2word = synth.make_idx2word(V)
    ids = ids_for(V)
    p = O.to_params(synth.make_weights(V, st, seed=1))
    senti_det = load_helper(SentimentDetector(synth.SENTIMENT_CATEGORIES, st), 51)
    sent_cls = load_helper(SentenceSentimentClassifier(idx2word, synth.SENTIMENT_CATEGORIES, st), 52)
    batches, _ = synth.make_rl_batches(2, B, V, st, seq_len=Tn, seed=90)
    t = torch.from_numpy
    sums = {'da_loss': 0.0, 'cls_reward': 0.0, 'all_rewards': 0.0, 'cap_loss': 0.0}
    for i, b in enumerate(batches):
        fc, att, cpts, sentis = t(b[1]), t(b[2]), t(b[4]), t(b[5])
        with torch.no_grad():
            labels = senti_det.sample(att, 0.7)[0]
            att3 = att.reshape(B, -1, att.shape[-1])
            seq, lp, mk, P, _, _ = O.forward_rl(p, ids, fc, att3, cpts, sentis, labels, Tn, 0,
                                                replay=t(g['dse/draws%d' % i]))
            gseq, _, gmk, _, _, _ = O.forward_rl(p, ids, fc, att3, cpts, sentis, labels, Tn, 1)
            sums['da_loss'] += float(O.domain_align_loss(P.cpt, P.fc_raw))
            cls = get_cls_reward(seq, mk, gseq, gmk, labels, sent_cls, on_device=True)
            sums['cls_reward'] += float(cls.mean(-1).mean(-1))
            rewards = 0.4 * cls                                  # fact_reward = 0 (decoder.py:100-101), cls_flag 0.4
            sums['all_rewards'] += float(rewards.mean(-1).mean(-1))
            sums['cap_loss'] += float(O.reward_criterion(lp, mk, rewards))
    n_data = 1                                                   # len(data) = len((senti_loader,)) in evaluation
    for k, v in sums.items():
        np.testing.assert_allclose(v / n_data, g['dse/loss_' + k][0], rtol=2e-4, atol=2e-6, err_msg=k)
