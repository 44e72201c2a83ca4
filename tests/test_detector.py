"""Helper nets (CPU + GPU) and the RL wrapper `Detector` (GPU) vs golden vectors produced by the
reference's own models/decoder.py, models/sentiment_detector.py and models/sent_senti_cls.py."""
import numpy as np
import pytest
import torch

from insenticap_model_amd import synth
from insenticap_model_amd.helper_nets import SentenceSentimentClassifier, SentimentDetector

V, TN, B = 64, 8, 4
ST = dict(synth.TINY_SETTINGS, **synth.HELPER_SETTINGS)


def load_helper(mod, seed):
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_module_weights(shapes, seed).items()})
    return mod.eval()


def test_helper_net_state_dict_layout():
    sd = SentimentDetector(synth.SENTIMENT_CATEGORIES, ST).state_dict()
    assert list(sd) == ['convs.conv_0.weight', 'convs.conv_0.bias', 'convs.conv_1.weight', 'convs.conv_1.bias',
                        'senti_conv.weight', 'senti_conv.bias', 'output.0.weight', 'output.0.bias',
                        'output.1.weight', 'output.1.bias']
    sc = SentenceSentimentClassifier(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, ST).state_dict()
    assert list(sc) == ['word_embed.0.weight', 'rnn.weight_ih_l0', 'rnn.weight_hh_l0', 'rnn.bias_ih_l0',
                        'rnn.bias_hh_l0', 'excitation.0.weight', 'excitation.0.bias', 'excitation.2.weight',
                        'excitation.2.bias', 'sent_senti_cls.0.weight', 'sent_senti_cls.0.bias',
                        'sent_senti_cls.3.weight', 'sent_senti_cls.3.bias']


def test_helper_nets_vs_reference_cpu(golden):
    g = golden('detector')
    det = load_helper(SentimentDetector(synth.SENTIMENT_CATEGORIES, ST), 51)
    cls = load_helper(SentenceSentimentClassifier(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, ST), 52)
    batches, _ = synth.make_rl_batches(2, B, V, ST, seq_len=TN)
    for i, b in enumerate(batches):
        att = torch.from_numpy(b[2])
        labels, maps, names, scores = det.sample(att, 0.7)
        assert (labels.numpy() == g['senti_det/labels%d' % i]).all()
        np.testing.assert_allclose(scores.numpy(), g['senti_det/scores%d' % i], atol=1e-5)
        np.testing.assert_allclose(maps.numpy(), g['senti_det/maps%d' % i], atol=1e-4)
        with torch.no_grad():
            logits, _ = det(att)
            pred, w = cls(torch.from_numpy(b[3][0])[:, 1:], b[3][1])
        np.testing.assert_allclose(logits.numpy(), g['senti_det/logits%d' % i], atol=1e-4)
        np.testing.assert_allclose(pred.numpy(), g['sent_cls/pred%d' % i], atol=1e-4)
        np.testing.assert_allclose(w.numpy(), g['sent_cls/weights%d' % i], atol=1e-5)
        assert names == [synth.SENTIMENT_CATEGORIES[int(x)] for x in labels]
    # threshold semantics: impossible threshold => everything neutral
    labels, _, _, _ = det.sample(torch.from_numpy(batches[0][2]), 2.0)
    assert (labels == det.neu_idx).all()


def _make_detector(dev):
    from insenticap_model_amd.detector import Detector
    d = Detector(synth.make_idx2word(V), TN, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, ST)
    d.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, ST, seed=1).items()})
    load_helper(d.senti_detector, 51)
    load_helper(d.sent_senti_cls, 52)
    return d.to(dev)


def _tensors(b):
    t = torch.from_numpy
    return (b[0], t(b[1]), t(b[2]), (t(b[3][0]), b[3][1]), t(b[4]), t(b[5]), b[6])


@pytest.mark.gpu
def test_detector_forward_eval_vs_reference(golden):
    """Detector.forward(data, 'fact', training=False): the 6-key loss dictionary (sampled + greedy
    roll-outs, CIDEr-D and classifier rewards, RL / XE / domain-align losses) vs the reference's."""
    g = golden('detector')
    dev = torch.device('cuda:0')
    det = _make_detector(dev)
    batches, split = synth.make_rl_batches(2, B, V, ST, seq_len=TN)
    det.set_ciderd_scorer(split)
    # replay the reference's multinomial draws in the sampled roll-outs
    orig = det.captioner.forward_rl
    calls = {'n': 0}

    def replay_rl(*a, **k):
        if not k.get('sample_max', a[-1] if len(a) >= 7 else 1):     # Detector passes sample_max by keyword
            k['_replay'] = torch.from_numpy(g['det/draws%d' % calls['n']]).to(dev)
            calls['n'] += 1
        return orig(*a, **k)
    det.captioner.forward_rl = replay_rl
    losses = det(([_tensors(b) for b in batches],), 'fact', False)
    assert calls['n'] == 2
    assert set(losses) == {'da_loss', 'fact_reward', 'cls_reward', 'all_rewards', 'cap_loss', 'xe_loss'}
    for k, v in losses.items():
        np.testing.assert_allclose(v, g['det/loss_' + k][0], rtol=2e-4, atol=2e-5, err_msg=k)
    # Detector.sample: one image, beam search + detected sentiment
    # Detector.sample (models/decoder.py:182-192): every image of both batches - its three beam captions and the
    # detected sentiment - against what the reference returned
    k = 0
    for b in batches:
        for i in range(B):
            caps, sentis = det.sample(torch.from_numpy(b[1][i]).to(dev), torch.from_numpy(b[2][i]).to(dev),
                                      torch.from_numpy(b[5][i]).to(dev), beam_size=3, decoding_constraint=1)
            assert list(caps) == [str(x) for x in g['det/sample_caps'][k] if str(x)], (k, caps)
            assert sentis[0] == str(g['det/sample_sentis'][k]), k
            k += 1


@pytest.mark.gpu
def test_detector_training_iteration_runs():
    """training=True: sampled roll-out with REINFORCE gradients + greedy + XE (ss 0.5) + seq2seq (ss 0.25)
    + backward + clamp + Adam. Stochastic (dropout, sampling), so checked for structure, not values."""
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    det = _make_detector(dev)
    batches, split = synth.make_rl_batches(2, B, V, ST, seq_len=TN)
    det.set_ciderd_scorer(split)
    s = synth.make_inputs(3, V, ST, regions=6, seq_len=TN, seed=77)
    t = torch.from_numpy
    scs = [((t(s['captions']), s['lengths']), t(s['cpt_words']), t(s['senti_words']), t(s['senti_labels']))]
    before = {k: v.detach().clone() for k, v in det.captioner.state_dict().items()}
    helper_before = {k: v.detach().clone() for k, v in det.sent_senti_cls.state_dict().items()}
    losses = det(([_tensors(b) for b in batches], scs), 'fact', True)
    assert set(losses) == {'da_loss', 'fact_reward', 'cls_reward', 'all_rewards', 'cap_loss', 'xe_loss',
                           'seq2seq_loss'}
    assert all(np.isfinite(v) for v in losses.values())
    after = det.captioner.state_dict()
    moved = [k for k in before if not torch.equal(before[k], after[k])]
    assert len(moved) >= 38                                   # every trained tensor moved (alpha biases: zero grad)
    for k in before:
        assert float((after[k] - before[k]).abs().max()) <= 2 * 4e-5 * 1.01 + 1e-7, k   # two Adam steps of lr
    for k, v in det.sent_senti_cls.state_dict().items():      # helper nets stay frozen
        assert torch.equal(v, helper_before[k])
    assert not det.sent_senti_cls.training and not det.senti_detector.training


@pytest.mark.gpu
def test_detector_training_iterations_match_the_reference(golden):
    """Detector.forward(data, 'fact', training=True) - the RL training iteration of train_rl.py (models/decoder.py:
    52-180: sampled roll-out with REINFORCE gradients, greedy baseline, CIDEr-D + classifier rewards, XE with ss_prob
    0.5, seq2seq with ss_prob 0.25, backward, clamp, Adam) - two iterations against the reference's own run
    (tests/golden/det_train.npz: dropout_p = 0, the multinomial draws of the sampled roll-outs and the tokens fed under
    scheduled sampling replayed): the 7-key loss dictionary, iteration 2's clamped gradient, every parameter after
    the two steps."""
    from insenticap_model_amd.detector import Detector
    g = golden('det_train')
    dev = torch.device('cuda:0')
    st = dict(ST, dropout_p=0.0)
    det = Detector(synth.make_idx2word(V), TN, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-4}, st)
    det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=1).items()})
    load_helper(det.senti_detector, 51)
    load_helper(det.sent_senti_cls, 52)
    det.to(dev)
    batches, split = synth.make_rl_batches(2, B, V, st, seq_len=TN)
    det.set_ciderd_scorer(split)
    s = synth.make_inputs(3, V, st, regions=6, seq_len=TN, seed=77)
    t = torch.from_numpy
    scs = [((t(s['captions']), s['lengths']), t(s['cpt_words']), t(s['senti_words']), t(s['senti_labels']))]
    cap = det.captioner
    o_rl, o_xe, o_s2s = cap.forward_rl, cap.forward_xe, cap.forward_seq2seq
    n = {'rl': 0, 'xe': 0, 's2s': 0}

    def fed_as_captions(key):
        fed = torch.from_numpy(g[key]).to(dev)
        return torch.cat([fed, fed[:, -1:]], dim=1)            # the unrolls feed captions[:, :-1]

    def replay_rl(*a, **k):
        if not k.get('sample_max', a[-1] if len(a) >= 7 else 1):
            k['_replay'] = torch.from_numpy(g['dt/draws%d' % n['rl']]).to(dev)
            n['rl'] += 1
        return o_rl(*a, **k)

    def replay_xe(fc, att, cpts, caps, labels, ss_prob=0.0, **k):
        assert ss_prob == 0.5                                    # models/decoder.py:139
        out = o_xe(fc, att, cpts, fed_as_captions('dt/fed_xe%d' % n['xe']), labels, 0.0, _targets=caps[:, 1:], **k)
        n['xe'] += 1
        return out

    def replay_s2s(caps, cpts, sentis, labels, ss_prob=0.0, **k):
        assert ss_prob == 0.25                                   # models/decoder.py:155
        out = o_s2s(fed_as_captions('dt/fed_s2s%d' % n['s2s']), cpts, sentis, labels, 0.0, _targets=caps[:, 1:], **k)
        n['s2s'] += 1
        return out
    o_pair = cap.forward_xe_seq2seq

    def replay_pair(fc, att, cpts, caps, labels, ss_prob, s_caps, s_cpts, s_sentis, s_labels, s_ss_prob=None, **k):
        # both unrolls through the merged step chain (Captioner.forward_xe_seq2seq): the same fed tokens
        assert ss_prob == 0.5 and s_ss_prob == 0.25
        out = o_pair(fc, att, cpts, fed_as_captions('dt/fed_xe%d' % n['xe']), labels, 0.0,
                     fed_as_captions('dt/fed_s2s%d' % n['s2s']), s_cpts, s_sentis, s_labels, 0.0, _targets=caps[:, 1:], _s_targets=s_caps[:, 1:], **k)
        n['xe'] += 1
        n['s2s'] += 1
        return out
    cap.forward_rl, cap.forward_xe, cap.forward_seq2seq = replay_rl, replay_xe, replay_s2s
    cap.forward_xe_seq2seq = replay_pair
    losses = det(([_tensors(b) for b in batches], scs), 'fact', True)
    assert n == {'rl': 2, 'xe': 2, 's2s': 2}
    assert set(losses) == {'da_loss', 'fact_reward', 'cls_reward', 'all_rewards', 'cap_loss', 'xe_loss',
                           'seq2seq_loss'}
    for k, v in losses.items():
        np.testing.assert_allclose(v, g['dt/loss_' + k][0], rtol=2e-4, atol=2e-5, err_msg=k)
    checked = 0
    for k, q in cap.named_parameters():
        key = 'dt/grad2/' + k
        if key in g.files:
            ref = g[key]
            np.testing.assert_allclose(q.grad.cpu().numpy(), ref, atol=1e-4 * np.abs(ref).max() + 1e-7, err_msg=k)
            checked += 1
    assert checked >= 38                                          # (the RL iteration trains the gate as well)
    for k, q in cap.state_dict().items():
        ref = g['dt/after/' + k]
        diff = np.abs(q.cpu().numpy() - ref)
        # two Adam steps of lr 4e-4.  A step is lr * m / (sqrt(v) + eps) - it normalises the gradient away - so where
        # a gradient element is small next to its rounding error the two sides may step differently by up to lr; the
        # gradients themselves are held to 1e-4 of the tensor's largest above.  Here: nothing moves further than two
        # full steps from the reference, and the typical element lands on it.
        assert diff.max() <= 2 * 2 * 4e-4 * 1.01, k
        gkey = 'dt/grad2/' + k
        if gkey in g.files and np.abs(g[gkey]).max() > 1e-6:       # (the alpha biases' true gradient is 0: pure noise)
            assert np.median(diff) <= 2e-6, (k, float(np.median(diff)))


def _senti_items(batches, seed):
    """rl_senti collate layout (dataloader.py:93-109): (fns, fc, att, cpts, sentis, senti_labels) - as the generator."""
    rng = np.random.default_rng(seed)
    t = torch.from_numpy
    items = []
    for b in batches:
        labels = rng.integers(0, len(synth.SENTIMENT_CATEGORIES), size=len(b[0])).astype(np.int64)
        items.append((b[0], t(b[1]), t(b[2]), t(b[4]), t(b[5]), t(labels)))
    return items


@pytest.mark.gpu
def test_detector_senti_branch_matches_the_reference(golden):
    """Detector.forward((senti_loader, scs_loader), 'senti', training) - train_rl.py:232-235; models/decoder.py:69-71,
    100-101: the sentiment labels come from the batch when training (from the image detector otherwise), there is no
    CIDEr reward (`fact_reward = 0`) and no XE term.  Against the reference's own run (tests/golden/det_senti.npz,
    dropout_p = 0, draws and fed tokens replayed): training - the 5-key dictionary, iteration 2's clamped gradient, every
    parameter after the two steps; evaluation - the 4-key dictionary."""
    from insenticap_model_amd.detector import Detector
    g = golden('det_senti')
    dev = torch.device('cuda:0')
    st = dict(ST, dropout_p=0.0)
    batches, _ = synth.make_rl_batches(2, B, V, st, seq_len=TN, seed=90)
    s = synth.make_inputs(3, V, st, regions=6, seq_len=TN, seed=78)
    t = torch.from_numpy
    scs = [((t(s['captions']), s['lengths']), t(s['cpt_words']), t(s['senti_words']), t(s['senti_labels']))]
    for i, it in enumerate(_senti_items(batches, 91)):
        assert (it[5].numpy() == g['ds/labels%d' % i]).all()

    def make():
        det = Detector(synth.make_idx2word(V), TN, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-4}, st)
        det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=1).items()})
        load_helper(det.senti_detector, 51)
        load_helper(det.sent_senti_cls, 52)
        return det.to(dev)

    def hook(det, prefix, n):
        cap = det.captioner
        o_rl, o_s2s = cap.forward_rl, cap.forward_seq2seq

        def replay_rl(*a, **k):
            if not k.get('sample_max', a[-1] if len(a) >= 7 else 1):
                k['_replay'] = torch.from_numpy(g['%s/draws%d' % (prefix, n['rl'])]).to(dev)
                n['rl'] += 1
            return o_rl(*a, **k)

        def replay_s2s(caps, cpts, sentis, labels, ss_prob=0.0, **k):
            assert ss_prob == 0.25
            fed = torch.from_numpy(g['%s/fed_s2s%d' % (prefix, n['s2s'])]).to(dev)
            n['s2s'] += 1
            return o_s2s(torch.cat([fed, fed[:, -1:]], dim=1), cpts, sentis, labels, 0.0, _targets=caps[:, 1:], **k)

        def no_xe(*a, **k):
            raise AssertionError("the 'senti' branch has no XE unroll (models/decoder.py:131)")
        cap.forward_rl, cap.forward_seq2seq, cap.forward_xe = replay_rl, replay_s2s, no_xe
    # ---- training
    det = make()
    n = {'rl': 0, 's2s': 0}
    hook(det, 'ds', n)
    losses = det((_senti_items(batches, 91), scs), 'senti', True)
    assert n == {'rl': 2, 's2s': 2}
    assert set(losses) == {'da_loss', 'cls_reward', 'all_rewards', 'cap_loss', 'seq2seq_loss'}
    for k, v in losses.items():
        np.testing.assert_allclose(v, g['ds/loss_' + k][0], rtol=2e-4, atol=2e-5, err_msg=k)
    checked = 0
    for k, q in det.captioner.named_parameters():
        key = 'ds/grad2/' + k
        if key in g.files:
            ref = g[key]
            np.testing.assert_allclose(q.grad.cpu().numpy(), ref, atol=1e-4 * np.abs(ref).max() + 1e-7, err_msg=k)
            checked += 1
    assert checked >= 30
    for k, q in det.captioner.state_dict().items():
        diff = np.abs(q.cpu().numpy() - g['ds/after/' + k])
        assert diff.max() <= 2 * 2 * 4e-4 * 1.01, k            # (two Adam steps of lr; see the 'fact' test)
        gkey = 'ds/grad2/' + k
        if gkey in g.files and np.abs(g[gkey]).max() > 1e-6:
            assert np.median(diff) <= 2e-6, (k, float(np.median(diff)))
    # ---- evaluation: labels from the image sentiment detector, no seq2seq pass, nothing trained
    det = make()
    before = {k: v.detach().clone() for k, v in det.captioner.state_dict().items()}
    n = {'rl': 0, 's2s': 0}
    hook(det, 'dse', n)
    losses = det((_senti_items(batches, 91),), 'senti', False)
    assert n == {'rl': 2, 's2s': 0}
    assert set(losses) == {'da_loss', 'cls_reward', 'all_rewards', 'cap_loss'}
    for k, v in losses.items():
        np.testing.assert_allclose(v, g['dse/loss_' + k][0], rtol=2e-4, atol=2e-5, err_msg=k)
    for k, v in det.captioner.state_dict().items():
        assert torch.equal(v, before[k])


@pytest.mark.gpu
def test_classifier_reward_with_the_lengths_on_the_device_equals_the_host_lengths_form():
    """rewards.get_cls_reward (self_critical/utils.py:120-151): the graph-served RL iteration hands the sampled lengths
    over as a DEVICE tensor - the sentence classifier then runs over all T columns with nothing read by the host - where
    the reference (and the eager path) packs to the longest row.  Same rewards: the LSTM is causal and positions at or
    beyond a row's length are masked, so the values agree to rounding and the padding is exactly zero."""
    from insenticap_model_amd.rewards import get_cls_reward
    DEV = torch.device('cuda:0')
    st = dict(synth.TINY_SETTINGS, **synth.HELPER_SETTINGS)
    Vn, Tn, Bn = 64, 12, 33
    cls_net = load_helper(SentenceSentimentClassifier(synth.make_idx2word(Vn), synth.SENTIMENT_CATEGORIES, st), 52).to(DEV)
    cls_net.eval()
    rng = np.random.default_rng(4)
    lens = rng.integers(1, Tn - 2, size=Bn)                 # every row shorter than T: the host form packs to max(lens) < T
    seq = torch.from_numpy(rng.integers(2, Vn, size=(Bn, Tn), dtype=np.int64)).to(DEV)
    mk = (torch.arange(Tn)[None, :] < torch.from_numpy(lens)[:, None]).float().to(DEV)
    labels = torch.from_numpy(rng.integers(0, len(synth.SENTIMENT_CATEGORIES), size=Bn, dtype=np.int64)).to(DEV)
    host = get_cls_reward(seq, mk, None, None, labels, cls_net, sample_lens=lens.tolist(), on_device=True)
    dev = get_cls_reward(seq, mk, None, None, labels, cls_net, sample_lens=mk.sum(-1).to(torch.int32), on_device=True)
    assert host.shape == dev.shape == (Bn, Tn)
    np.testing.assert_allclose(dev.cpu().numpy(), host.cpu().numpy(), rtol=1e-5, atol=1e-7)
    pad = (torch.arange(Tn, device=DEV)[None, :] >= mk.sum(-1, keepdim=True))
    assert float(dev[pad].abs().max()) == 0.0 and float(host[pad].abs().max()) == 0.0
    assert float(dev.abs().max()) > 0.0
