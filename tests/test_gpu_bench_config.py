"""Parity at the configurations that are benchmarked (BASELINE.json configs[2] and [4], and bench.py's own headline
workload), not only at the small ones: the B=4096 greedy roll-out in auto mode (256-row split-f16 tiles, token /
sentiment-word tables, weights scope), roll-outs on every side of the GEMM-engine thresholds, beam 5 over all 64
images against the reference's own output, and one full-size RL iteration's loss dictionary.  pytest -m gpu."""
import numpy as np
import pytest
import torch

from conftest import case_setup, trusted_prefix
from insenticap_model_amd import Captioner, ops, synth

pytestmark = pytest.mark.gpu
LOGP_TOL = 1e-4
KEYS = ('fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')


def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda:0')


def make_cfg1():
    c, st, w, _, _ = case_setup('cfg1')
    cap = Captioner(synth.make_idx2word(c['V']), synth.SENTIMENT_CATEGORIES, st)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    cap.to(dev()).eval()
    return cap, c, st, w


def oracle_greedy(w, c, d, Tn):
    from oracle import captioner_oracle as O
    p = O.to_params(w)
    oid = O.Ids(synth.make_idx2word(c['V']), synth.SENTIMENT_CATEGORIES)
    ca = [torch.from_numpy(np.asarray(d[k])) for k in KEYS]
    with torch.no_grad():
        oseq, olp, omk, _, _, margins = O.forward_rl(p, oid, *ca, Tn, 1)
    return oseq.numpy(), olp.numpy(), omk.numpy(), margins.numpy()


def check_rollout(seq, lp, mk, oseq, olp, omk, margins, Tn):
    n = trusted_prefix(margins, omk, 2e-3)
    assert n.mean() >= 0.8 * Tn
    seq, lp, mk = seq.cpu().numpy(), lp.cpu().numpy(), mk.cpu().numpy()
    # vectorised over rows: compare only each row's trusted prefix
    cols = np.arange(Tn)[None, :] < n[:, None]
    assert (seq[cols] == oseq[cols]).all(), 'rows with a token mismatch: %s' % np.nonzero(
        ((seq != oseq) & cols).any(1))[0][:8]
    assert (mk[cols] == omk[cols]).all()
    err = float(np.abs(lp[cols] - olp[cols]).max())
    assert err < LOGP_TOL, err
    return err


@pytest.mark.parametrize('B', [4096, 16384])
def test_greedy_at_the_bench_batch_in_auto_mode_is_the_bench_path_and_matches_the_oracle(B):
    """bench.py's headline workload: B captions per step (16384 is its default since round 3, 4096 was rounds 1-2
    and stays in its batch sweep), R=36x2048, V=10k, T=20, auto engine selection.  Asserts that the launches really
    went out on the split-f16 path AND on its 256x128 eight-wave tile (both LSTM cells need >3328 rows for that),
    then holds the roll-out token-exact against the CPU oracle on trusted prefixes - every one of the B rows.  At
    16384 rows the feature tensor is 4.8 GB: byte offsets pass 2^32, so this is also the 64-bit-indexing case."""
    import bench
    assert bench.DEFAULT_BATCH == 16384                               # the default this test pins
    cap, c, st, w = make_cfg1()
    Tn = 20
    d = synth.make_inputs(B, c['V'], st, regions=36, seq_len=Tn, seed=B)
    a = [torch.from_numpy(np.asarray(d[k])).to(dev()) for k in KEYS]
    lib = ops._lib.load()
    assert ops.set_h3_mode(1) == 1                                   # auto is the default and stays
    h3_0, h3x_0 = lib.isc_h3_launches(), lib.isc_h3x_launches()
    with torch.no_grad():
        seq, lp, mk = cap(*a, Tn, 1, mode='rl')
    torch.cuda.synchronize()
    n_h3, n_h3x = lib.isc_h3_launches() - h3_0, lib.isc_h3x_launches() - h3x_0
    assert n_h3 >= 5 * Tn, n_h3            # classifier, 2 LSTM cells, h-projections, gate sum - every step
    assert n_h3x >= 2 * Tn, n_h3x          # both LSTM cells on the 256-row kernel, every step
    assert cap._tab_cache.get(False) is not None and cap._senti_tab_cache.get(False) is not None   # token / sentiment-word tables (split-f16 engine's slot)
    del a
    oseq, olp, omk, margins = oracle_greedy(w, c, d, Tn)
    check_rollout(seq, lp, mk, oseq, olp, omk, margins, Tn)


@pytest.mark.parametrize('B', [192, 640, 1408, 2304])
def test_greedy_on_every_side_of_the_engine_thresholds(B):
    """Auto mode picks engines per launch from the tile count: B=192 stays on fp32 split-K, 640 puts the classifier
    on split-f16, 1408 adds the LSTM cells (128-row tiles) and the 64-row projections, 2304 everything large."""
    cap, c, st, w = make_cfg1()
    Tn = 20
    d = synth.make_inputs(B, c['V'], st, regions=36, seq_len=Tn, seed=B)
    a = [torch.from_numpy(np.asarray(d[k])).to(dev()) for k in KEYS]
    with torch.no_grad():
        seq, lp, mk = cap(*a, Tn, 1, mode='rl')
    torch.cuda.synchronize()
    oseq, olp, omk, margins = oracle_greedy(w, c, d, Tn)
    check_rollout(seq, lp, mk, oseq, olp, omk, margins, Tn)


def test_beam5_all_64_images_vs_the_reference(golden):
    """BASELINE config 2 in full: beam 5, 64 images, sentiment words on, decoding_constraint=1, T=20.  Every image's
    five captions (order included) and fp64 scores against what the reference's own one-image `sample` returned
    (tests/golden/beam64.npz).  The smallest gap between two neighbouring beam scores in that fixture is 1.0e-3,
    two orders above fp32 summation noise, so the comparison is unconditional."""
    g = golden('beam64')
    cap, c, st, w = make_cfg1()
    n, Tn = 64, 20
    d = synth.make_inputs(n, c['V'], st, regions=36, seq_len=Tn, seed=321)
    t = lambda k: torch.from_numpy(np.asarray(d[k])).to(dev())
    caps, scores, _ = cap.sample_batch(t('fc_feats'), t('att_feats'), t('senti_words'), t('senti_labels'), 5, 1, Tn)
    ref_caps, ref_scores = g['beam/beam5_senti1_caps'], g['beam/beam5_senti1_scores']
    assert np.abs(np.diff(ref_scores, axis=1)).min() > 5e-4
    bad = [i for i in range(n) if caps[i] != [str(x) for x in ref_caps[i]]]
    assert not bad, bad
    np.testing.assert_allclose(np.asarray(scores), ref_scores, atol=2e-4)
    # the reference's one-image API on a few of them (5 rows take the small-M kernels, 320 rows the batched ones)
    for i in (0, 17, 63):
        cp, sc = cap.sample(t('fc_feats')[i], t('att_feats')[i], t('senti_words')[i], t('senti_labels')[i:i + 1],
                            5, 1, Tn)
        assert cp == [str(x) for x in ref_caps[i]], i
        np.testing.assert_allclose(sc, ref_scores[i], atol=2e-4)


def test_detector_forward_b512_fullsize_vs_the_reference(golden):
    """BASELINE config 4's iteration at full size (B=512, V=10k, T=20, 6x6x2048 grid), evaluation mode, one batch:
    sampled roll-out (the reference's raw multinomial draws replayed), greedy roll-out, CIDEr-D and classifier
    rewards, RL / XE / domain-align losses - the 6-key dictionary of models/decoder.py:52-180 against the
    reference's own (tests/golden/det512.npz), plus the per-row quantities it is made of."""
    from insenticap_model_amd.detector import Detector
    from test_detector import load_helper
    g = golden('det512')
    V, Tn, B = 10000, 20, 512
    st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
    det = Detector(synth.make_idx2word(V), Tn, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
    det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
    load_helper(det.senti_detector, 51)
    load_helper(det.sent_senti_cls, 52)
    det.to(dev())
    batches, split = synth.make_rl_batches(1, B, V, st, grid=(6, 6), seq_len=Tn, seed=60)
    det.set_ciderd_scorer(split)
    b = batches[0]
    t = torch.from_numpy
    item = (b[0], t(b[1]), t(b[2]), (t(b[3][0]), b[3][1]), t(b[4]), t(b[5]), b[6])
    orig = det.captioner.forward_rl
    got = {}

    def replay_rl(*a, **k):
        greedy = bool(k.get('sample_max', a[-1] if len(a) >= 7 else 1))
        if not greedy:
            k['_replay'] = torch.from_numpy(g['det/draws']).to(dev())
        r = orig(*a, **k)
        got['greedy' if greedy else 'sample'] = [x.detach().cpu().numpy() for x in r]
        got.setdefault('labels', a[4].cpu().numpy())
        return r
    det.captioner.forward_rl = replay_rl
    losses = det(([item],), 'fact', False)
    assert set(losses) == {'da_loss', 'fact_reward', 'cls_reward', 'all_rewards', 'cap_loss', 'xe_loss'}
    # image sentiment labels (conv net on MIOpen vs the reference on CPU; a label may only differ on a score that sits
    # on the 0.7 threshold or on a near-tie of the two best classes)
    lab_diff = np.nonzero(got['labels'] != g['det/senti_labels'])[0]
    assert len(lab_diff) == 0 or (np.abs(g['det/senti_scores'][lab_diff] - 0.7) < 1e-4).all(), lab_diff
    # sampled roll-out: replayed draws -> identical tokens and masks, log-probs within the bound
    s_seq, s_lp, s_mk = got['sample']
    live = g['det/sample_masks'] > 0
    assert (s_seq == g['det/sample_seq']).all() and (s_mk == g['det/sample_masks']).all()
    assert np.abs(s_lp - g['det/sample_logprobs'])[live].max() < LOGP_TOL
    # greedy roll-out: token-exact on trusted prefixes; rows that flipped on a near-tie are counted
    g_seq, g_lp, g_mk = got['greedy']
    n = trusted_prefix(g['det/greedy_margins'], g['det/greedy_masks'], 2e-3)
    cols = np.arange(Tn)[None, :] < n[:, None]
    assert (g_seq[cols] == g['det/greedy_seq'][cols]).all()
    flipped = int((g_seq != g['det/greedy_seq']).any(1).sum())
    assert flipped <= 3, flipped
    # the dictionary: keys untouched by the greedy baseline are tight; the reward-dependent ones get
    # 10/B (CIDEr-D's range over the batch mean) per flipped greedy row on top
    for k in ('da_loss', 'xe_loss', 'cls_reward'):
        np.testing.assert_allclose(losses[k], g['det/loss_' + k][0], rtol=2e-4, atol=2e-6, err_msg=k)
    slack = flipped * 10.0 / B
    for k in ('fact_reward', 'all_rewards', 'cap_loss'):
        np.testing.assert_allclose(losses[k], g['det/loss_' + k][0], rtol=2e-4, atol=2e-5 + slack, err_msg=k)


def test_config4_rl_training_iteration_at_full_size_vs_reference(golden):
    """BASELINE config 4's TRAINING iteration at full size: TWO Detector.forward(data, 'fact', True) calls on two fact
    batches with B = 512, V = 10k, T = 20, 6x6x2048 grid, an 80-caption seq2seq batch, dropout 0 (models/decoder.py:52-180
    incl. the update at :161-167) against the reference's own run (tests/golden/det512_train.npz).  Replayed from the
    reference: the sampled roll-out's multinomial draws, the tokens the XE / seq2seq unrolls fed under scheduled sampling
    - and the greedy baseline's tokens and masks, so that the REINFORCE term carries the reference's own reward
    (self_critical/utils.py:56-83: a greedy near-tie flipped by fp32 reassociation would change CIDEr-D of that row by
    up to its range; the product's own greedy roll-out at this size is pinned, with counted flips, by
    test_detector_forward_b512_fullsize_vs_the_reference above).  Checked per iteration at SURVEY 8(d)'s bar with no
    allowance for flipped greedy rows: the 7-key dictionary of both iterations, all 40 clamped gradients of iteration 1
    within 1e-4 of the tensor's largest gradient (64 strided samples + l2; iteration 2, which starts from each side's own
    post-step weights: 1e-3), and after the SECOND step - whose Adam moments are no longer a pure sign test - every
    parameter."""
    from insenticap_model_amd.detector import Detector
    from test_detector import load_helper
    g = golden('det512_train')
    V, Tn, B, Bs = 10000, 20, 512, 80
    st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
    st['dropout_p'] = 0.0
    det = Detector(synth.make_idx2word(V), Tn, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
    det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
    load_helper(det.senti_detector, 51)
    load_helper(det.sent_senti_cls, 52)
    det.to(dev())
    t = torch.from_numpy
    s = synth.make_inputs(Bs, V, st, regions=6, seq_len=Tn, seed=79)
    scs = [((t(s['captions']), s['lengths']), t(s['cpt_words']), t(s['senti_words']), t(s['senti_labels']))]
    cap = det.captioner
    o_rl, o_xe, o_s2s, o_pair = cap.forward_rl, cap.forward_xe, cap.forward_seq2seq, cap.forward_xe_seq2seq
    n = {'rl': 0, 'greedy': 0, 'xe': 0, 's2s': 0}
    cur = {'sfx': ''}

    def fed_as_captions(key):
        fed = torch.from_numpy(g[key + cur['sfx']]).to(dev())
        return torch.cat([fed, fed[:, -1:]], dim=1)

    def replay_rl(*a, **k):
        if not k.get('sample_max', a[-1] if len(a) >= 7 else 1):
            k['_replay'] = torch.from_numpy(g['d5t/draws' + cur['sfx']]).to(dev())
            n['rl'] += 1
            return o_rl(*a, **k)
        n['greedy'] += 1                      # the reference's greedy baseline, as it returned it
        seq = torch.from_numpy(g['d5t/greedy_seq' + cur['sfx']]).to(dev())
        mk = torch.from_numpy(g['d5t/greedy_masks' + cur['sfx']]).to(dev())
        return seq, torch.zeros_like(mk), mk

    def replay_xe(fc, att, cpts, caps, labels, ss_prob=0.0, **k):
        assert ss_prob == 0.5
        n['xe'] += 1
        return o_xe(fc, att, cpts, fed_as_captions('d5t/fed_xe'), labels, 0.0, _targets=caps[:, 1:], **k)

    def replay_s2s(caps, cpts, sentis, labels, ss_prob=0.0, **k):
        assert ss_prob == 0.25
        n['s2s'] += 1
        return o_s2s(fed_as_captions('d5t/fed_s2s'), cpts, sentis, labels, 0.0, _targets=caps[:, 1:], **k)

    def replay_pair(fc, att, cpts, caps, labels, ss_prob, s_caps, s_cpts, s_sentis, s_labels, s_ss_prob=None, **k):
        # both unrolls through the merged step chain (Captioner.forward_xe_seq2seq): the same fed tokens
        assert ss_prob == 0.5 and s_ss_prob == 0.25
        n['xe'] += 1
        n['s2s'] += 1
        return o_pair(fc, att, cpts, fed_as_captions('d5t/fed_xe'), labels, 0.0, fed_as_captions('d5t/fed_s2s'),
                      s_cpts, s_sentis, s_labels, 0.0, _targets=caps[:, 1:], _s_targets=s_caps[:, 1:], **k)
    cap.forward_rl, cap.forward_xe, cap.forward_seq2seq = replay_rl, replay_xe, replay_s2s
    cap.forward_xe_seq2seq = replay_pair

    def digest(x):
        flat = x.reshape(-1).astype(np.float64)
        idx = (np.arange(64, dtype=np.int64) * 2654435761 % flat.size)
        return np.concatenate([[flat.sum(), np.abs(flat).sum(), np.sqrt((flat ** 2).sum())], flat[idx]])
    for it, seed in enumerate((60, 61)):
        sfx = cur['sfx'] = '' if it == 0 else '2'
        batches, split = synth.make_rl_batches(1, B, V, st, grid=(6, 6), seq_len=Tn, seed=seed)
        det.set_ciderd_scorer(split)
        b = batches[0]
        item = (b[0], t(b[1]), t(b[2]), (t(b[3][0]), b[3][1]), t(b[4]), t(b[5]), b[6])
        losses = det(([item], scs), 'fact', True)
        assert n == {'rl': it + 1, 'greedy': it + 1, 'xe': it + 1, 's2s': it + 1}
        assert set(losses) == {'da_loss', 'fact_reward', 'cls_reward', 'all_rewards', 'cap_loss', 'xe_loss', 'seq2seq_loss'}
        for k in losses:                      # rewards are the reference's own: no slack for flipped greedy rows
            np.testing.assert_allclose(losses[k], g['d5t/loss%s_%s' % (sfx, k)][0], rtol=2e-4, atol=2e-6, err_msg=k)
        checked = 0
        for k, q in cap.named_parameters():
            key = 'd5t/grad%s/%s' % (sfx, k)
            if key not in g.files:
                continue
            ref, gmax = g[key], float(g['d5t/gmax%s/%s' % (sfx, k)][0])
            got = digest(q.grad.cpu().numpy())
            # (+ 1e-8: the two alpha biases' true gradient is 0 - softmax is shift invariant - and the reference's autograd
            # leaves rounding noise of ~3e-9 there.)
            # Iteration 1 starts from the reference's exact weights: SURVEY 8(d)'s 1e-4, no allowance.  Iteration 2 starts
            # from each side's OWN weights after step 1 - a first Adam step moves every element by +-lr whatever its
            # gradient's size, so elements whose gradient is rounding noise differ by 2 lr = 8e-5 between the two sides
            # (the post-step check below bounds how many) - and a ReLU pre-activation of ~0 in att2att / att_embed then
            # flips for a few of the 18 432 x 512 region activations: 1e-3 there (measured worst element 9e-4).
            tol = (1e-4 if it == 0 else 1e-3) * gmax + 1e-8
            np.testing.assert_allclose(got[3:], ref[3:], atol=tol, err_msg='%s (iteration %d)' % (k, it + 1))
            assert abs(got[2] - ref[2]) <= 1e-3 * ref[2] + 1e-7, (k, got[2], ref[2])          # l2 norm
            checked += 1
        assert checked == 40
    # after the SECOND step: m / (sqrt(v) + eps) of two different gradients - an element steps by lr (4e-5) only where both
    # gradients agree in sign, so the parameters are a finer pin than after a first step (always +-lr).  Elements whose
    # gradient is ~0 on either iteration may still differ by a step; everything else to a small fraction of lr.
    for k, q in cap.state_dict().items():
        ref = g['d5t/after2/' + k]
        got = digest(q.cpu().numpy())
        d = np.abs(got[3:] - ref[3:])
        assert d.max() <= 2 * 2 * 4e-5 * 1.01, k
        gk = 'd5t/gmax2/' + k
        if gk in g.files and float(g[gk][0]) > 1e-6:          # (the alpha biases' true gradient is 0: Adam steps on noise)
            assert np.median(d) <= 1e-6, (k, float(np.median(d)))
            assert (d > 4e-6).mean() <= 0.1, (k, float((d > 4e-6).mean()))
