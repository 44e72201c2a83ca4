"""train_graph.RLTrainGraph - the self-critical RL training iteration of Detector.forward(data, 'fact', True)
(models/decoder.py:65-167) from HIP graphs: replays against the same phases run eagerly (bit for bit, with the sampled
roll-out's draws forced and scheduled sampling off: no random numbers), against the plain eager Detector.forward (the
gradient sum in another order: fp32 rounding), counters of replays / captures, and a run with live random draws
(train-mode dropout, sampling, scheduled sampling) that must keep training.  pytest -m gpu."""
import numpy as np
import pytest
import torch

from insenticap_model_amd import synth
from insenticap_model_amd.detector import Detector
from test_detector import load_helper

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
V, TN, B, S = 64, 8, 8, 4
ST = dict(synth.TINY_SETTINGS, **synth.HELPER_SETTINGS)


@pytest.fixture(autouse=True, params=['two_branches', 'merged_chain'])
def unroll_form(request, monkeypatch):
    """Graph-served and eager iterations are compared in the SAME form of the XE / seq2seq unrolls: one chain per unroll
    (the default inside graphs) and the merged step chain (autograd_pair; the default of eager steps)."""
    monkeypatch.setenv('ISC_PAIR_UNROLLS', '1' if request.param == 'merged_chain' else '0')
    return request.param


def make(dropout=0.0, graphs=True, warmup=2):
    st = dict(ST, dropout_p=dropout)
    det = Detector(synth.make_idx2word(V), TN, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-4}, st)
    det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=3).items()})
    load_helper(det.senti_detector, 51)
    load_helper(det.sent_senti_cls, 52)
    det.to(DEV)
    det.train_graphs = graphs
    det._graph_warmup = warmup
    return det


def data(n_iter, B=B):
    st = dict(ST, dropout_p=0.0)
    batches, split = synth.make_rl_batches(n_iter, B, V, st, seq_len=TN, seed=70)
    t = torch.from_numpy
    items = [(b[0], t(b[1]), t(b[2]), (t(b[3][0]), b[3][1]), t(b[4]), t(b[5]), b[6]) for b in batches]
    s = synth.make_inputs(S, V, st, regions=6, seq_len=TN, seed=72)
    scs = [((t(s['captions']), s['lengths']), t(s['cpt_words']), t(s['senti_words']), t(s['senti_labels']))]
    draws = [np.random.default_rng(80 + i).integers(2, V, size=(B, TN), dtype=np.int64) for i in range(n_iter)]
    return items, scs, split, draws


def force_draws(det, draws):
    orig = det.captioner.__dict__.get('_orig_forward_rl') or det.captioner.forward_rl
    det.captioner._orig_forward_rl = orig
    # ONE device tensor per distinct draw matrix, alive as long as the detector: a captured graph keeps reading the
    # address it was captured with (no host copy inside a capture either)
    keep = det.__dict__.setdefault('_test_draws', {})
    dev_draws, n = [keep.setdefault(d.tobytes(), torch.from_numpy(d).to(DEV)) for d in draws], {'i': 0}

    def replay_rl(*a, **k):
        if not k.get('sample_max', a[-1] if len(a) >= 7 else 1):
            k['_replay'] = dev_draws[n['i'] % len(dev_draws)]
            n['i'] += 1
        return orig(*a, **k)
    det.captioner.forward_rl = replay_rl


def run(det, items, scs, split, draws=None):
    det.set_ciderd_scorer(split)
    det.xe_ss_prob = det.seq2seq_ss_prob = 0.0 if draws is not None else det.xe_ss_prob
    out = []
    for i, it in enumerate(items):
        if draws is not None:
            force_draws_for = draws[i:i + 1]
            force_draws(det, force_draws_for)
        if det.train_graphs and det._rl_graph is None:
            from insenticap_model_amd.train_graph import RLTrainGraph
            det._rl_graph = RLTrainGraph(det, warmup=det._graph_warmup)
        out.append(det(([it], scs), 'fact', True))
    torch.cuda.synchronize()
    return out


def run_updating_draws(det, items, scs, split, draws):
    """As run(), but the forced draws live in ONE device tensor whose CONTENT changes every iteration: a captured graph
    reads that address, so its replays sample other tokens each time - and stay comparable with the eager phases."""
    det.set_ciderd_scorer(split)
    det.xe_ss_prob = det.seq2seq_ss_prob = 0.0
    buf = torch.from_numpy(draws[0]).to(DEV)
    det._test_draws = {'buf': buf}
    orig = det.captioner.forward_rl

    def replay_rl(*a, **k):
        if not k.get('sample_max', a[-1] if len(a) >= 7 else 1):
            k['_replay'] = buf
        return orig(*a, **k)
    det.captioner.forward_rl = replay_rl
    out = []
    for it, d in zip(items, draws):
        buf.copy_(torch.from_numpy(d).to(DEV))
        if det.train_graphs and det._rl_graph is None:
            from insenticap_model_amd.train_graph import RLTrainGraph
            det._rl_graph = RLTrainGraph(det, warmup=det._graph_warmup)
        out.append(det(([it], scs), 'fact', True))
    torch.cuda.synchronize()
    return out


def same_params(a, b):
    for (k, p), (_, q) in zip(a.captioner.named_parameters(), b.captioner.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), k


def test_replays_equal_the_eager_phases_bit_for_bit_and_track_plain_eager():
    """One batch geometry, the SAME batch and forced draws every iteration (a captured graph bakes the draws it was
    captured with: forced draws are a test device), six iterations: graph path (two eager, capture, replays) vs the same
    phases never captured: identical parameters and statistics; vs plain eager Detector.forward: within rounding."""
    items, scs, split, draws = data(1)
    items, draws = items * 6, draws * 6
    g = make(graphs=True, warmup=2)
    e = make(graphs=True, warmup=10 ** 6)        # the phases, never captured
    p = make(graphs=False)                        # the plain eager sequence
    og, oe, op_ = run(g, items, scs, split, draws), run(e, items, scs, split, draws), run(p, items, scs, split, draws)
    assert g._rl_graph.captures == 1 and g._rl_graph.replays == 4 and g._rl_graph.eager_steps == 2
    assert e._rl_graph.captures == 0 and e._rl_graph.eager_steps == 6
    same_params(g, e)
    for a, b in zip(og, oe):
        assert a.keys() == b.keys() == set(g._rl_graph.KEYS)
        for k in a:
            assert a[k] == b[k], k
    for a, c in zip(og, op_):
        for k in a:
            np.testing.assert_allclose(a[k], c[k], rtol=2e-4, atol=2e-6, err_msg=k)
    for (k, x), (_, y) in zip(g.captioner.named_parameters(), p.captioner.named_parameters()):
        assert float((x - y).abs().max()) <= 6 * 2 * 4e-4 * 1.01, k


def test_replays_on_new_batches_with_new_draws_equal_the_eager_phases():
    """Every iteration another batch (images, ground truth, captions) AND other sampled tokens, 128 rows x 8 steps = 1024
    fed tokens per unroll - the indexed embedding backward with its position index, whose counters a captured
    hipMemsetAsync node once failed to clear on replay (wrong embedding gradients as soon as the ids changed): graph path
    vs the same phases never captured, identical statistics and parameters after six iterations."""
    n, Bb = 6, 128
    items, scs, split, _ = data(n, B=Bb)
    draws = [np.random.default_rng(500 + i).integers(2, V, size=(Bb, TN), dtype=np.int64) for i in range(n)]
    g = make(graphs=True, warmup=2)
    e = make(graphs=True, warmup=10 ** 6)
    og, oe = run_updating_draws(g, items, scs, split, draws), run_updating_draws(e, items, scs, split, draws)
    assert g._rl_graph.captures == 1 and g._rl_graph.replays == 4
    for a, b in zip(og, oe):
        for k in a:
            assert a[k] == b[k], k
    same_params(g, e)


def test_live_draws_dropout_and_scheduled_sampling_keep_training_from_graphs():
    """Train-mode dropout, on-device sampling and scheduled sampling inside the captured graphs: replays draw afresh
    (the losses of consecutive replays on one batch differ), everything stays finite, the helper nets stay frozen."""
    items, scs, split, _ = data(1)
    det = make(dropout=0.5, graphs=True, warmup=2)
    torch.manual_seed(5)
    helper_before = {k: v.detach().clone() for k, v in det.sent_senti_cls.state_dict().items()}
    outs = run(det, items * 6, scs, split)
    assert det._rl_graph.replays == 4
    assert all(np.isfinite(v) for o in outs for v in o.values())
    assert len({round(o['xe_loss'], 6) for o in outs[2:]}) > 1          # fresh draws in every replay
    assert len({round(o['cap_loss'], 6) for o in outs[2:]}) > 1
    for k, v in det.sent_senti_cls.state_dict().items():
        assert torch.equal(v, helper_before[k])


def test_a_new_batch_geometry_gets_its_own_graphs():
    items, scs, split, draws = data(1)
    det = make(graphs=True, warmup=1)
    run(det, items * 3, scs, split, draws * 3)
    assert det._rl_graph.captures == 1
    st = dict(ST, dropout_p=0.0)
    b2, split2 = synth.make_rl_batches(1, 4, V, st, seq_len=TN, seed=71)      # four images instead of eight
    t = torch.from_numpy
    it2 = [(b[0], t(b[1]), t(b[2]), (t(b[3][0]), b[3][1]), t(b[4]), t(b[5]), b[6]) for b in b2]
    det.set_ciderd_scorer(split2)
    d2 = [np.random.default_rng(9).integers(2, V, size=(4, TN), dtype=np.int64)]
    for _ in range(3):
        force_draws(det, d2)
        out = det((it2, scs), 'fact', True)
    assert det._rl_graph.captures == 2 and all(np.isfinite(v) for v in out.values())


def test_an_xe_training_graph_and_the_rl_graph_take_turns_on_one_captioner():
    """Two training-graph objects on ONE captioner (an XE stage's XETrainGraph next to the Detector's RLTrainGraph): each
    keeps the autograd graph of its captured forward alive, hence the parameters' accumulation nodes on ITS stream - the
    other object's capture used to pull that stream in and the runtime's end-of-capture crashed (a soak run: 600 RL
    iterations, then the first XE capture).  The object that steps takes the captioner over and the other's graphs are
    dropped; it captures again when its turn comes."""
    from insenticap_model_amd.train_graph import XETrainGraph
    items, scs, split, draws = data(1)
    det = make(graphs=True, warmup=1)
    run(det, items * 3, scs, split, draws * 3)
    assert det._rl_graph.captures == 1 and det._rl_graph._geoms
    cap = det.captioner
    g = XETrainGraph(cap, det.cap_optim, det.cap_xe_crit, det.cap_da_crit, grad_clip=0.1, warmup=1)
    fns, fc, att, (caps, lengths), cpts = items[0][:5]
    fact = (None, fc.to(DEV), att.to(DEV), (caps.to(DEV), lengths), cpts.to(DEV))
    labels = torch.zeros(fc.shape[0], dtype=torch.int64, device=DEV)
    (s_caps, s_len), s_cpts, s_sentis, s_labels = scs[0]
    scs_d = ((s_caps.to(DEV), s_len), s_cpts.to(DEV), s_sentis.to(DEV), s_labels.to(DEV))
    for _ in range(3):
        out = g.step(fact, labels, scs_d, 0.0)
    torch.cuda.synchronize()
    assert g.captures == 1 and not det._rl_graph._geoms            # the RL graphs went when the XE object took over
    assert all(np.isfinite(float(v)) for v in out.values())
    outs = run(det, items * 3, scs, split, draws * 3)              # ... and back: two eager phases' worth, a new capture
    assert det._rl_graph.captures == 2 and not g._geoms
    assert all(np.isfinite(v) for o in outs for v in o.values())


def test_senti_iterations_from_graphs_equal_the_eager_phases_and_track_plain_eager():
    """The other half of the reference's RL epochs (train_rl.py:232-235, decoder.py with data_type 'senti'): images with
    sentiment labels and no captions - no XE unroll, no CIDEr-D reward.  Served from the same graphs (sampled roll-out,
    greedy baseline, seq2seq branch, backward): replays == the phases run eagerly, bit for bit; against the plain eager
    Detector.forward within rounding; the dictionary has the reference's five keys."""
    items, scs, split, draws = data(1)
    fns, fc, att, _, cpts, sentis, _ = items[0]
    labels = torch.from_numpy(np.random.default_rng(5).integers(0, len(synth.SENTIMENT_CATEGORIES), size=fc.shape[0]))
    senti_items = [(fns, fc, att, cpts, sentis, labels)] * 6

    def run_senti(det):
        det.set_ciderd_scorer(split)
        det.xe_ss_prob = det.seq2seq_ss_prob = 0.0
        out = []
        for it in senti_items:
            force_draws(det, draws)
            if det.train_graphs and det._rl_graph is None:
                from insenticap_model_amd.train_graph import RLTrainGraph
                det._rl_graph = RLTrainGraph(det, warmup=det._graph_warmup)
            out.append(det(([it], scs), 'senti', True))
        torch.cuda.synchronize()
        return out
    g, e, p = make(graphs=True, warmup=2), make(graphs=True, warmup=10 ** 6), make(graphs=False)
    og, oe, op_ = run_senti(g), run_senti(e), run_senti(p)
    assert g._rl_graph.captures == 1 and g._rl_graph.replays == 4 and e._rl_graph.captures == 0
    same_params(g, e)
    for a, b, c in zip(og, oe, op_):
        assert set(a) == set(b) == set(c) == {'da_loss', 'cls_reward', 'all_rewards', 'cap_loss', 'seq2seq_loss'}
        for k in a:
            assert a[k] == b[k], k
            np.testing.assert_allclose(a[k], c[k], rtol=2e-4, atol=2e-6, err_msg=k)
    for (k, x), (_, y) in zip(g.captioner.named_parameters(), p.captioner.named_parameters()):
        assert float((x - y).abs().max()) <= 6 * 2 * 4e-4 * 1.01, k


def test_graph_served_iterations_between_the_other_things_a_trainer_does():
    """A short form of tools/soak_mix.py: 'fact' and 'senti' iterations from the graphs with an eager XE step, an evaluation
    roll-out and a beam search on the same captioner in between (weights moved behind the graphs' back: an eager iteration
    and, when a scope was rebuilt, a new capture), every result finite, the stream pool and the kept geometries bounded."""
    from insenticap_model_amd import ops
    from insenticap_model_amd.train import xe_train_step
    items, scs, split, _ = data(2)
    det = make(dropout=0.0, graphs=True, warmup=1)
    det.set_ciderd_scorer(split)
    cap = det.captioner
    fns, fc, att, (caps, lengths), cpts, sentis, _ = items[0]
    labels = torch.zeros(fc.shape[0], dtype=torch.int64)
    senti_item = (fns, fc, att, cpts, sentis, labels)
    fact = (None, fc.to(DEV), att.to(DEV), (caps.to(DEV), lengths), cpts.to(DEV))
    (s_caps, s_len), s_cpts, s_sentis, s_labels = scs[0]
    scs_d = ((s_caps.to(DEV), s_len), s_cpts.to(DEV), s_sentis.to(DEV), s_labels.to(DEV))
    streams0 = None
    for r in range(5):
        for it in items:
            out = det(([it], scs), 'fact', True)
            assert all(np.isfinite(v) for v in out.values())
        out = det(([senti_item], scs), 'senti', True)
        assert set(out) == {'da_loss', 'cls_reward', 'all_rewards', 'cap_loss', 'seq2seq_loss'}
        assert all(np.isfinite(v) for v in out.values())
        if r % 2 == 0:
            l = xe_train_step(cap, det.cap_optim, det.cap_xe_crit, det.cap_da_crit, fact, labels.to(DEV), scs_d, 0.0, 0.1)
            assert np.isfinite(float(l['all_loss']))
            cap.cpt_feats = cap.fc_feats = None
        cap.eval()
        with torch.no_grad():
            seq, lp, mk = cap(fc.to(DEV), att.to(DEV), cpts.to(DEV), sentis.to(DEV), labels.to(DEV), TN, 1, mode='rl')
            words, scores = cap.sample(fc[0].to(DEV), att[0].to(DEV), sentis[0].to(DEV), labels[:1].to(DEV), 3, 1, TN)
        assert bool(torch.isfinite(lp).all()) and len(words) == 3
        cap.train()
        if r == 1:
            streams0 = len(ops._OWNED_STREAMS)
    torch.cuda.synchronize()
    ops.check_numerics('mixed use')
    g = det._rl_graph
    assert len(g._geoms) <= 2 and g.replays >= 4 and len(ops._OWNED_STREAMS) == streams0
