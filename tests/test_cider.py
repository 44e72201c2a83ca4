"""CIDEr-D self-critical reward: native C++ library and the Python oracle vs golden vectors produced
by the reference's own scorer (tests/golden/make_golden.py cider). CPU only."""
import time

import numpy as np
import pytest

from insenticap_model_amd import rewards, synth
from oracle.ciderd_oracle import CiderDOracle

CASES = {'small': (40, 60, 16, 12, 7), 'cfg5': (600, 10000, 256, 20, 8)}


def setup(name):
    n_img, V, B, Tn, seed = CASES[name]
    return synth.make_cider_data(n_img, V, B, seq_len=Tn, seed=seed)


@pytest.mark.parametrize('name', ['small', 'cfg5'])
def test_native_reward_matches_reference(golden, name):
    g = golden('cider')
    split, fns, gt, sample, greedy = setup(name)
    scorer = rewards.get_ciderd_scorer(split, 1, 2)
    assert scorer.num_images == CASES[name][0]
    rew = rewards.get_self_critical_reward(sample, greedy, fns, gt, 1, 2, scorer)
    assert rew.dtype == np.float64 and rew.shape == sample.shape
    assert (rew == rew[:, :1]).all()                      # repeated over T (utils.py:82)
    np.testing.assert_allclose(rew[:, 0], g[name + '/reward'], rtol=0, atol=1e-12)
    raw = scorer.score_arrays(np.concatenate([sample, greedy]), [gt[f] for f in fns] * 2)
    np.testing.assert_allclose(raw, g[name + '/scores'], rtol=0, atol=1e-12)
    assert raw.max() > 1.0                                # the fixture has real n-gram overlap
    # the two halves the graph-served RL iteration scores at different moments (train_graph.RLTrainGraph._rewards)
    halves = rewards.self_critical_scores(sample, fns, gt, scorer) - rewards.self_critical_scores(greedy, fns, gt, scorer)
    assert (halves == rew[:, 0]).all()


def test_oracle_matches_reference(golden):
    g = golden('cider')
    split, fns, gt, sample, greedy = setup('small')
    caps = {}
    for v in split.values():
        caps.update(v)
    orc = CiderDOracle(list(caps.values()), 1, 2)
    rew = orc.self_critical_reward(sample, greedy, [gt[f] for f in fns])
    np.testing.assert_allclose(rew, g['small/reward'], rtol=0, atol=1e-12)


def test_thread_count_does_not_change_scores():
    split, fns, gt, sample, greedy = setup('cfg5')
    scorer = rewards.get_ciderd_scorer(split, 1, 2)
    refs = [gt[f] for f in fns]
    scorer.n_threads = 1
    a = scorer.score_arrays(sample, refs)
    scorer.n_threads = 8
    b = scorer.score_arrays(sample, refs)
    assert (a == b).all()


def test_edge_cases():
    split, fns, gt, sample, greedy = setup('small')
    scorer = rewards.get_ciderd_scorer(split, 1, 2)
    refs = [gt[f] for f in fns]
    # an all-<PAD> row (the roll-out of a row that emitted <EOS> first) and a row with a leading <SOS>
    hyp = sample.copy()
    hyp[0, :] = 0
    hyp[0, 0] = 2
    hyp[1, 1:] = hyp[1, :-1].copy()
    hyp[1, 0] = 1
    s = scorer.score_arrays(hyp, refs)
    assert np.isfinite(s).all() and s[0] >= 0
    orc_caps = {}
    for v in split.values():
        orc_caps.update(v)
    orc = CiderDOracle(list(orc_caps.values()), 1, 2)
    np.testing.assert_allclose(s[:2], [orc.score(hyp[0], refs[0]), orc.score(hyp[1], refs[1])], atol=1e-12)
    # identical hypothesis and greedy rows => zero reward
    rew = rewards.get_self_critical_reward(sample, sample, fns, gt, 1, 2, scorer)
    assert (rew == 0).all()
    with pytest.raises(RuntimeError):
        scorer.score_arrays(sample[:1], [[]])             # a hypothesis without references


def test_native_scorer_is_fast():
    """The reference's pure-Python scorer does ~1,240 hypotheses/s on one core."""
    split, fns, gt, sample, greedy = setup('cfg5')
    scorer = rewards.get_ciderd_scorer(split, 1, 2)
    refs = [gt[f] for f in fns]
    scorer.n_threads = 1
    t0 = time.perf_counter()
    for _ in range(4):
        scorer.score_arrays(sample, refs)
    rate = 4 * len(sample) / (time.perf_counter() - t0)
    assert rate > 5000, rate
