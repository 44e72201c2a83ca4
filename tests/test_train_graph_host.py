"""Host-side contract of train_graph.XETrainGraph that needs no GPU: it is the product path (no CPU fallback), takes
the fused optimizer only, and the three scalars it hands the captured optimizer launch are the eager launch's."""
import math

import pytest
import torch

from insenticap_model_amd import Captioner, ops, synth
from insenticap_model_amd._lib import HipLibraryError
from insenticap_model_amd.train_graph import XETrainGraph


def _tiny():
    return Captioner(synth.make_idx2word(64), synth.SENTIMENT_CATEGORIES, synth.TINY_SETTINGS)


def test_refuses_a_foreign_optimizer_and_cpu_parameters():
    cap = _tiny()
    optim, xc, dc = cap.get_optim_criterion(4e-4)
    with pytest.raises(TypeError):
        XETrainGraph(cap, torch.optim.Adam(cap.parameters(), lr=4e-4), xc, dc)
    with pytest.raises(HipLibraryError):          # parameters on the CPU: there is no CPU path
        XETrainGraph(cap, optim, xc, dc)


def test_adam_hyper_is_what_the_eager_launch_derives():
    """isc_clamp_adam computes 1 - beta1^t and sqrt(1 - beta2^t) in double and rounds to float (backward.hip:
    clamp_adam_launch); ops.adam_hyper must give the same three floats for the device-resident form."""
    for step in (1, 2, 10, 1000, 123456):
        lr, b1, b2 = 4e-4, 0.9, 0.999
        got = torch.tensor(ops.adam_hyper(lr, b1, b2, step), dtype=torch.float32)
        want = torch.tensor([lr, 1.0 - math.pow(b1, float(step)), math.sqrt(1.0 - math.pow(b2, float(step)))],
                            dtype=torch.float32)
        assert torch.equal(got, want), step


def test_merged_unroll_policy(monkeypatch):
    """autograd_pair.use_pair: merged where the host issues the launches (eager steps), two branches inside graphs;
    `captioner.pair_unrolls` forces either; ISC_PAIR_UNROLLS overrides both (A/B runs)."""
    from insenticap_model_amd.autograd_pair import use_pair

    class Cap:
        pair_unrolls = None
    monkeypatch.delenv('ISC_PAIR_UNROLLS', raising=False)
    c = Cap()
    assert use_pair(c, in_graph=False) is True and use_pair(c, in_graph=True) is False
    c.pair_unrolls = True
    assert use_pair(c, True) is True and use_pair(c, False) is True
    c.pair_unrolls = False
    assert use_pair(c, True) is False and use_pair(c, False) is False
    monkeypatch.setenv('ISC_PAIR_UNROLLS', '1')
    assert use_pair(c, True) is True
    monkeypatch.setenv('ISC_PAIR_UNROLLS', '0')
    c.pair_unrolls = True
    assert use_pair(c, False) is False


def test_gradient_buckets_cover_the_arena_in_order():
    """dp.GradSink: four contiguous buckets over the arena in parameter order (embeddings + fc | att_embed .. senti2att |
    attention .. lang-LSTM | classifier), every view 256-byte aligned, nothing left out.  CPU tensors: layout only."""
    import torch
    from insenticap_model_amd import Captioner, dp, synth
    cap = Captioner(synth.make_idx2word(64), synth.SENTIMENT_CATEGORIES, synth.TINY_SETTINGS)
    arena = dp.GradArena(cap.parameters())
    sink = dp.GradSink(cap, arena, exchange=False)
    names = [n for n, _ in cap.named_parameters()]
    assert [bk['names'][0] for bk in sink.buckets] == ['word_embed.0.weight', 'att_embed.0.weight',
                                                        'attention.cont_att.h2att.weight', 'classifier.weight']
    assert sum((bk['names'] for bk in sink.buckets), []) == names
    assert sum(bk['flat'].numel() for bk in sink.buckets) == arena.flat.numel()
    for q, off in zip(arena.params, arena.offsets):
        assert off % dp.GradArena.ALIGN == 0 and q.grad.data_ptr() == arena.flat.data_ptr() + 4 * off
    for a, b in zip(sink.buckets, sink.buckets[1:]):
        assert a['flat'].data_ptr() + 4 * a['flat'].numel() == b['flat'].data_ptr()
    assert sink.bucket_of['lang_lstm.weight_hh'] == 2 and sink.bucket_of['senti2att.0.bias'] == 1
