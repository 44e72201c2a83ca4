"""Run-to-run determinism and internal-path equivalences of the HIP path:
* greedy roll-out, XE forward+backward and the fused clamp+Adam step are bit-identical across repeats
  (no atomics on floating-point data; split-K slabs are reduced in a fixed order),
* the split-K route (small M) gives the same numbers as the single-pass route up to fp32 summation order,
  and is itself bit-repeatable,
* the one-call step plan (isc_step_fwd) and the per-kernel Python path (the one bench.py times) agree bit for bit.
"""
import numpy as np
import pytest
import torch

from conftest import case_setup
from insenticap_model_amd import Captioner, ops, synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def _cap(V=1000, settings=synth.DEFAULT_SETTINGS, seed=0):
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, settings)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, settings, seed=seed).items()})
    return cap.to(dev())


def _inputs(B, V, settings=synth.DEFAULT_SETTINGS, seed=3, T=12):
    d = synth.make_inputs(B, V, settings, regions=36, seq_len=T, seed=seed)
    t = lambda k: torch.from_numpy(np.asarray(d[k])).to(dev())
    return d, t


@pytest.mark.parametrize('B', [6, 300])
def test_greedy_rollout_is_bit_repeatable(B):
    cap = _cap().eval()
    d, t = _inputs(B, 1000)
    outs = []
    with torch.no_grad():
        for _ in range(3):
            seq, lp, mk = cap(t('fc_feats'), t('att_feats'), t('cpt_words'), t('senti_words'), t('senti_labels'), 12,
                              sample_max=1, mode='rl')
            outs.append((seq.clone(), lp.clone(), mk.clone()))
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))


def test_training_step_is_bit_repeatable():
    d, t = _inputs(48, 1000, T=10)
    results = []
    for rep in range(2):
        cap = _cap().train()
        optim, xe_crit, da_crit = cap.get_optim_criterion(4e-4)
        torch.manual_seed(7)                                   # same dropout masks
        pred = cap(t('fc_feats'), t('att_feats'), t('cpt_words'), t('captions'), t('senti_labels'), 0.0, mode='xe')
        loss = xe_crit(pred, t('captions')[:, 1:], d['lengths']) + da_crit(cap.cpt_feats, cap.fc_feats.detach())
        optim.zero_grad()
        loss.backward()
        grads = {k: p.grad.clone() for k, p in cap.named_parameters() if p.grad is not None}
        from insenticap_model_amd import clip_gradient
        clip_gradient(optim, 0.1)
        optim.step()
        results.append((float(loss.detach()), grads, {k: v.clone() for k, v in cap.state_dict().items()}))
    assert results[0][0] == results[1][0]
    for k in results[0][1]:
        assert torch.equal(results[0][1][k], results[1][1][k]), k
    for k in results[0][2]:
        assert torch.equal(results[0][2][k], results[1][2][k]), k


@pytest.mark.parametrize('M,N,K', [(5, 2048, 1536), (128, 512, 2048), (64, 10000, 512)])
def test_splitk_route_matches_single_pass(M, N, K):
    g = torch.Generator().manual_seed(M + K)
    a = (torch.rand(M, K, generator=g) * 2 - 1).to(dev())
    w = ((torch.rand(N, K, generator=g) * 2 - 1) * K ** -0.5).to(dev())
    b = (torch.rand(N, generator=g) - 0.5).to(dev())
    ref = (a.double() @ w.double().t() + b.double()).float()
    outs = []
    for rep in range(2):                                       # small M -> plan_splitk engages (workspace attached)
        o = torch.empty(M, N, device=dev())
        ops.linear_fwd([ops.linear_problem([(a, w)], o, b)])
        outs.append(o)
    assert torch.equal(outs[0], outs[1])                       # fixed-order slab reduction
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref.cpu().numpy(), atol=2e-5, rtol=1e-5)
    # single pass: the same rows inside a problem tall enough to have plenty of tiles on its own
    big = torch.cat([a, torch.zeros(4096 - M, K, device=dev())], 0)
    ob = torch.empty(4096, N, device=dev())
    ops.linear_fwd([ops.linear_problem([(big, w)], ob, b)])
    np.testing.assert_allclose(ob[:M].cpu().numpy(), outs[0].cpu().numpy(), atol=1e-5, rtol=1e-5)


def test_step_plan_equals_per_kernel_path():
    """Captioner._step (one isc_step_fwd call) vs Captioner._step_py (the per-kernel route bench.py's event
    timer uses): same kernels, same order -> same bits."""
    cap = _cap().eval()
    d, t = _inputs(40, 1000)
    args = (t('fc_feats'), t('att_feats'), t('cpt_words'), t('senti_words'), t('senti_labels'), 12)
    with torch.no_grad():
        a = cap(*args, sample_max=1, mode='rl')
        saved = ops.TIMER.arm_step
        try:
            outs = []
            for step in range(0, 12, 5):                       # arm different steps: those run through _step_py
                ops.TIMER.arm_step = step
                outs.append(cap(*args, sample_max=1, mode='rl'))
        finally:
            ops.TIMER.arm_step = saved
            ops.TIMER.armed = False
            ops.TIMER.records.clear()
    for o in outs:
        assert all(torch.equal(x, y) for x, y in zip(a, o))


def test_side_stream_unroll_gives_identical_step():
    """train.xe_train_step with the seq2seq unroll on a side stream == everything on one stream, bit for bit
    (losses, parameters after clamp + Adam)."""
    from insenticap_model_amd.train import xe_train_step
    V = 1000
    d = synth.make_inputs(24, V, synth.DEFAULT_SETTINGS, regions=36, seq_len=10, seed=21)
    s = synth.make_inputs(16, V, synth.DEFAULT_SETTINGS, regions=36, seq_len=10, seed=22)
    tt = lambda x: torch.from_numpy(np.asarray(x)).to(dev())
    fact = (None, tt(d['fc_feats']), tt(d['att_feats']), (tt(d['captions']), d['lengths']), tt(d['cpt_words']))
    scs = ((tt(s['captions']), s['lengths']), tt(s['cpt_words']), tt(s['senti_words']), tt(s['senti_labels']))
    res = []
    for overlap in (False, True, True):
        cap = _cap(V).train()
        optim, xe_crit, da_crit = cap.get_optim_criterion(4e-4)
        torch.manual_seed(5)
        outs = []
        for it in range(3):
            o = xe_train_step(cap, optim, xe_crit, da_crit, fact, tt(d['senti_labels']), scs, 0.0, 0.1,
                              overlap_unrolls=overlap)
            outs.append({k: float(v) for k, v in o.items()})
        torch.cuda.synchronize()
        res.append((outs, {k: v.clone() for k, v in cap.state_dict().items()}))
    for other in res[1:]:
        assert other[0] == res[0][0]
        for k in res[0][1]:
            assert torch.equal(res[0][1][k], other[1][k]), k


@pytest.mark.parametrize('B', [5, 130])
def test_graph_replay_equals_eager_rollout(B):
    """enable_rollout_graphs(): the captured T-step roll-out returns the eager path's bits for new inputs of the
    same geometry, and follows in-place weight changes."""
    cap = _cap().eval()
    ref = _cap().eval()
    cap.enable_rollout_graphs(True)        # (the default since round 3, for at most ROLLOUT_GRAPH_MAX_ROWS captions)
    ref.enable_rollout_graphs(False)
    outs = []
    with torch.no_grad():
        for seed in (3, 4, 5, 6):                              # call 1 eager, call 2 captures, calls 3-4 replay
            d, t = _inputs(B, 1000, seed=seed)
            args = (t('fc_feats'), t('att_feats'), t('cpt_words'), t('senti_words'), t('senti_labels'), 12)
            got = cap(*args, sample_max=1, mode='rl')
            exp = ref(*args, sample_max=1, mode='rl')
            assert all(torch.equal(a, b) for a, b in zip(got, exp)), seed
            assert torch.equal(cap.cont_weights, ref.cont_weights)
        assert any(isinstance(v, tuple) for v in cap._rollout_graphs.values())
        # in-place weight update (classifier bias): graphs belong to the weight values they were captured under -
        # this call runs eagerly (and rebuilds the planes), the next one would capture anew
        for m in (cap, ref):
            m.classifier.bias.add_(torch.linspace(-1, 1, m.classifier.bias.numel(), device=dev()))
        got = cap(*args, sample_max=1, mode='rl')
        exp = ref(*args, sample_max=1, mode='rl')
        assert all(torch.equal(a, b) for a, b in zip(got, exp))


def test_default_graph_serving_switches_itself_off_when_geometries_keep_changing():
    """Roll-out graphs are on by default for small batches.  A caller whose batch size wanders evicts captured graphs
    from the 4-geometry cache again and again - each capture is a device-wide synchronisation: after
    Captioner.GRAPH_THRASH_LIMIT evictions the default caches switch themselves off (a warning says so) and the calls
    run eagerly, with the same results throughout."""
    import warnings
    cap, ref = _cap().eval(), _cap().eval()
    ref.enable_rollout_graphs(False)
    assert cap._rollout_graphs == {} and not cap.__dict__.get('_graphs_explicit', False)
    with torch.no_grad(), warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        for B in list(range(1, 15)) * 2:
            d, t = _inputs(B, 1000, seed=B, T=6)
            args = (t('fc_feats'), t('att_feats'), t('cpt_words'), t('senti_words'), t('senti_labels'), 6)
            for rep in range(2):                   # second call of a geometry captures, so every new B evicts one
                got, exp = cap(*args, sample_max=1, mode='rl'), ref(*args, sample_max=1, mode='rl')
                assert all(torch.equal(a, b) for a, b in zip(got, exp)), (B, rep)
    assert cap._rollout_graphs is None and cap._beam_graphs is None
    assert cap.__dict__['_graph_evictions'] == Captioner.GRAPH_THRASH_LIMIT
    assert any('default graph serving is off' in str(w.message) for w in caught)
    cap.enable_rollout_graphs(True, max_graphs=16)                # the caller's answer: room for its geometries
    assert cap._rollout_graphs == {} and cap._graphs_explicit
