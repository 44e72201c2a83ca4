"""The merged step chain of an iteration's two sibling unrolls (insenticap_model_amd/autograd_pair.py,
`Captioner.forward_xe_seq2seq`: forward_xe + forward_seq2seq of train_xe.py:160-181 through ONE chain of per-step
launches) against the two separate calls (`captioner.pair_unrolls = False`, the round-4 form that the reference-written
goldens pin in tests/test_gpu_parity.py / test_gpu_bench_config.py): log-probs, losses, all 40 gradients and the
post-step parameters, at the tiny geometry and at BASELINE configs[1]'s (128 + 80 rows, V = 10k, T = 20, 36 x 2048),
with equal and with different caption lengths, with replayed dropout masks, and the launch count of an iteration.
pytest -m gpu."""
import numpy as np
import pytest
import torch

from insenticap_model_amd import Captioner, ops, synth
from insenticap_model_amd.train import xe_train_step

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TINY = dict(V=64, st=synth.TINY_SETTINGS, R=6, T=8, B=8, S=4)
FULL = dict(V=10000, st=synth.DEFAULT_SETTINGS, R=36, T=20, B=128, S=80)
GRAD_TOL = 1e-4          # gradients within 1e-4 of the tensor's largest element


def make(cfg, seed=9, pair=True):
    cap = Captioner(synth.make_idx2word(cfg['V']), synth.SENTIMENT_CATEGORIES, cfg['st'])
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(cfg['V'], cfg['st'], seed=seed).items()})
    cap.pair_unrolls = pair
    return cap.to(DEV).eval()


def batch(cfg, seed, T=None, Ts=None):
    T, Ts = T or cfg['T'], Ts or T or cfg['T']
    d = synth.make_inputs(cfg['B'], cfg['V'], cfg['st'], regions=cfg['R'], seq_len=T, seed=seed)
    s = synth.make_inputs(cfg['S'], cfg['V'], cfg['st'], regions=cfg['R'], seq_len=Ts, seed=seed + 1000)
    t = lambda x: torch.from_numpy(x).to(DEV)      # noqa: E731
    fact = (None, t(d['fc_feats']), t(d['att_feats']), (t(d['captions']), d['lengths']), t(d['cpt_words']))
    scs = ((t(s['captions']), s['lengths']), t(s['cpt_words']), t(s['senti_words']), t(s['senti_labels']))
    return fact, t(d['senti_labels']), scs


def forward_backward(cap, fact, labels, scs, masks=(None, None)):
    """One iteration's forward + backward through the public call surface; returns log-probs, losses, gradients."""
    _, fc, att, (caps, lengths), cpts = fact
    (s_caps, s_len), s_cpts, s_sentis, s_labels = scs
    _, xc, dc = cap.get_optim_criterion(4e-4)
    for q in cap.parameters():
        q.grad = None
    pred, pred2 = cap.forward_xe_seq2seq(fc, att, cpts, caps, labels, 0.0, s_caps, s_cpts, s_sentis, s_labels, 0.0,
                                         _masks=masks[0], _s_masks=masks[1])
    xe = xc(pred, caps[:, 1:], lengths)
    da = dc(cap.cpt_feats, cap.fc_feats.detach())
    s2s = xc(pred2, s_caps[:, 1:], s_len)
    (xe + da + s2s).backward()
    # (the gate's parameters - attention.h2att / cont2att / senti2att / att_alpha - are in neither unroll: no gradient)
    grads = {k: q.grad.detach().clone() for k, q in cap.named_parameters() if q.grad is not None}
    return pred.detach(), pred2.detach(), (float(xe.detach()), float(da.detach()), float(s2s.detach())), grads


def assert_close_grads(ga, gb, tol=GRAD_TOL):
    assert ga.keys() == gb.keys() and len(ga) == 32
    for k in ga:
        gmax = float(gb[k].abs().max())
        err = float((ga[k] - gb[k]).abs().max())
        assert err <= tol * max(gmax, 1e-12), (k, err, gmax)


@pytest.mark.parametrize('cfg,T,Ts', [(TINY, 8, 8), (TINY, 8, 5), (TINY, 5, 8), (FULL, 20, 20), (FULL, 20, 14)],
                         ids=['tiny', 'tiny_s2s_shorter', 'tiny_xe_shorter', 'b128_v10k', 'b128_s2s_shorter'])
def test_merged_unrolls_equal_the_two_separate_calls(cfg, T, Ts):
    fact, labels, scs = batch(cfg, 70, T, Ts)
    a = forward_backward(make(cfg, pair=True), fact, labels, scs)
    before = ops._lib.load().isc_h3s_launches()
    b = forward_backward(make(cfg, pair=False), fact, labels, scs)
    assert ops._lib.load().isc_h3s_launches() >= before
    for x, y in ((a[0], b[0]), (a[1], b[1])):
        assert x.shape == y.shape
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(a[2], b[2], rtol=2e-6, atol=2e-6)
    assert_close_grads(a[3], b[3])
    assert ops.device_status() == 0


def test_merged_unrolls_with_replayed_dropout_masks():
    """Train-mode dropout with the SAME keep-masks on both paths (prologue masks and the per-step masks on h_lang)."""
    cfg = TINY
    fact, labels, scs = batch(cfg, 71)
    st = cfg['st']
    g = torch.Generator().manual_seed(5)
    E, H, Wd = st['feat_emb_dim'], st['rnn_hid_dim'], st['word_emb_dim']
    B, S, R, T = cfg['B'], cfg['S'], cfg['R'], cfg['T']

    def m(*shape):
        return (torch.rand(*shape, generator=g) < 0.5).to(torch.uint8)
    m1 = {'fc': m(B, E), 'att': m(B * R, E), 'label': m(B, Wd)}
    m2 = {'cpt': m(S, E), 'words': m(S * 11, Wd), 'label': m(S, Wd)}
    for t in range(T):
        m1['out%d' % t], m2['out%d' % t] = m(B, H), m(S, H)
    a = forward_backward(make(cfg, pair=True).train(), fact, labels, scs, (m1, m2))
    b = forward_backward(make(cfg, pair=False).train(), fact, labels, scs, (m1, m2))
    np.testing.assert_allclose(a[0].cpu().numpy(), b[0].cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(a[1].cpu().numpy(), b[1].cpu().numpy(), atol=2e-5)
    assert_close_grads(a[3], b[3])


@pytest.mark.parametrize('cfg', [TINY, FULL], ids=['tiny', 'b128_v10k'])
def test_training_steps_merged_vs_separate(cfg):
    """Three whole iterations (forward, backward, clamp + Adam): the losses of iterations 2 and 3 - functions of the
    parameters the earlier steps left - agree; parameters stay within the optimiser's reach of each other (an Adam step
    moves an element by ~lr whatever the gradient's size, so a sign flip of a ~0 gradient shows as 2 lr)."""
    batches = [batch(cfg, 80 + i) for i in range(3)]
    out = []
    for pair in (True, False):
        cap = make(cfg, pair=pair)
        optim, xc, dc = cap.get_optim_criterion(4e-4)
        losses = [xe_train_step(cap, optim, xc, dc, f, l, s, 0.0, 0.1) for f, l, s in batches]
        out.append((cap, [{k: float(v) for k, v in d.items()} for d in losses]))
    (ca, la), (cb, lb) = out
    for x, y in zip(la, lb):
        for k in x:
            assert abs(x[k] - y[k]) <= 5e-4 * max(1.0, abs(y[k])), (k, x[k], y[k])
    for (k, p), (_, q) in zip(ca.named_parameters(), cb.named_parameters()):
        d = (p.detach() - q.detach()).abs()
        assert float(d.max()) <= 3 * 2 * 4e-4 + 1e-7, k


def test_scheduled_sampling_and_dropout_run_through_the_merged_chain():
    """Train mode, ss_prob > 0 (per-step draws + per-step classifier), equal and different probabilities for the two
    branches: finite losses, every gradient finite, fed tokens inside the vocabulary."""
    cfg = TINY
    fact, labels, scs = batch(cfg, 72, 8, 6)
    _, fc, att, (caps, lengths), cpts = fact
    (s_caps, s_len), s_cpts, s_sentis, s_labels = scs
    for p1, p2 in ((0.25, 0.25), (0.5, 0.25), (0.5, 0.0)):
        cap = make(cfg, pair=True).train()
        _, xc, dc = cap.get_optim_criterion(4e-4)
        torch.manual_seed(3)
        pred, pred2 = cap.forward_xe_seq2seq(fc, att, cpts, caps, labels, p1, s_caps, s_cpts, s_sentis, s_labels, p2)
        loss = xc(pred, caps[:, 1:], lengths) + xc(pred2, s_caps[:, 1:], s_len) + dc(cap.cpt_feats, cap.fc_feats.detach())
        loss.backward()
        assert torch.isfinite(loss)
        for k, q in cap.named_parameters():
            assert q.grad is None or bool(torch.isfinite(q.grad).all()), k
        # log-probs are normalised rows
        for x in (pred, pred2):
            np.testing.assert_allclose(x.detach().exp().sum(-1).cpu().numpy(), 1.0, atol=1e-4)
    assert ops.device_status() == 0


def test_launch_count_of_a_merged_iteration():
    """The B = 128 + 80 iteration (BASELINE configs[1] / the unit of configs[3]'s curve) from its HIP graph: at most 380
    kernel nodes (round 4: 609 with one chain per unroll)."""
    from insenticap_model_amd.train_graph import XETrainGraph
    cfg = FULL
    cap = make(cfg)
    optim, xc, dc = cap.get_optim_criterion(4e-4)
    g = XETrainGraph(cap, optim, xc, dc, grad_clip=0.1, warmup=2)
    g.KEEP_GRAPHS = True
    for i in range(4):
        f, l, s = batch(cfg, 90 + i)
        g.step(f, l, s, 0.0)
    torch.cuda.synchronize()
    assert g.replays >= 1
    geo = next(iter(g._geoms.values()))
    nodes = ops.graph_kernel_nodes(geo.g_iter)
    assert 0 < nodes <= 380, nodes


@pytest.mark.parametrize('pair', [True, False], ids=['merged', 'two_calls'])
@pytest.mark.parametrize('cfg,ss', [(TINY, 0.0), (TINY, 0.3), (FULL, 0.0)], ids=['tiny', 'tiny_scheduled_sampling', 'b128_v10k'])
def test_token_logprobs_keep_the_bits_of_the_full_logprob_tensor(cfg, ss, pair):
    """`with captioner.token_logprobs():` - the unrolls return log p(target) [B,T] from the raw logits + tile statistics
    (isc_gather_logp_raw) and the backward recomputes the softmax term from them (isc_logsoftmax_bwd_raw); the [B,T,V]
    log-probs are never formed.  Against the same calls outside the context (full tensor, XECriterion's gather, the stored
    log-probs in the backward): the returned values equal the gathered ones, the losses and ALL gradients are the same
    bits - with scheduled sampling too (same seed: same draws)."""
    fact, labels, scs = batch(cfg, 75)
    _, fc, att, (caps, lengths), cpts = fact
    (s_caps, s_len), s_cpts, s_sentis, s_labels = scs
    res = {}
    for lazy in (True, False):
        cap = make(cfg, pair=pair)
        if ss:
            cap.train()
            cap.drop.p = 0.0             # (scheduled sampling needs train mode; no dropout: the draws are the only randomness)
        _, xc, dc = cap.get_optim_criterion(4e-4)
        torch.manual_seed(11)
        with cap.token_logprobs(lazy):
            pred, pred2 = cap.forward_xe_seq2seq(fc, att, cpts, caps, labels, ss, s_caps, s_cpts, s_sentis, s_labels, ss)
        assert pred.dim() == (2 if lazy else 3)
        xe, s2s = xc(pred, caps[:, 1:], lengths), xc(pred2, s_caps[:, 1:], s_len)
        (xe + s2s + dc(cap.cpt_feats, cap.fc_feats.detach())).backward()
        tl = pred.detach() if lazy else pred.detach().gather(2, caps[:, 1:].unsqueeze(2)).squeeze(2)
        res[lazy] = (tl, float(xe.detach()), float(s2s.detach()),
                     {k: q.grad.detach().clone() for k, q in cap.named_parameters() if q.grad is not None})
    assert torch.equal(res[True][0], res[False][0])
    assert res[True][1:3] == res[False][1:3]
    assert res[True][3].keys() == res[False][3].keys()
    for k in res[True][3]:
        assert torch.equal(res[True][3][k], res[False][3][k]), k
