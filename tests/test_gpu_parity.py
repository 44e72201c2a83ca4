"""Parity of the HIP path (through the C-ABI library) against the CPU oracle and the golden
vectors produced by the reference. Run on the MI355X box: pytest -m gpu."""
import contextlib

import numpy as np
import pytest
import torch

from conftest import case_setup, trusted_prefix
from insenticap_model_amd import Captioner, XECriterion, ops, synth

pytestmark = pytest.mark.gpu

LOGP_TOL = 1e-4   # BASELINE.json north_star: log-probs within 1e-4 fp32


def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda:0')


def make_captioner(name):
    c, st, w, d, s2s = case_setup(name)
    cap = Captioner(synth.make_idx2word(c['V']), synth.SENTIMENT_CATEGORIES, st)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    cap.to(dev()).eval()
    return cap, c, st, w, d, s2s


def T(d, k):
    return torch.from_numpy(np.asarray(d[k])).to(dev())


def oracle():
    from oracle import captioner_oracle as O
    return O


# ----------------------------------------------------------------------------- kernels
@pytest.mark.parametrize('M,N,K1,K2', [(4, 512, 512, 0), (128, 512, 2048, 0), (300, 1536, 512, 512),
                                       (2048, 512, 512, 0), (37, 64, 32, 64), (1000, 10000 // 8 * 4, 512, 0)])
def test_linear_kernel_vs_fp64(M, N, K1, K2):
    g = torch.Generator().manual_seed(M * 7 + N)
    x1 = torch.randn(M, K1, generator=g)
    w1 = torch.randn(N, K1, generator=g) / K1 ** 0.5
    b = torch.randn(N, generator=g)
    ref = x1.double() @ w1.double().t() + b.double()
    segs = [(x1.to(dev()), w1.to(dev()))]
    if K2:
        x2 = torch.randn(M, K2, generator=g)
        w2 = torch.randn(N, K2, generator=g) / K2 ** 0.5
        ref = ref + x2.double() @ w2.double().t()
        segs.append((x2.to(dev()), w2.to(dev())))
    ref = torch.relu(ref)
    out = torch.empty(M, N, device=dev())
    ops.linear_fwd([ops.linear_problem(segs, out, b.to(dev()), relu=True)])
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), atol=2e-5, rtol=1e-5)


def test_linear_grouped_and_views():
    """3 problems in one launch; A given as a strided view, W as a column slice."""
    g = torch.Generator().manual_seed(5)
    big = torch.randn(70, 96, generator=g).to(dev())
    wfull = torch.randn(64, 160, generator=g).to(dev())
    outs, probs, refs = [], [], []
    for i, (a, w) in enumerate([(big[:, 0:32], wfull[:, 32:64]), (big[:, 32:96], wfull[:, 96:160]),
                                (big[:, 64:96], wfull[:32, 0:32])]):
        o = torch.empty(a.shape[0], w.shape[0], device=dev())
        probs.append(ops.linear_problem([(a, w)], o))
        outs.append(o)
        refs.append(a.double().cpu() @ w.double().cpu().t())
    ops.linear_fwd(probs)
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        np.testing.assert_allclose(o.cpu().numpy(), r.float().numpy(), atol=2e-5)


@pytest.mark.parametrize('M,H', [(3, 32), (128, 512), (700, 512)])
def test_lstm_kernel_vs_fp64(M, H):
    g = torch.Generator().manual_seed(M + H)
    E = H
    x = torch.randn(M, E, generator=g)
    h = torch.randn(M, H, generator=g) * 0.5
    c = torch.randn(M, H, generator=g)
    wih = torch.randn(4 * H, E, generator=g) / E ** 0.5
    whh = torch.randn(4 * H, H, generator=g) / H ** 0.5
    bih, bhh = torch.randn(4 * H, generator=g), torch.randn(4 * H, generator=g)
    gates = x.double() @ wih.double().t() + bih.double() + h.double() @ whh.double().t() + bhh.double()
    i, f, gg, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c.double() + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    D = dev()
    ho, co = torch.empty(M, H, device=D), torch.empty(M, H, device=D)
    go = torch.empty(M, 4 * H, device=D)
    ops.lstm_fwd([(x.to(D), wih.to(D)), (h.to(D), whh.to(D))], bih.to(D), bhh.to(D), c.to(D), ho, co, gates_out=go)
    torch.cuda.synchronize()
    np.testing.assert_allclose(ho.cpu().numpy(), h2.float().numpy(), atol=2e-5)
    np.testing.assert_allclose(co.cpu().numpy(), c2.float().numpy(), atol=2e-5)
    act = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)], dim=1)
    np.testing.assert_allclose(go.cpu().numpy(), act.float().numpy(), atol=2e-5)


@pytest.mark.parametrize('M,V,K', [(5, 64, 32), (130, 10000, 512), (64, 9487, 512)])
def test_vocab_kernel_logsoftmax(M, V, K):
    g = torch.Generator().manual_seed(V)
    h = torch.randn(M, K, generator=g)
    w = torch.randn(V, K, generator=g) * 0.3
    b = torch.randn(V, generator=g)
    ref = torch.log_softmax(h.double() @ w.double().t() + b.double(), dim=1)
    D = dev()
    nt = (V + 127) // 128
    pm, ps = torch.empty(M, nt, device=D), torch.empty(M, nt, device=D)
    pi = torch.empty(M, nt, dtype=torch.int32, device=D)
    logits = torch.empty(M, V, device=D)
    ops.vocab_fwd(h.to(D), w.to(D), b.to(D), pm, ps, pi, logits)
    ops.logsoftmax_apply(logits, pm, ps)
    torch.cuda.synchronize()
    np.testing.assert_allclose(logits.cpu().numpy(), ref.float().numpy(), atol=5e-5)
    # arg-max from the tile statistics
    best = pm.cpu().argmax(dim=1)
    am = pi.cpu()[torch.arange(M), best].long()
    assert (am == ref.argmax(dim=1)).all()


@pytest.mark.parametrize('B,R,A', [(3, 6, 32), (64, 36, 512), (5, 196, 512), (9, 11, 512)])
def test_attention_scan_vs_fp64(B, R, A):
    g = torch.Generator().manual_seed(B * R)
    Pm = torch.randn(B, R, A, generator=g)
    Vm = torch.randn(B, R, A, generator=g)
    q = torch.randn(B, A, generator=g)
    q2 = torch.randn(B, A, generator=g)
    w = torch.randn(1, A, generator=g) * 0.3
    wb = torch.randn(1, generator=g)
    e = (torch.tanh(Pm.double() + q.double().unsqueeze(1) + q2.double().unsqueeze(1)) @ w.double().t()).squeeze(-1) + wb.double()
    al = torch.softmax(e, dim=-1)
    ref = torch.bmm(al.unsqueeze(1), Vm.double()).squeeze(1)
    D = dev()
    out, alpha = torch.empty(B, A, device=D), torch.empty(B, R, device=D)
    ops.attn_scan_fwd([ops.scan_problem(Pm.to(D), Vm.to(D), q.to(D), w.to(D), wb.to(D), out, alpha, q2=q2.to(D))], B)
    torch.cuda.synchronize()
    np.testing.assert_allclose(alpha.cpu().numpy(), al.float().numpy(), atol=2e-6)
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), atol=1e-5)


def test_attention_scan_streaming_policy_does_not_change_the_numbers():
    """A launch that streams more than 128 MB of per-caption rows loads them non-temporally (attention.hip); the same
    rows in two half launches take the default policy: the outputs must be bit-identical (and so must the backward's,
    which applies the same rule to its P / V reads)."""
    D = dev()
    B, R, A = 1024, 36, 512                                   # 2 * 36 * 512 * 4 B * 1024 = 151 MB > 128 MB
    g = torch.Generator().manual_seed(5)
    Pm, Vm = torch.randn(B, R, A, generator=g).to(D), torch.randn(B, R, A, generator=g).to(D)
    q, w, wb = torch.randn(B, A, generator=g).to(D), (torch.randn(1, A, generator=g) * 0.3).to(D), torch.randn(1, generator=g).to(D)

    def fwd(sl):
        n = sl.stop - sl.start
        out, alpha = torch.empty(n, A, device=D), torch.empty(n, R, device=D)
        ops.attn_scan_fwd([ops.scan_problem(Pm[sl].contiguous(), Vm[sl].contiguous(), q[sl].contiguous(), w, wb, out, alpha)], n)
        return out, alpha
    whole = fwd(slice(0, B))
    halves = [fwd(slice(0, B // 2)), fwd(slice(B // 2, B))]
    torch.cuda.synchronize()
    assert torch.equal(whole[0], torch.cat([h[0] for h in halves])) and torch.equal(whole[1], torch.cat([h[1] for h in halves]))
    dout = torch.randn(B, A, generator=g).to(D)

    def bwd(sl):
        n = sl.stop - sl.start
        dP, dV = torch.empty(n, R, A, device=D), torch.empty(n, R, A, device=D)
        dq, dw = torch.empty(n, A, device=D), torch.empty(n, A, device=D)
        ops.attn_scan_bwd([ops.scan_bwd_problem(Pm[sl].contiguous(), Vm[sl].contiguous(), q[sl].contiguous(), w,
                                                whole[1][sl].contiguous(), dout[sl].contiguous(), dP, dV, dq, dw, False)], n)
        return dP, dV, dq, dw
    bw = bwd(slice(0, B))
    bh = [bwd(slice(0, B // 2)), bwd(slice(B // 2, B))]
    torch.cuda.synchronize()
    for k in range(4):
        assert torch.equal(bw[k], torch.cat([h[k] for h in bh])), k


# ----------------------------------------------------------------------------- end to end
def test_tiny_xe_and_seq2seq_forward_vs_golden(golden):
    g = golden('tiny')
    cap, c, st, w, d, s2s = make_captioner('tiny')
    with torch.no_grad():
        logp = cap(T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'cpt_words'), T(d, 'captions'),
                   T(d, 'senti_labels'), 0.0, mode='xe')
    np.testing.assert_allclose(logp.cpu().numpy(), g['it/xe_logp'], atol=LOGP_TOL)
    np.testing.assert_allclose(cap.fc_feats.cpu().numpy(), g['it/xe_fc_feats'], atol=1e-5)
    np.testing.assert_allclose(cap.cpt_feats.cpu().numpy(), g['it/xe_cpt_feats'], atol=1e-5)
    np.testing.assert_allclose(cap.cont_weights.cpu().numpy(), g['it/xe_cont_weights'], atol=1e-5)
    assert cap.senti_weights == [] and cap.cont_senti_weights == []
    loss = XECriterion()(logp, T(d, 'captions')[:, 1:], d['lengths'])
    np.testing.assert_allclose(float(loss), g['it/losses'][0], rtol=2e-5)
    with torch.no_grad():
        logp2 = cap(T(s2s, 'captions'), T(s2s, 'cpt_words'), T(s2s, 'senti_words'), T(s2s, 'senti_labels'),
                    0.0, mode='seq2seq')
    np.testing.assert_allclose(logp2.cpu().numpy(), g['it/s2s_logp'], atol=LOGP_TOL)
    np.testing.assert_allclose(cap.senti_weights.cpu().numpy(), g['it/s2s_senti_weights'], atol=1e-5)
    loss2 = XECriterion()(logp2, T(s2s, 'captions')[:, 1:], s2s['lengths'])
    np.testing.assert_allclose(float(loss2), g['it/losses'][2], rtol=2e-5)
    # 4-D grid input == flat regions
    dg = synth.make_inputs(c['B'], c['V'], st, regions=c['R'], seq_len=c['T'], seed=c['in_seed'], grid=(2, 3))
    with torch.no_grad():
        lg = cap(T(dg, 'fc_feats'), T(dg, 'att_feats'), T(dg, 'cpt_words'), T(dg, 'captions'),
                 T(dg, 'senti_labels'), 0.0, mode='xe')
    np.testing.assert_allclose(lg.cpu().numpy(), g['grid/xe_logp'], atol=LOGP_TOL)


@pytest.mark.parametrize('prefix', ['rl/', 'early/'])
def test_tiny_rollouts_vs_golden(golden, prefix):
    g = golden('tiny')
    cap, c, st, w, d, _ = make_captioner('tiny')
    if prefix == 'early/':
        d = {k: v[3:6] for k, v in d.items()}
    a = (T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'cpt_words'), T(d, 'senti_words'), T(d, 'senti_labels'))
    with torch.no_grad():
        seq, lp, mk = cap(*a, c['T'], 1, mode='rl')
    assert seq.dtype == torch.int64
    assert (seq.cpu().numpy() == g[prefix + 'greedy_seq']).all()          # token-exact
    assert (mk.cpu().numpy() == g[prefix + 'greedy_masks']).all()
    np.testing.assert_allclose(lp.cpu().numpy(), g[prefix + 'greedy_logprobs'], atol=LOGP_TOL)
    for attr, key in (('cont_weights', 'cont'), ('senti_weights', 'senti'), ('cont_senti_weights', 'gate')):
        ref = g[prefix + 'greedy_%s_weights' % key]
        got = getattr(cap, attr).cpu().numpy()
        assert got.shape == ref.shape, (attr, got.shape, ref.shape)       # early break => fewer steps
        np.testing.assert_allclose(got, ref, atol=1e-5)
    np.testing.assert_allclose(cap.fc_feats.cpu().numpy(), g[prefix + 'greedy_fc_feats'], atol=1e-5)
    np.testing.assert_allclose(cap.cpt_feats.cpu().numpy(), g[prefix + 'greedy_cpt_feats'], atol=1e-5)
    # sampled roll-out, replaying the reference's raw multinomial draws
    with torch.no_grad():
        seq, lp, mk = cap.forward_rl(*a, c['T'], 0, _replay=torch.from_numpy(g[prefix + 'sample_draws']).to(dev()))
    assert (seq.cpu().numpy() == g[prefix + 'sample_seq']).all()
    assert (mk.cpu().numpy() == g[prefix + 'sample_masks']).all()
    np.testing.assert_allclose(lp.cpu().numpy(), g[prefix + 'sample_logprobs'], atol=LOGP_TOL)


def test_tiny_beam_vs_golden(golden):
    g = golden('tiny')
    cap, c, st, w, d, _ = make_captioner('tiny')
    fc, att, sw, lab = T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'senti_words'), T(d, 'senti_labels')
    for b in (3, 5):
        for s in (0, 1):
            caps, scores, _ = cap.sample_batch(fc, att, sw if s else None, lab if s else None, b, 1, c['T'])
            for i in range(c['B']):
                assert caps[i] == list(g['beam/beam%d_senti%d_caps' % (b, s)][i]), (b, s, i)
                np.testing.assert_allclose(scores[i], g['beam/beam%d_senti%d_scores' % (b, s)][i], atol=2e-4)
    caps, scores, _ = cap.sample_batch(fc, att, sw, lab, 3, 0, c['T'])
    for i in range(c['B']):
        assert caps[i] == list(g['beam/beam3_nocons_caps'][i])
    # the reference's one-image API
    cp, sc = cap.sample(fc[1], att[1], sw[1], lab[1:2], 3, 1, c['T'])
    assert cp == list(g['beam/beam3_senti1_caps'][1])


def test_tiny_dropout_and_scheduled_sampling_replay(golden):
    g = golden('tiny')
    cap, c, st, w, d, s2s = make_captioner('tiny')
    masks = {k: torch.from_numpy(g['drop/mask_' + k]) for k in ('fc', 'att', 'label')}
    for i in range(c['T']):
        masks['out%d' % i] = torch.from_numpy(g['drop/mask_out%d' % i])
    with torch.no_grad():
        logp = cap.forward_xe(T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'cpt_words'), T(d, 'captions'),
                              T(d, 'senti_labels'), 0.0, _masks=masks)
    np.testing.assert_allclose(logp.cpu().numpy(), g['drop/xe_logp'], atol=2e-4)
    masks = {k: torch.from_numpy(g['drop_s2s/mask_' + k]) for k in ('cpt', 'words', 'label')}
    for i in range(c['T']):
        masks['out%d' % i] = torch.from_numpy(g['drop_s2s/mask_out%d' % i])
    with torch.no_grad():
        logp = cap.forward_seq2seq(T(s2s, 'captions'), T(s2s, 'cpt_words'), T(s2s, 'senti_words'),
                                   T(s2s, 'senti_labels'), 0.0, _masks=masks)
    np.testing.assert_allclose(logp.cpu().numpy(), g['drop_s2s/logp'], atol=2e-4)
    # scheduled sampling: feed the tokens the reference fed
    fed = torch.from_numpy(g['ss/fed_tokens']).to(dev())
    caps = torch.cat([fed, fed[:, -1:]], dim=1)      # forward_xe feeds captions[:, :-1]
    with torch.no_grad():
        logp = cap.forward_xe(T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'cpt_words'), caps,
                              T(d, 'senti_labels'), 0.0)
    np.testing.assert_allclose(logp.cpu().numpy(), g['ss/xe_logp'], atol=LOGP_TOL)


@pytest.mark.parametrize('name', ['cfg1', 'b128'])
def test_fullsize_greedy_token_exact(golden, name):
    """BASELINE.json configs[0]/[1] shapes (36x2048 feats, V=10k, T=20): token-exact where the
    reference's own top-1/top-2 margin is above fp32 reassociation noise."""
    g = golden(name)
    cap, c, st, w, d, _ = make_captioner(name)
    a = (T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'cpt_words'), T(d, 'senti_words'), T(d, 'senti_labels'))
    with torch.no_grad():
        seq, lp, mk = cap(*a, c['T'], 1, mode='rl')
    seq, lp, mk = seq.cpu().numpy(), lp.cpu().numpy(), mk.cpu().numpy()
    gm = g['rl/greedy_margins']
    gm = np.pad(gm, ((0, 0), (0, c['T'] - gm.shape[1])))
    n = trusted_prefix(gm, g['rl/greedy_masks'], 2e-3)
    assert n.mean() >= 0.8 * c['T']
    for b in range(c['B']):
        assert (seq[b, :n[b]] == g['rl/greedy_seq'][b, :n[b]]).all(), b
        assert (mk[b, :n[b]] == g['rl/greedy_masks'][b, :n[b]]).all(), b
        np.testing.assert_allclose(lp[b, :n[b]], g['rl/greedy_logprobs'][b, :n[b]], atol=LOGP_TOL)


def test_cfg1_beam5_and_xe_vs_golden(golden):
    g = golden('cfg1')
    cap, c, st, w, d, s2s = make_captioner('cfg1')
    fc, att, sw, lab = T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'senti_words'), T(d, 'senti_labels')
    caps, scores, _ = cap.sample_batch(fc[:2], att[:2], sw[:2], lab[:2], 5, 1, c['T'])
    for i in range(2):
        assert caps[i] == list(g['beam/beam5_senti1_caps'][i])
        np.testing.assert_allclose(scores[i], g['beam/beam5_senti1_scores'][i], atol=1e-3)
    with torch.no_grad():
        logp = cap(fc, att, T(d, 'cpt_words'), T(d, 'captions'), lab, 0.0, mode='xe')
    np.testing.assert_allclose(logp.cpu().numpy()[:, :, :32], g['it/xe_logp'], atol=LOGP_TOL)
    tgt = logp.gather(2, T(d, 'captions')[:, 1:].unsqueeze(2)).squeeze(2).cpu().numpy()
    np.testing.assert_allclose(tgt, g['it/xe_logp_tgt'], atol=LOGP_TOL)
    loss = XECriterion()(logp, T(d, 'captions')[:, 1:], d['lengths'])
    np.testing.assert_allclose(float(loss), g['it/losses'][0], rtol=2e-5)


def test_large_batch_properties():
    """BASELINE-size batch (B=1024): size-independent properties - rows are independent
    (row-in-batch == row-alone), masks are monotone, tokens after <EOS> are <PAD>."""
    cap, c, st, w, _, _ = make_captioner('cfg1')
    B, Tn = 1024, 20
    d = synth.make_inputs(B, c['V'], st, regions=36, seq_len=Tn, seed=77)
    a = [T(d, k) for k in ('fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')]
    with torch.no_grad():
        seq, lp, mk = cap(*a, Tn, 1, mode='rl')
        sub = [x[100:164] for x in a]
        seq2, lp2, mk2 = cap(*sub, Tn, 1, mode='rl')
    assert (seq[100:164] == seq2).all()
    np.testing.assert_allclose(lp[100:164].cpu().numpy(), lp2.cpu().numpy(), atol=1e-5)
    mkc, sq = mk.cpu().numpy(), seq.cpu().numpy()
    assert (np.diff(mkc, axis=1) <= 0).all()
    assert (sq[mkc == 0] == 0).all()
    eos_pos = (sq == cap.eos_id) & (mkc == 1)
    assert (eos_pos.sum(1) <= 1).all()
    assert np.isfinite(lp.cpu().numpy()).all()


def test_beam5_batch64_matches_per_image_and_oracle(golden):
    """BASELINE config 2: beam 5 over 64 images with sentiment words.  The batched search must return, image by
    image, the captions the one-image API returns (images are independent; scores agree to fp32 summation order -
    320 rows and 5 rows take different split-K routes), and agree with the CPU oracle's beam search on a sample of
    them (captions where the oracle's own beams are not near-ties, scores to 1e-3)."""
    cap, c, st, w, _, _ = make_captioner('cfg1')
    n, Tn = 64, 20
    d = synth.make_inputs(n, c['V'], st, regions=36, seq_len=Tn, seed=321)
    fc, att, sw, lab = T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'senti_words'), T(d, 'senti_labels')
    caps, scores, ids = cap.sample_batch(fc, att, sw, lab, 5, 1, Tn)
    assert len(caps) == n and all(len(x) == 5 for x in caps)
    for i in (0, 7, 31, 63):
        cp, sc = cap.sample(fc[i], att[i], sw[i], lab[i:i + 1], 5, 1, Tn)
        assert cp == caps[i], i
        np.testing.assert_allclose(sc, scores[i], atol=1e-4)
    O = oracle()
    p = O.to_params(w)
    oid = O.Ids(synth.make_idx2word(c['V']), synth.SENTIMENT_CATEGORIES)
    for i in (3, 40):
        ocaps, oscores, _ = O.beam_search(p, oid, synth.make_idx2word(c['V']), torch.from_numpy(d['fc_feats'][i]),
                                          torch.from_numpy(d['att_feats'][i]), torch.from_numpy(d['senti_words'][i]),
                                          torch.from_numpy(d['senti_labels'][i:i + 1]), 5, 1, Tn)
        np.testing.assert_allclose(scores[i], oscores, atol=1e-3)
        gaps = np.abs(np.diff(np.asarray(oscores)))
        if (gaps > 2e-3).all():                    # well separated beams: order and words must match exactly
            assert caps[i] == list(ocaps), i
        else:
            assert caps[i][0] == ocaps[0] or abs(oscores[0] - oscores[1]) <= 2e-3


@pytest.mark.parametrize('mode', [1, 2])
def test_greedy_b1024_vs_oracle_on_the_split_f16_path(mode):
    """B=1024 greedy roll-out at full size against the CPU oracle with the split-f16 GEMM path in play (mode 1: the
    classifier takes it; mode 2: every launch large enough not to be split over K, i.e. both LSTM cells, the
    projections and the prologue as well).  Token-exact wherever the oracle's own top-1/top-2 margin is above fp32
    reassociation noise, log-probs within the 1e-4 bound."""
    cap, c, st, w, _, _ = make_captioner('cfg1')
    B, Tn = 1024, 20
    d = synth.make_inputs(B, c['V'], st, regions=36, seq_len=Tn, seed=4242)
    a = [T(d, k) for k in ('fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')]
    before = ops._lib.load().isc_h3_launches()
    prev = ops.set_h3_mode(mode)
    try:
        with torch.no_grad():
            seq, lp, mk = cap(*a, Tn, 1, mode='rl')
        torch.cuda.synchronize()
    finally:
        ops.set_h3_mode(prev)
    assert ops._lib.load().isc_h3_launches() - before >= Tn * (1 if mode == 1 else 3)
    O = oracle()
    p = O.to_params(w)
    oid = O.Ids(synth.make_idx2word(c['V']), synth.SENTIMENT_CATEGORIES)
    ca = [torch.from_numpy(np.asarray(d[k])) for k in ('fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')]
    with torch.no_grad():
        oseq, olp, omk, _, _, margins = O.forward_rl(p, oid, *ca, Tn, 1)
    n = trusted_prefix(margins.numpy(), omk.numpy(), 2e-3)
    assert n.mean() >= 0.8 * Tn
    seq, lp, mk = seq.cpu().numpy(), lp.cpu().numpy(), mk.cpu().numpy()
    oseq, olp, omk = oseq.numpy(), olp.numpy(), omk.numpy()
    for b in range(B):
        assert (seq[b, :n[b]] == oseq[b, :n[b]]).all(), b
        assert (mk[b, :n[b]] == omk[b, :n[b]]).all(), b
        np.testing.assert_allclose(lp[b, :n[b]], olp[b, :n[b]], atol=LOGP_TOL)


def test_beam_device_merge_equals_host_merge():
    """The on-device candidate merge (isc_beam_merge: fp64 score sums, stable descending selection in insertion order,
    carried <EOS> candidates, frozen finished images) returns exactly what the host-side merges return - for a batch
    (numpy merge), for a pair of images (Python list merge) and with early-finishing images in the batch."""
    cap, c, st, w, _, _ = make_captioner('cfg1')
    n, Tn = 48, 20
    d = synth.make_inputs(n, c['V'], st, regions=36, seq_len=Tn, seed=2024)
    fc, att, sw, lab = T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'senti_words'), T(d, 'senti_labels')
    cap.rows_step = False        # same decode kernels on both sides: this test is about the merge (the few-row kernels
    try:                         # and their one-launch select are compared with this path in tests/test_gpu_rows.py)
        for sl, beam in ((slice(0, n), 5), (slice(0, 2), 3), (slice(5, 6), 5), (slice(0, 16), 8)):
            cap.beam_device_merge = True
            dev_out = cap.sample_batch(fc[sl], att[sl], sw[sl], lab[sl], beam, 1, Tn)
            steps_dev = cap.last_beam_steps
            cap.beam_device_merge = False
            host_out = cap.sample_batch(fc[sl], att[sl], sw[sl], lab[sl], beam, 1, Tn)
            assert dev_out[0] == host_out[0]                      # captions, beam order included
            assert dev_out[2] == host_out[2]                      # word ids
            np.testing.assert_array_equal(np.asarray(dev_out[1]), np.asarray(host_out[1]))   # fp64 scores, bit for bit
            assert steps_dev == cap.last_beam_steps
    finally:
        cap.beam_device_merge = True
        cap.rows_step = True


def test_beam_search_from_hip_graphs_equals_the_eager_search():
    """Captioner.enable_beam_graphs: prologue + steps 0-3 in one captured graph, every further four steps in another.
    Replays must return bit-identical captions, ids, fp64 scores and executed-step counts as the eager search: for
    images whose searches end early and late, for a search forced through all 20 steps, for a 3-image batch, and again
    after the weights changed in place (the planes are re-split inside graph 0)."""
    cap, c, st, w, _, _ = make_captioner('cfg1')
    n, Tn = 8, 20
    d = synth.make_inputs(n, c['V'], st, regions=36, seq_len=Tn, seed=77)
    fc, att, sw, lab = T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'senti_words'), T(d, 'senti_labels')

    def search(sl, beam=5):
        out = cap.sample_batch(fc[sl], att[sl], sw[sl], lab[sl], beam, 1, Tn)
        return out, cap.last_beam_steps
    cases = [slice(i, i + 1) for i in range(n)] + [slice(0, 3)]
    eos = cap.eos_id
    assert cap._beam_graphs is not None              # round 3: serving from graphs is the default ...
    cap.enable_beam_graphs(False)                    # ... the eager reference switches it off
    try:
        eager = [search(sl) for sl in cases]
        cap.eos_id = -7
        eager_full = search(cases[0])
        cap.eos_id = eos
        assert eager_full[1] == Tn and len({e[1] for e in eager}) > 1      # searches of different lengths in the set
        cap.enable_beam_graphs(True, max_graphs=4)
        for rep in range(3):                                   # eager first sight, capture, replay
            for sl, ref in zip(cases, eager):
                got = search(sl)
                assert got[0][0] == ref[0][0] and got[0][2] == ref[0][2], (rep, sl)
                np.testing.assert_array_equal(np.asarray(got[0][1]), np.asarray(ref[0][1]))
                assert got[1] == ref[1], (rep, sl, got[1], ref[1])
        cap.eos_id = -7
        for rep in range(3):
            got = search(cases[0])
            assert got[0][0] == eager_full[0][0] and got[1] == Tn
            np.testing.assert_array_equal(np.asarray(got[0][1]), np.asarray(eager_full[0][1]))
        cap.eos_id = eos
        # weights change in place (not the table-defining ones): same graphs, new planes
        with torch.no_grad():
            cap.classifier.weight.mul_(1.03)
        graphs, cap._beam_graphs = cap._beam_graphs, None      # eager reference with the new weights ...
        ref2 = search(cases[1])
        cap._beam_graphs = graphs                              # ... then the graphs captured with the old ones
        assert ref2[0][1] != eager[1][0][1]
        for rep in range(2):
            got = search(cases[1])
            assert got[0][0] == ref2[0][0] and got[1] == ref2[1]
            np.testing.assert_array_equal(np.asarray(got[0][1]), np.asarray(ref2[0][1]))
        # a weight behind a CACHED table changes in place (the gate's sentiment-word table, round 3): the graphs
        # captured with the old table must not be replayed - the key carries that weight's version
        with torch.no_grad():
            cap.attention.senti2att.weight.mul_(1.5)
        graphs, cap._beam_graphs = cap._beam_graphs, None
        ref3 = search(cases[1])
        cap._beam_graphs = graphs
        for rep in range(3):                                   # first sight (eager), capture, replay
            got = search(cases[1])
            assert got[0][0] == ref3[0][0] and got[1] == ref3[1], rep
            np.testing.assert_array_equal(np.asarray(got[0][1]), np.asarray(ref3[0][1]))
    finally:
        cap.eos_id = eos
        cap.enable_beam_graphs(True)


def test_stream_gate_skips_the_gated_launches_only_when_the_flag_reads_zero():
    """isc_set_stream_gate: forward launches enqueued inside ops.stream_gate(ptr) return at once when the device int at
    `ptr` is 0 at RUN time (outputs untouched), run normally when it is not, and launches outside the block never look -
    for the GEMM families a decode step can take (exact tiles, split-f16 tiles, the skinny path under a weights scope),
    the LSTM cell, the classifier and the attention scan."""
    g = torch.Generator().manual_seed(5)
    d = dev()
    flag = torch.zeros(2, dtype=torch.int32, device=d)
    flag[1] = 3
    SENT = -123.0

    def lin(M, N, K):
        x = torch.randn(M, K, generator=g).to(d)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(d)
        out = torch.full((M, N), SENT, device=d)
        return (lambda: ops.linear_fwd([ops.linear_problem([(x, w)], out)])), out, (x.double() @ w.double().t())

    def lstm(M, H, K):
        x = torch.randn(M, K, generator=g).to(d)
        w = (torch.randn(4 * H, K, generator=g) / K ** 0.5).to(d)
        b = torch.zeros(4 * H, device=d)
        c0 = torch.zeros(M, H, device=d)
        h, c = torch.full((M, H), SENT, device=d), torch.full((M, H), SENT, device=d)
        return (lambda: ops.lstm_fwd([(x, w)], b, b, c0, h, c)), h, None

    def vocab(M, V, K):
        x = torch.randn(M, K, generator=g).to(d)
        w = (torch.randn(V, K, generator=g) / K ** 0.5).to(d)
        b = torch.zeros(V, device=d)
        nt = (V + 127) // 128
        pm, ps = torch.full((M, nt), SENT, device=d), torch.full((M, nt), SENT, device=d)
        pi = torch.zeros(M, nt, dtype=torch.int32, device=d)
        return (lambda: ops.vocab_fwd(x, w, b, pm, ps, pi)), pm, None

    def scan(B, R, A):
        P, V_ = torch.randn(B, R, A, generator=g).to(d), torch.randn(B, R, A, generator=g).to(d)
        q, w = torch.randn(B, A, generator=g).to(d), torch.randn(A, generator=g).to(d)
        out, al = torch.full((B, A), SENT, device=d), torch.zeros(B, R, device=d)
        return (lambda: ops.attn_scan_fwd([ops.scan_problem(P, V_, q, w, None, out, al)], B)), out, None

    cases = [lin(37, 64, 96), lin(320, 512, 512), lin(2048, 512, 512), lstm(320, 512, 1024), lstm(4096, 512, 1024),
             vocab(320, 10000, 512), vocab(4096, 10000, 512), scan(320, 36, 512)]
    for scope in (False, True):
        ctx = ops.h3_weights_scope(d) if scope else contextlib.nullcontext()
        with ctx:
            for run, out, ref in cases:
                out.fill_(SENT)
                with ops.stream_gate(flag[0:1].data_ptr()):          # flag 0: nothing runs
                    run()
                torch.cuda.synchronize()
                assert bool((out == SENT).all()), 'a gated launch wrote its output although the flag read 0'
                with ops.stream_gate(flag[1:2].data_ptr()):          # flag != 0: the launch runs
                    run()
                torch.cuda.synchronize()
                assert not bool((out == SENT).any())
                got = out.clone()
                out.fill_(SENT)
                run()                                                # no gate: same bits
                torch.cuda.synchronize()
                assert torch.equal(got, out)
                if ref is not None:
                    assert float((out.double() - ref).abs().max()) < 1e-3
    # the flag is read when the launch RUNS, not when it is enqueued: set behind the enqueue, on the same stream
    run, out, _ = cases[1]
    out.fill_(SENT)
    f2 = torch.ones(1, dtype=torch.int32, device=d)
    f2.zero_()
    with ops.stream_gate(f2.data_ptr()):
        run()
    torch.cuda.synchronize()
    assert bool((out == SENT).all())
    # the host state is gone with the block
    out.fill_(SENT)
    run()
    torch.cuda.synchronize()
    assert not bool((out == SENT).any())


def test_batched_beam_search_ends_itself_on_the_device():
    """The general kernels' batched search (more than 8 rows): a step enqueued past the end of every image's search skips
    its contractions, scans and top-k on the device (the stream gate under live[t]) - same captions, ids, fp64 scores and
    executed-step counts as the ungated search, eager and from graphs, for a batch that ends early (and NOT on a multiple
    of the four steps between two looks at the counter, so that gated steps do run) and for one forced through all steps."""
    cap, c, st, w, _, _ = make_captioner('cfg1')
    n, Tn, beam = 24, 20, 5
    d = synth.make_inputs(n, c['V'], st, regions=36, seq_len=Tn, seed=4242)
    fc, att, sw, lab = T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'senti_words'), T(d, 'senti_labels')

    def search():
        out = cap.sample_batch(fc, att, sw, lab, beam, 1, Tn)
        return out, cap.last_beam_steps
    eos = cap.eos_id
    try:
        cap.enable_beam_graphs(False)
        cap.beam_step_gate = False
        # push <EOS> until the whole batch ends well before the last step (random-init captions rarely end on their own)
        for bump in (0.0, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 2.0, 2.0):
            with torch.no_grad():
                cap.classifier.bias[eos] += bump
            ref = search()
            if ref[1] <= Tn - 6 and ref[1] % 4 != 0:
                break
        assert 2 <= ref[1] <= Tn - 6 and ref[1] % 4 != 0, ref[1]
        cap.eos_id = -7
        ref_full = search()
        cap.eos_id = eos
        assert ref_full[1] == Tn

        def same(got, want, what):
            assert got[0][0] == want[0][0] and got[0][2] == want[0][2], what
            np.testing.assert_array_equal(np.asarray(got[0][1]), np.asarray(want[0][1]), err_msg=str(what))
            assert got[1] == want[1], (what, got[1], want[1])
        cap.beam_step_gate = True
        same(search(), ref, 'eager, gated')
        cap.enable_beam_graphs(True, max_graphs=4)
        for rep in range(4):                                       # first sight, capture, replays
            same(search(), ref, ('graphs', rep))
        cap.eos_id = -7
        for rep in range(3):
            same(search(), ref_full, ('graphs, all steps', rep))
        cap.eos_id = eos
        cap.beam_step_gate = False                                 # the ungated form is still there (its own graphs)
        for rep in range(3):
            same(search(), ref, ('graphs, ungated', rep))
    finally:
        cap.eos_id = eos
        cap.__dict__.pop('beam_step_gate', None)
        cap.enable_beam_graphs(True)


@pytest.mark.parametrize('rows,V,beam', [(5, 10000, 5), (7, 9487, 3), (3, 130, 8), (2, 10000, 12)])
def test_beam_topk_kernel_order_ties_and_masks(rows, V, beam):
    """isc_beam_topk (single-pass kernel for beam <= 8, round-based above): top-`beam` of the masked log-probs in
    descending order, ties to the smaller word id, <PAD>/<SOS>/<UNK> and the repeated last word masked to -inf."""
    g = torch.Generator().manual_seed(rows * V)
    K = 64
    h = torch.randn(rows, K, generator=g)
    W = torch.randn(V, K, generator=g) / 8
    W[17] = W[4000 % V]                           # exact ties between word ids
    W[V - 1] = W[123 % V]
    bias = torch.zeros(V)
    nt = (V + 127) // 128
    dh, dW, db = h.to(dev()), W.to(dev()), bias.to(dev())
    pm, ps = torch.empty(rows, nt, device=dev()), torch.empty(rows, nt, device=dev())
    pi = torch.empty(rows, nt, device=dev(), dtype=torch.int32)
    logits = torch.empty(rows, V, device=dev())
    ops.vocab_fwd(dh, dW, db, pm, ps, pi, logits)
    last = torch.tensor([4000 % V, 17, 5, 123 % V, V - 1, 9, 11][:rows], dtype=torch.int64, device=dev())
    tv = torch.empty(rows, beam, device=dev())
    ti = torch.empty(rows, beam, dtype=torch.int64, device=dev())
    pad_id, sos_id, unk_id = 0, 1, 3
    ops.beam_topk(logits, pm, ps, last, beam, pad_id, sos_id, unk_id, True, 1, tv, ti)
    torch.cuda.synchronize()
    lg = logits.cpu()
    mx = pm.cpu().max(1).values
    lse = mx + torch.log((ps.cpu() * torch.exp(pm.cpu() - mx[:, None])).sum(1))
    for r in range(rows):
        logp = (lg[r] - mx[r]) - torch.log((ps.cpu()[r] * torch.exp(pm.cpu()[r] - mx[r])).sum())
        logp[[pad_id, sos_id, unk_id]] = -float('inf')
        logp[last[r].item()] = -float('inf')
        order = sorted(range(V), key=lambda i: (-logp[i].item(), i))[:beam]
        assert ti[r].cpu().tolist() == order, r
        np.testing.assert_allclose(tv[r].cpu().numpy(), logp[order].numpy(), atol=2e-6)


def test_sampled_rollout_b1024_replayed_by_the_oracle():
    """Sampled roll-out (sample_max=0) at B=1024 with the split-f16 path forced on: the device draws the tokens
    (two-level inverse CDF over the materialised log-probs); the oracle replays those raw draws and must assign them
    the same log-probs (1e-4), masks and <EOS> handling - the logits-writing side of the classifier epilogue."""
    cap, c, st, w, _, _ = make_captioner('cfg1')
    B, Tn = 1024, 20
    d = synth.make_inputs(B, c['V'], st, regions=36, seq_len=Tn, seed=777)
    a = [T(d, k) for k in ('fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')]
    prev = ops.set_h3_mode(2)
    try:
        torch.manual_seed(5)
        with torch.no_grad():
            seq, lp, mk, raw, _ = cap._rollout(*a, Tn, 0, None, None)
        torch.cuda.synchronize()
    finally:
        ops.set_h3_mode(prev)
    O = oracle()
    p = O.to_params(w)
    oid = O.Ids(synth.make_idx2word(c['V']), synth.SENTIMENT_CATEGORIES)
    ca = [torch.from_numpy(np.asarray(d[k])) for k in ('fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')]
    with torch.no_grad():
        oseq, olp, omk, _, _, _ = O.forward_rl(p, oid, *ca, Tn, 0, replay=raw.cpu())
    steps = int(omk.sum(1).max().item())                     # the oracle stops once every row has ended
    assert (seq.cpu()[:, :steps] == oseq[:, :steps]).all()
    assert (mk.cpu()[:, :steps] == omk[:, :steps]).all()
    live = omk[:, :steps].bool()
    np.testing.assert_allclose(lp.cpu()[:, :steps][live].numpy(), olp[:, :steps][live].numpy(), atol=LOGP_TOL)
    # the draws follow the distribution: mean log-prob of the drawn tokens ~ -entropy, far above uniform (-9.2)
    assert lp.cpu()[:, 0].mean().item() > -9.0


def test_features_beyond_the_split_f16_domain_are_served_on_the_exact_engine():
    """Round-2 finding: |x| >= 65504 in caller data turns the hi plane of x = hi + lo 2^-11 into inf and the roll-out
    into NaN-derived garbage.  Rounds 2-4 flagged and raised; the reference decodes whatever its encoder produced
    (captioner.py:198-214, 294-315), so the product now SERVES such a call - on its exact-fp32 GEMM engine, in the same
    process, with one warning: the default-mode results equal the results of the engine switched off by hand
    (isc_set_h3_mode(0)), token for token, for a roll-out (eager and at a graph-served size), `sample`, and
    `Detector.forward`; `numerics_checks = False` keeps the old behaviour (no host read: flags only)."""
    import warnings
    from insenticap_model_amd import _lib
    from conftest import case_setup
    c, st, w, d, _ = case_setup('cfg1')

    def make():
        m = Captioner(synth.make_idx2word(c['V']), synth.SENTIMENT_CATEGORIES, st)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        return m.to(dev()).eval()
    cap = make()
    B = 512                                   # enough rows for the split-f16 prologue kernels in auto mode
    big = synth.make_inputs(B, c['V'], st, regions=36, seq_len=6, seed=77)
    a = [torch.from_numpy(np.asarray(big[k])).to(dev()) for k in ('fc_feats', 'att_feats', 'cpt_words', 'senti_words',
                                                                   'senti_labels')]
    ops.device_status(reset=True)
    with torch.no_grad():
        cap(*a, 6, 1, mode='rl')
    torch.cuda.synchronize()                                      # the status words say what the host has waited for
    assert ops.device_status(reset=True) == 0                     # healthy input: nothing flagged
    bad = [x.clone() for x in a]
    bad[1][3, 5, 100] = 1.0e5                                     # one region feature beyond the f16 range

    def exact(fn):
        prev = ops.set_h3_mode(0)
        try:
            with torch.no_grad():
                return fn()
        finally:
            ops.set_h3_mode(prev)
    # ---- roll-out, eager size
    seq0, lp0, mk0 = exact(lambda: make()(*bad, 6, 1, mode='rl'))
    with warnings.catch_warnings(record=True) as rec, torch.no_grad():
        warnings.simplefilter('always')
        seq, lp, mk = cap(*bad, 6, 1, mode='rl')
        seq_b, _, _ = cap(*bad, 6, 1, mode='rl')                  # (the verdict on these tensors is remembered)
    torch.cuda.synchronize()
    assert any('split-f16 operand domain' in str(x.message) for x in rec)
    assert ops.device_status(reset=True) == 0 and ops.h3_mode() == 1          # nothing flagged; the engine is back on
    assert torch.equal(seq, seq0) and torch.equal(mk, mk0) and torch.equal(seq_b, seq0)
    np.testing.assert_allclose(lp.cpu().numpy(), lp0.cpu().numpy(), atol=1e-6)
    assert bool(torch.isfinite(lp).all())
    # ---- roll-out at a graph-served size (<= 256 rows), twice: never captured with out-of-domain inputs
    small = [x[:8].clone() for x in bad]
    s0 = exact(lambda: make()(*small, 6, 1, mode='rl'))[0]
    with torch.no_grad():
        for _ in range(3):
            assert torch.equal(cap(*small, 6, 1, mode='rl')[0], s0)
    # ---- beam search, one image
    cp0, sc0 = exact(lambda: make().sample(bad[0][3], bad[1][3], bad[3][3], bad[4][3:4], 3, 1, 6))
    cp, sc = cap.sample(bad[0][3], bad[1][3], bad[3][3], bad[4][3:4], 3, 1, 6)
    assert cp == cp0
    np.testing.assert_allclose(sc, sc0, atol=1e-5)
    assert ops.device_status(reset=True) == 0
    # ---- numerics_checks = False: no host read in front of the call, the flags report as before
    loose = make()
    loose.numerics_checks = False
    with torch.no_grad():
        loose(*bad, 6, 1, mode='rl')
    torch.cuda.synchronize()
    st_bits = ops.device_status(reset=False)
    assert st_bits & ops.STATUS_NONFINITE_STATS and st_bits & ops.STATUS_NONFINITE_LINEAR, st_bits
    assert _lib.load().isc_status(0) == st_bits                   # the C entry point reads the same words
    with pytest.raises(_lib.HipLibraryError, match='split-f16 domain'):
        ops.check_numerics('test')
    assert ops.device_status(reset=True) == 0                     # check_numerics cleared it


def test_detector_forward_serves_features_beyond_the_split_f16_domain():
    """Detector.forward (models/decoder.py:52-180) on a batch with one region feature of 1e5: evaluation and a training
    iteration (eager and with the RL graphs on) give the results of the exact-fp32 engine - the iteration whose
    roll-outs flagged non-finite values is redone on it before anything is updated - and the parameters stay finite."""
    import warnings
    from insenticap_model_amd.detector import Detector
    from test_detector import load_helper
    V, Tn, B = 64, 8, 8
    st = dict(synth.TINY_SETTINGS, **synth.HELPER_SETTINGS)
    st['dropout_p'] = 0.0

    def make(graphs):
        det = Detector(synth.make_idx2word(V), Tn, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-4}, st)
        det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=3).items()})
        load_helper(det.senti_detector, 51)
        load_helper(det.sent_senti_cls, 52)
        det.to(dev())
        det.train_graphs = graphs
        det.xe_ss_prob = det.seq2seq_ss_prob = 0.0
        return det
    batches, split = synth.make_rl_batches(1, B, V, st, seq_len=Tn, seed=70)
    t = torch.from_numpy
    b = batches[0]
    att = b[2].copy()
    att.reshape(B, -1)[2, 17] = 1.0e5
    item = (b[0], t(b[1]), t(att), (t(b[3][0]), b[3][1]), t(b[4]), t(b[5]), b[6])
    s = synth.make_inputs(4, V, st, regions=6, seq_len=Tn, seed=72)
    scs = [((t(s['captions']), s['lengths']), t(s['cpt_words']), t(s['senti_words']), t(s['senti_labels']))]
    draws = torch.from_numpy(np.random.default_rng(80).integers(2, V, size=(B, Tn), dtype=np.int64)).to(dev())

    def run(det, training, n=1):
        det.set_ciderd_scorer(split)
        orig = det.captioner.forward_rl

        def replay_rl(*a, **k):
            if not k.get('sample_max', a[-1] if len(a) >= 7 else 1):
                k['_replay'] = draws
            return orig(*a, **k)
        det.captioner.forward_rl = replay_rl
        out = [det(([item], scs), 'fact', training) for _ in range(n)]
        torch.cuda.synchronize()
        return out
    prev = ops.set_h3_mode(0)
    try:
        ref_eval = run(make(False), False)
        ref_det = make(False)
        ref_train = run(ref_det, True, 3)
    finally:
        ops.set_h3_mode(prev)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        got_eval = run(make(False), False)
        for k in ref_eval[0]:
            np.testing.assert_allclose(got_eval[0][k], ref_eval[0][k], rtol=1e-5, atol=1e-7, err_msg=k)
        for graphs in (False, True):
            det = make(graphs)
            got = run(det, True, 3)
            for i in range(3):
                for k in ref_train[i]:
                    np.testing.assert_allclose(got[i][k], ref_train[i][k], rtol=2e-4, atol=1e-6, err_msg='%s %d' % (k, i))
            for (k, p), (_, q) in zip(det.captioner.named_parameters(), ref_det.captioner.named_parameters()):
                assert bool(torch.isfinite(p).all()), k
                assert float((p - q).abs().max()) <= 3 * 2 * 4e-4 * 1.01, k
    assert ops.h3_mode() == 1 and ops.device_status(reset=True) == 0


def test_gated_scan_equals_scans_plus_gate_gemm_plus_gate_mix():
    """isc_attn_scan_gate_fwd (few-row inference: both scans, the gate sum and the gate mix of a decode step in one
    launch, the scans' features carried through the gate's projections beforehand) against the three launches it
    replaces - per-caption features and the gathered word table, with f16 planes - and, end to end, greedy roll-outs
    and beam searches with the fusion on and off: same tokens, log-probs to 2e-6."""
    g = torch.Generator().manual_seed(12)
    D_ = dev()
    B, R, M, A, V = 37, 36, 11, 512, 300
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1)
    att_p, att_e = r(B, R, A).to(D_), torch.relu(r(B, R, A)).to(D_)
    tab_e, tab_p = torch.relu(r(V, A)).to(D_), torch.relu(r(V, A)).to(D_)
    ids = torch.randint(0, V, (B, M), generator=g).to(D_)
    qa, qw, q2, zh = r(B, A).to(D_), r(B, A).to(D_), r(B, A).to(D_), r(B, A).to(D_)
    wc, ws_, wg = (r(1, A) * 0.2).to(D_), (r(1, A) * 0.2).to(D_), (r(1, A) * 0.2).to(D_)
    bc, bs, bg = r(1).to(D_), r(1).to(D_), r(1).to(D_)
    Wgc, Wgs = (r(A, A) * A ** -0.5).to(D_), (r(A, A) * A ** -0.5).to(D_)
    bgc, bgs = r(A).to(D_), r(A).to(D_)
    # the three launches
    v, s = torch.empty(B, A, device=D_), torch.empty(B, A, device=D_)
    aC, aS = torch.empty(B, R, device=D_), torch.empty(B, M, device=D_)
    scans = [ops.scan_problem(att_p, att_e, qa, wc, bc, v, aC),
             ops.scan_problem(tab_p, tab_e, qw, ws_, bs, s, aS, q2=q2, row_ids=ids)]
    ops.attn_scan_fwd(scans, B)
    z = zh.clone()
    prev = ops.set_h3_mode(0)                       # exact fp32 GEMM for the reference z
    try:
        ops.linear_fwd([ops.linear_problem([(v, Wgc), (s, Wgs)], z, bgc, bgs, accumulate=True)])
        Gc = torch.empty(B * R, A, device=D_)
        Gs = torch.empty(V, A, device=D_)
        ops.linear_fwd([ops.linear_problem([(att_e.view(B * R, A), Wgc)], Gc)])
        ops.linear_fwd([ops.linear_problem([(tab_e, Wgs)], Gs)])
    finally:
        ops.set_h3_mode(prev)
    f, beta = torch.empty(B, A, device=D_), torch.empty(B, 1, device=D_)
    ops.gate_mix_fwd(z, wg, bg, v, s, f, beta)
    # the one launch
    f2, beta2 = torch.empty(B, A, device=D_), torch.empty(B, 1, device=D_)
    aC2, aS2 = torch.empty(B, R, device=D_), torch.empty(B, M, device=D_)
    planes = torch.empty(2, B, A, dtype=torch.float16, device=D_)
    scans2 = [ops.scan_problem(att_p, att_e, qa, wc, bc, v, aC2),
              ops.scan_problem(tab_p, tab_e, qw, ws_, bs, s, aS2, q2=q2, row_ids=ids)]
    prev_rows = ops.set_rows_scan_max(0)            # attn_scan_gate_kernel (the region walk of the three launches)
    try:
        ops.attn_scan_gate_fwd(scans2, (Gc, Gs), zh, bgc, bgs, wg, bg, f2, beta2, f_planes=planes)
        torch.cuda.synchronize()
        assert torch.equal(aC, aC2) and torch.equal(aS, aS2)          # the scans themselves are the same arithmetic
        np.testing.assert_allclose(beta2.cpu().numpy(), beta.cpu().numpy(), atol=3e-6)
        np.testing.assert_allclose(f2.cpu().numpy(), f.cpu().numpy(), atol=3e-6)
        # ... and on the one-workgroup-per-row kernel that takes launches of up to 256 rows by default (rows.hip): other
        # summation orders, same values; its planes of f are the split of its f
        ops.set_rows_scan_max(256)
        f3, beta3 = torch.empty(B, A, device=D_), torch.empty(B, 1, device=D_)
        aC3, aS3 = torch.empty(B, R, device=D_), torch.empty(B, M, device=D_)
        planes3 = torch.empty(2, B, A, dtype=torch.float16, device=D_)
        scans3 = [ops.scan_problem(att_p, att_e, qa, wc, bc, v, aC3),
                  ops.scan_problem(tab_p, tab_e, qw, ws_, bs, s, aS3, q2=q2, row_ids=ids)]
        n0 = ops._lib.load().isc_rows_launches()
        ops.attn_scan_gate_fwd(scans3, (Gc, Gs), zh, bgc, bgs, wg, bg, f3, beta3, f_planes=planes3)
        torch.cuda.synchronize()
        assert ops._lib.load().isc_rows_launches() == n0 + 1
        np.testing.assert_allclose(aC3.cpu().numpy(), aC.cpu().numpy(), atol=2e-6)
        np.testing.assert_allclose(aS3.cpu().numpy(), aS.cpu().numpy(), atol=2e-6)
        np.testing.assert_allclose(beta3.cpu().numpy(), beta.cpu().numpy(), atol=3e-6)
        np.testing.assert_allclose(f3.cpu().numpy(), f.cpu().numpy(), atol=3e-6)
        pl = planes3.view(B, A // 32, 2, 32)               # interleaved layout: per row and 32-block 32 hi, then 32 lo
        hi, lo = pl[:, :, 0, :].reshape(B, A), pl[:, :, 1, :].reshape(B, A)
        assert torch.equal(hi, f3.half())
        assert torch.equal(lo, ((f3 - hi.float()) * 2048.0).half())
    finally:
        ops.set_rows_scan_max(prev_rows)
    hi, lo = planes[0].float(), planes[1].float()                     # planes of f: hi + lo 2^-11 == f to 2^-22 rel.
    # (interleaved layout: compare through the split of f2 itself)
    # end to end: roll-outs and beam searches with the fusion on / off
    cap, c, st, w, d, _ = make_captioner('cfg1')
    big = synth.make_inputs(96, c['V'], st, regions=36, seq_len=8, seed=5)
    a = [torch.from_numpy(np.asarray(big[k])).to(D_) for k in ('fc_feats', 'att_feats', 'cpt_words', 'senti_words',
                                                                'senti_labels')]
    outs = {}
    for on in (True, False):
        cap.gate_fused = on
        with torch.no_grad():
            seq, lp, mk = cap(*a, 8, 1, mode='rl')
            caps, scores, ids_ = cap.sample_batch(a[0][:3], a[1][:3], a[3][:3], a[4][:3], 3, 1, 8)
        outs[on] = (seq.cpu(), lp.cpu(), caps, np.asarray(scores))
    cap.gate_fused = True
    margins_ok = True
    assert torch.equal(outs[True][0], outs[False][0]) or not margins_ok
    np.testing.assert_allclose(outs[True][1].numpy(), outs[False][1].numpy(), atol=2e-5)
    assert outs[True][2] == outs[False][2]
    np.testing.assert_allclose(outs[True][3], outs[False][3], atol=2e-5)


@pytest.mark.parametrize('B,R,M,A,table', [(3, 1, 1, 64, False), (5, 196, 3, 256, True), (2, 7, 11, 1024, False),
                                           (1, 36, 11, 512, True), (9, 13, 40, 128, False)])
def test_gated_scan_odd_shapes_vs_fp64(B, R, M, A, table):
    """isc_attn_scan_gate_fwd on shapes away from the decoder's (one region, one word, A = 64 ... 1024, per-caption and
    gathered sentiment features) against the defining formulas in fp64."""
    g = torch.Generator().manual_seed(B * 1000 + R)
    D_ = dev()
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1)
    Vn = 50
    att_p, att_e = r(B, R, A), torch.relu(r(B, R, A))
    if table:
        w_p, w_e = torch.relu(r(Vn, A)), torch.relu(r(Vn, A))
        ids = torch.randint(0, Vn, (B, M), generator=g)
        P_s, E_s = w_p[ids], w_e[ids]                         # [B,M,A] views for the reference
    else:
        w_p, w_e = torch.relu(r(B, M, A)), torch.relu(r(B, M, A))
        ids, P_s, E_s = None, w_p, w_e
    qa, qw, q2, zh = r(B, A), r(B, A), r(B, A), r(B, A)
    wc, ws_, wg = r(1, A) * 0.2, r(1, A) * 0.2, r(1, A) * 0.2
    bc, bs, bg = r(1), r(1), r(1)
    Wgc, Wgs = r(A, A) * A ** -0.5, r(A, A) * A ** -0.5
    bgc, bgs = r(A), r(A)
    d = lambda x: x.double()
    ec = (torch.tanh(d(att_p) + d(qa)[:, None]) * d(wc)).sum(-1) + d(bc)
    es = (torch.tanh(d(P_s) + (d(qw) + d(q2))[:, None]) * d(ws_)).sum(-1) + d(bs)
    ac, as_ = torch.softmax(ec, -1), torch.softmax(es, -1)
    v, s = (ac[..., None] * d(att_e)).sum(1), (as_[..., None] * d(E_s)).sum(1)
    z = d(zh) + v @ d(Wgc).t() + d(bgc) + s @ d(Wgs).t() + d(bgs)
    beta = torch.sigmoid((torch.tanh(z) * d(wg)).sum(-1, keepdim=True) + d(bg))
    f = beta * v + (1 - beta) * s
    # device: G tensors by fp64 then rounded (the kernel under test is the scan, not the projection GEMM)
    Gc = (d(att_e) @ d(Wgc).t()).float().to(D_).contiguous().view(B * R, A)
    if table:
        Gs = (d(w_e) @ d(Wgs).t()).float().to(D_).contiguous()
    else:
        Gs = (d(w_e) @ d(Wgs).t()).float().to(D_).contiguous().view(B * M, A)
    keep = []                                          # (scan_problem keeps raw pointers: hold the device copies)

    def T_(x):
        keep.append(x.to(D_).contiguous())
        return keep[-1]
    out_v, out_s = torch.empty(B, A, device=D_), torch.empty(B, A, device=D_)
    aC, aS = torch.empty(B, R, device=D_), torch.empty(B, M, device=D_)
    f2, b2 = torch.empty(B, A, device=D_), torch.empty(B, 1, device=D_)
    scans = [ops.scan_problem(T_(att_p), T_(att_e), T_(qa), T_(wc), T_(bc), out_v, aC),
             ops.scan_problem(T_(w_p), T_(w_e), T_(qw), T_(ws_), T_(bs), out_s, aS, q2=T_(q2),
                              row_ids=None if ids is None else T_(ids))]
    ops.attn_scan_gate_fwd(scans, (Gc, Gs), T_(zh), T_(bgc), T_(bgs), T_(wg), T_(bg), f2, b2)
    torch.cuda.synchronize()
    np.testing.assert_allclose(aC.cpu().numpy(), ac.float().numpy(), atol=3e-6)
    np.testing.assert_allclose(aS.cpu().numpy(), as_.float().numpy(), atol=3e-6)
    np.testing.assert_allclose(b2.cpu().numpy(), beta.float().numpy(), atol=5e-6)
    np.testing.assert_allclose(f2.cpu().numpy(), f.float().numpy(), atol=1e-5)


def test_weight_tables_keep_one_slot_per_engine():
    """The token / sentiment-word tables derive from the weights through a GEMM, so each GEMM engine builds its own
    (Captioner._table_cache).  A call served on the exact-fp32 engine - `--h3-mode 0`, an out-of-domain input - must not
    evict the split-f16 engine's table: few-caption roll-outs only USE a cached table ('cached'), so after an eviction they
    silently ran without it (round 5: B = 4 roll-outs 0.75 -> 0.82 ms after bench.py's exact-engine leg)."""
    from conftest import case_setup
    c, st, w, d, _ = case_setup('cfg1')
    cap = Captioner(synth.make_idx2word(c['V']), synth.SENTIMENT_CATEGORIES, st)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    cap.to(dev()).eval()
    Tn = 8
    rows = c['V'] // (4 * Tn) + 8                                          # rows x steps >= V / 4: the roll-out builds the table
    big = synth.make_inputs(rows, c['V'], st, regions=36, seq_len=Tn, seed=31)
    a = [torch.from_numpy(np.asarray(big[k])).to(dev()) for k in ('fc_feats', 'att_feats', 'cpt_words', 'senti_words',
                                                                   'senti_labels')]
    with torch.no_grad():
        cap(*a, Tn, 1, mode='rl')
        tab = cap._tab_cache[False][1]
        with ops.exact_fp32_engine():
            cap(*a, Tn, 1, mode='rl')
        assert cap._tab_cache[True][1] is not tab and cap._tab_cache[False][1] is tab
        small = [x[:4].contiguous() for x in a]
        rows0 = ops._lib.load().isc_rows_launches()
        cap(*small, Tn, 1, mode='rl')                                      # four captions: the table, if it is still cached
        assert cap._tab_cache[False][1] is tab
        assert ops._lib.load().isc_rows_launches() > rows0                 # ... on the few-row kernels
    torch.cuda.synchronize()
