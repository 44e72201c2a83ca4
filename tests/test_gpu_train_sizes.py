"""The XE training iteration (train_xe.py:160-192) at the sizes whose backward takes the size-dependent routes, against
the CPU oracle's autograd - one iteration each, every gradient tensor:

  * BASELINE configs[3]'s single-GPU point: B=1024 (+80 seq2seq rows), V=10k, T=20, 36x2048 features.  Above B=128 the
    backward changes engine - classifier dX zero-padded / K-split on the split-f16 tiles, K-chunked TN contractions,
    non-temporal loads in the scan backward above 128 MB, indexed embedding backward - and until round 3 only
    kernel-level fp64 tests crossed those routes.
  * the reference encoder's own feature grid: 14 x 14 = 196 regions (models/encoder.py att_size=14), where the
    post-sweep dP / dV kernels walk several region chunks (round-2 advisor finding: they refused R > 36).

Run with pytest -m gpu."""
import numpy as np
import pytest
import torch

from insenticap_model_amd import Captioner, XECriterion, clip_gradient, synth

pytestmark = pytest.mark.gpu
GRAD_RTOL = 1e-4            # gradients within 1e-4 relative to the tensor's max
DEV = 'cuda:0'


def _oracle_iteration(w, V, d, s2s, chunk=256, dtype=torch.float32):
    """xe + domain-align + seq2seq losses and the gradient of their sum by the oracle's autograd, in row chunks (each
    loss is a sum over rows divided by a global normaliser, so chunk gradients add).  dtype float64: the same
    restatement as the arbiter between two fp32 implementations."""
    from oracle import captioner_oracle as O
    torch.set_num_threads(max(1, min(64, len(__import__('os').sched_getaffinity(0)))))
    p = O.to_params(w, dtype=dtype, requires_grad=True)
    ids = O.Ids(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES)

    def t(a):
        x = torch.from_numpy(np.ascontiguousarray(a))
        return x.to(dtype) if x.is_floating_point() else x
    losses = [0.0, 0.0, 0.0]
    n_tok, n_rows = float(sum(d['lengths'])), len(d['lengths'])
    E = None
    for lo in range(0, n_rows, chunk):
        hi = min(n_rows, lo + chunk)
        logp, P, _ = O.forward_xe(p, ids, t(d['fc_feats'][lo:hi]), t(d['att_feats'][lo:hi]), t(d['cpt_words'][lo:hi]),
                                  t(d['captions'][lo:hi]), t(d['senti_labels'][lo:hi]))
        lens = list(d['lengths'][lo:hi])
        xe = O.xe_criterion(logp[:, :max(lens)], t(d['captions'][lo:hi, 1:1 + max(lens)]), lens) * (sum(lens) / n_tok)
        da = O.domain_align_loss(P.cpt, P.fc_raw) * ((hi - lo) / n_rows)
        (xe + da).backward()
        losses[0] += float(xe.detach())
        losses[1] += float(da.detach())
        del logp, P, xe, da
    n_tok2 = float(sum(s2s['lengths']))
    for lo in range(0, len(s2s['lengths']), chunk):
        hi = min(len(s2s['lengths']), lo + chunk)
        logp, _, _ = O.forward_seq2seq(p, ids, t(s2s['captions'][lo:hi]), t(s2s['cpt_words'][lo:hi]),
                                       t(s2s['senti_words'][lo:hi]), t(s2s['senti_labels'][lo:hi]))
        lens = list(s2s['lengths'][lo:hi])
        l2 = O.xe_criterion(logp[:, :max(lens)], t(s2s['captions'][lo:hi, 1:1 + max(lens)]), lens) * (sum(lens) / n_tok2)
        l2.backward()
        losses[2] += float(l2.detach())
    grads = {k: q.grad.numpy() for k, q in p.items() if q.grad is not None}
    # one clamp + Adam step of the oracle on its own gradients (train_xe.py:191-192)
    params = {k: q.detach().clone() for k, q in p.items() if q.grad is not None}
    m = {k: torch.zeros_like(q) for k, q in params.items()}
    v = {k: torch.zeros_like(q) for k, q in params.items()}
    O.clamp_adam_step(params, {k: torch.from_numpy(g) for k, g in grads.items()}, m, v, 1, 4e-4)
    return losses, grads, {k: q.numpy() for k, q in params.items()}


def _hip_iteration(w, V, st, d, s2s):
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
    cap.load_state_dict({k: torch.from_numpy(x) for k, x in w.items()})
    cap.to(DEV).eval()                 # eval-mode dropout (identity), gradients still flow: SURVEY 8(d) config 2
    optim, _, _ = cap.get_optim_criterion(4e-4)
    T = lambda a, k: torch.from_numpy(np.asarray(a[k])).to(DEV)
    xe_crit, da_crit = XECriterion(), torch.nn.MSELoss()
    cap.zero_grad()
    pred = cap(T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'cpt_words'), T(d, 'captions'), T(d, 'senti_labels'), 0.0,
               mode='xe')
    xe = xe_crit(pred, T(d, 'captions')[:, 1:], d['lengths'])
    da = da_crit(cap.cpt_feats, cap.fc_feats.detach())
    pred2 = cap(T(s2s, 'captions'), T(s2s, 'cpt_words'), T(s2s, 'senti_words'), T(s2s, 'senti_labels'), 0.0,
                mode='seq2seq')
    l2 = xe_crit(pred2, T(s2s, 'captions')[:, 1:], s2s['lengths'])
    (xe + da + l2).backward()
    grads = {k: q.grad.detach().cpu().numpy().copy() for k, q in cap.named_parameters() if q.grad is not None}
    clip_gradient(optim, 0.1)
    optim.step()
    torch.cuda.synchronize()
    params = {k: q.detach().cpu().numpy() for k, q in cap.named_parameters()}
    return (float(xe.detach()), float(da.detach()), float(l2.detach())), grads, params


def _relu_sign_flips(w, d):
    """How many pre-activations of the two prologue ReLUs over the regions (att_embed: captioner.py:141-143, att2att:
    :149-150) fall on different sides of zero in fp32 and in fp64 arithmetic (CPU, the oracle's operations)."""
    att = torch.from_numpy(np.asarray(d['att_feats'])).reshape(-1, w['att_embed.0.weight'].shape[1])
    counts, x32, x64 = {}, att.float(), att.double()
    for name in ('att_embed', 'att2att'):
        W, b = torch.from_numpy(w[name + '.0.weight']), torch.from_numpy(w[name + '.0.bias'])
        p32 = x32 @ W.float().t() + b.float()
        p64 = x64 @ W.double().t() + b.double()
        counts[name] = int(((p32 > 0) != (p64 > 0)).sum())
        x32, x64 = torch.relu(p32), torch.relu(p64)
    return counts


def _compare(hip, ora, n_expected=32, ora64=None, widen=True):
    """Gradients within 1e-4 of the tensor's max.  With `ora64` (the fp64 oracle) the comparison is
    against IT, and a tensor on which the reference's own fp32 arithmetic is further than 1e-4/3 from the fp64 result
    gets three times that error as its bound: the HIP path must be as close to the exact gradient as the fp32
    reference is, to a factor.  At B=1024 that is `att2att` (fp32 reference: 1.8e-4): of the 18.9 M pre-activations of
    its ReLU a few lie within 1e-7 of zero, fp32 and fp64 put them on different sides (counted on the CPU: 1 in
    att2att, 2 in att_embed), and each flip adds or drops one whole row's contribution to the weight gradient -
    a kink of the function, not an arithmetic error; which elements flip depends on the rounding of the forward GEMM."""
    (hl, hg, hp), (ol, og, op_) = hip, ora
    np.testing.assert_allclose(hl, ol, rtol=3e-5)
    assert set(hg) == set(og) and len(og) == n_expected, (sorted(set(hg) ^ set(og)), len(og))
    worst = {}
    for k, ref32 in og.items():
        ref = ref32 if ora64 is None else ora64[1][k]
        scale = np.abs(ref).max()
        tol = GRAD_RTOL
        if ora64 is not None and widen:
            tol = max(GRAD_RTOL, 3.0 * float(np.abs(ref32 - ref).max() / (scale + 1e-30)))
        if scale < 1e-12:              # softmax-shift-invariant biases: the true gradient is 0, all sides hold noise
            assert np.abs(hg[k]).max() < 1e-6, k
            continue
        worst[k] = (float(np.abs(hg[k] - ref).max() / (scale + 1e-30)), tol)
        np.testing.assert_allclose(hg[k], ref, atol=tol * scale + 1e-7, err_msg=k)
    for k, ref in op_.items():
        # one Adam step moves an element by at most lr = 4e-4; elements whose gradient is rounding noise may step
        # the other way (Adam normalises by |g|), everything else lands on the oracle's value
        assert np.abs(hp[k] - ref).max() <= 2 * 4e-4 * 1.01, k
        big = np.abs(og[k]) > 1e-3 * np.abs(og[k]).max()
        if big.any() and np.abs(og[k]).max() > 1e-6:        # (a gradient of ~1e-9 is rounding noise on every side)
            np.testing.assert_allclose(hp[k][big], ref[big], atol=2e-5, err_msg=k)
    return worst


def test_xe_train_iteration_b1024_v10k_vs_oracle_autograd():
    """BASELINE configs[3], one GPU's whole global batch: B=1024 + 80 seq2seq rows, V=10k, T=20, R=36x2048."""
    V, st, B = 10000, synth.DEFAULT_SETTINGS, 1024
    w = synth.make_weights(V, st, seed=0)
    d = synth.make_inputs(B, V, st, regions=36, seq_len=20, seed=1024)
    s2s = synth.make_inputs(80, V, st, regions=36, seq_len=20, seed=1025)
    hip = _hip_iteration(w, V, st, d, s2s)
    ora = _oracle_iteration(w, V, d, s2s, chunk=256)
    ora64 = _oracle_iteration(w, V, d, s2s, chunk=256, dtype=torch.float64)
    # the widened bound exists for ONE reason - a ReLU pre-activation that fp32 and fp64 put on different sides of zero -
    # so it is granted only when the oracle's own arithmetic shows such flips on these inputs (counted here)
    flips = _relu_sign_flips(w, d)
    worst = _compare(hip, ora, ora64=ora64, widen=sum(flips.values()) > 0)
    loose = {k: v for k, v in worst.items() if v[0] > GRAD_RTOL}
    assert set(loose) <= {'att2att.0.weight', 'att2att.0.bias'}, loose     # every other tensor holds the plain 1e-4
    assert not loose or sum(flips.values()) > 0, (loose, flips)


def test_xe_train_iteration_at_196_regions_vs_oracle_autograd():
    """The reference encoder's 14x14 grid (R=196) at the real hidden sizes (A=512: six region chunks in the post-sweep
    dP kernel), few rows and a small vocabulary so the oracle stays quick."""
    V, st, B = 300, synth.DEFAULT_SETTINGS, 5
    w = synth.make_weights(V, st, seed=4)
    d = synth.make_inputs(B, V, st, regions=196, seq_len=20, seed=196)
    s2s = synth.make_inputs(3, V, st, regions=196, seq_len=20, seed=197)
    _compare(_hip_iteration(w, V, st, d, s2s), _oracle_iteration(w, V, d, s2s))


def test_post_sweep_dp_dv_kernels_odd_shapes_vs_fp64():
    """isc_attn_dv_from_alpha / isc_attn_dp_from_de alone on shapes the per-step kernel does not take: column counts
    that do not divide the workgroup (A=96), more than one column block (A=1040), T*R beyond one LDS image."""
    from insenticap_model_amd import ops
    g = torch.Generator().manual_seed(5)
    for B, T, R, A in ((3, 4, 50, 96), (2, 3, 7, 1040), (2, 40, 500, 64)):
        P = torch.randn(B, R, A, generator=g)
        w = torch.randn(1, A, generator=g) * 0.3
        q = torch.randn(T, B, A, generator=g)
        q2 = torch.randn(B, A, generator=g)
        de = torch.randn(T, B, R, generator=g)
        alpha = torch.softmax(torch.randn(B, T, R, generator=g), dim=-1)
        dout = torch.randn(T, B, A, generator=g)
        dV, dP = torch.empty(B, R, A, device=DEV), torch.empty(B, R, A, device=DEV)
        ops.attn_dv_from_alpha(alpha.to(DEV), dout.to(DEV), dV)
        ops.attn_dp_from_de(P.to(DEV), q.to(DEV), w.to(DEV), de.to(DEV), dP, q2=q2.to(DEV))
        torch.cuda.synchronize()
        ref = torch.einsum('btr,tbd->brd', alpha.double(), dout.double())
        np.testing.assert_allclose(dV.cpu().numpy(), ref.float().numpy(), atol=3e-5 * max(1.0, T / 20))
        th = torch.tanh(P.double().unsqueeze(0) + (q.double() + q2.double()).unsqueeze(2))
        refP = (de.double().unsqueeze(-1) * w.double().view(1, 1, 1, A) * (1 - th * th)).sum(0)
        np.testing.assert_allclose(dP.cpu().numpy(), refP.float().numpy(), atol=5e-5 * max(1.0, T / 20))


def test_three_iterations_above_8192_rows_split_f16_engine_vs_exact_fp32_engine():
    """Above 8192 unrolled rows with a vocabulary that is not a multiple of 32 the backward contracts d-logits with a
    zero-padded COPY of the classifier.  A weights scope remembers (pointer, planes) of every W operand and the optimizer
    re-splits them all after its step, from that pointer: the copy was once made inside the scope, i.e. a dead
    temporary was re-read after every optimizer step (a GPU fault once the allocator has returned the block to the
    driver - torch.cuda.graph's capture does that) and its planes could be found again under a recycled address.
    The copy now stays outside the scope; this test pins what must hold either way: third-iteration gradients of the
    split-f16 engine equal those of the exact-fp32 engine (which has no planes) to the engines' usual distance, not
    to the ~1e-2 that one-step-stale classifier weights would give at this step size."""
    from insenticap_model_amd import ops
    from insenticap_model_amd.train import xe_train_step
    V, st, B, T = 300, synth.DEFAULT_SETTINGS, 512, 16
    assert (V + 31) // 32 * 32 != V and B * T >= 8192
    w = synth.make_weights(V, st, seed=2)
    d = synth.make_inputs(B, V, st, regions=36, seq_len=T, seed=77)
    tt = lambda a: torch.from_numpy(np.asarray(a)).to(DEV)
    fact = (None, tt(d['fc_feats']), tt(d['att_feats']), (tt(d['captions']), d['lengths']), tt(d['cpt_words']))
    labels = tt(d['senti_labels'])

    def third_iteration_grads(mode):
        prev = ops.set_h3_mode(mode)
        try:
            cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
            cap.load_state_dict({k: torch.from_numpy(x) for k, x in w.items()})
            cap.to(DEV).eval()
            optim, xc, dc = cap.get_optim_criterion(4e-3)          # a large step: stale weights would show
            for _ in range(3):
                xe_train_step(cap, optim, xc, dc, fact, labels, None, 0.0, 0.0)
            torch.cuda.synchronize()
            return {k: q.grad.detach().cpu().numpy().copy() for k, q in cap.named_parameters() if q.grad is not None}
        finally:
            ops.set_h3_mode(prev)
    fast, exact = third_iteration_grads(1), third_iteration_grads(0)
    for k in ('lang_lstm.weight_ih', 'lang_lstm.weight_hh', 'att_lstm.weight_ih', 'classifier.weight'):
        scale = np.abs(exact[k]).max()
        err = float(np.abs(fast[k] - exact[k]).max() / scale)
        assert err < 2e-3, (k, err)        # (two steps of Adam amplify the engines' 1e-5..1e-4 per-step differences)
