"""Backward / training parity of the HIP path vs the reference's autograd results (golden vectors)
and vs fp64 for the individual kernels. Run on the MI355X box: pytest -m gpu."""
import numpy as np
import pytest
import torch

from conftest import assert_digest_close, case_setup, digest
from insenticap_model_amd import Captioner, XECriterion, clip_gradient, ops, synth

pytestmark = pytest.mark.gpu
GRAD_RTOL = 1e-4   # gradients within 1e-4 relative to the tensor's max


def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def T(d, k):
    return torch.from_numpy(np.asarray(d[k])).to(dev())


def make_captioner(name, train=False):
    c, st, w, d, s2s = case_setup(name)
    cap = Captioner(synth.make_idx2word(c['V']), synth.SENTIMENT_CATEGORIES, st)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    cap.to(dev())
    cap.train(train)
    return cap, c, st, w, d, s2s


@pytest.mark.parametrize('M,N,K', [(6, 32, 128), (128, 512, 2048), (300, 1024, 2048), (2560, 512, 10000),
                                   (77, 64, 9487)])
def test_gemm_nn_vs_fp64(M, N, K):
    g = torch.Generator().manual_seed(M + N)
    Kp = (K + 31) // 32 * 32
    a = torch.zeros(M, Kp)
    a[:, :K] = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / K ** 0.5
    ref = a[:, :K].double() @ w.double()
    out = torch.empty(M, N, device=dev())
    ops.gemm_bwd([ops.gemm_problem([(a.to(dev()), w.to(dev()))], out, ops.NN)], ops.NN)
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), atol=3e-5, rtol=1e-5)
    # accumulate
    ops.gemm_bwd([ops.gemm_problem([(a.to(dev()), w.to(dev()))], out, ops.NN, accumulate=True)], ops.NN)
    np.testing.assert_allclose(out.cpu().numpy(), 2 * ref.float().numpy(), atol=6e-5, rtol=1e-5)


@pytest.mark.parametrize('rows,M,N', [(48, 128, 32), (2560, 2048, 1536), (1000, 10000, 512), (6, 2048, 512),
                                      (333, 512, 512)])
def test_gemm_tn_vs_fp64(rows, M, N):
    g = torch.Generator().manual_seed(rows + M)
    a = torch.randn(rows, M, generator=g)
    x = torch.randn(rows, N, generator=g)
    ref = a.double().t() @ x.double()
    big = torch.zeros(M, N + 64, device=dev())
    out = big[:, 32:32 + N]            # column slice of a wider gradient tensor
    ops.gemm_bwd([ops.gemm_problem([(a.to(dev()), x.to(dev()))], out, ops.TN)], ops.TN)
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), atol=2e-4 * max(1.0, rows ** 0.5 / 8), rtol=1e-5)
    assert float(big[:, :32].abs().max()) == 0.0 and float(big[:, 32 + N:].abs().max()) == 0.0


def test_clamp_adam_vs_torch():
    g = torch.Generator().manual_seed(3)
    shapes = [(1000, 37), (5,), (64, 64), (1,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    gs = [torch.randn(s, generator=g) * 0.3 for s in shapes]
    ref_p = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.Adam(ref_p, lr=4e-4)
    mine = [torch.nn.Parameter(p.clone().to(dev())) for p in ps]
    from insenticap_model_amd import FusedClampAdam
    mopt = FusedClampAdam(mine, lr=4e-4)
    for it in range(3):
        for q, gg in zip(ref_p, gs):
            q.grad = (gg * (it + 1)).clone().clamp_(-0.1, 0.1)
        opt.step()
        for q, gg in zip(mine, gs):
            q.grad = (gg * (it + 1)).clone().to(dev())
        clip_gradient(mopt, 0.1)
        mopt.step()
    for a, b in zip(ref_p, mine):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), atol=2e-6)
    assert set(mopt.state_dict()['state'][0].keys()) == set(opt.state_dict()['state'][0].keys())


def run_iteration(cap, d, s2s):
    xe_crit, da_crit = XECriterion(), torch.nn.MSELoss()
    cap.zero_grad()
    pred = cap(T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'cpt_words'), T(d, 'captions'), T(d, 'senti_labels'),
               0.0, mode='xe')
    xe = xe_crit(pred, T(d, 'captions')[:, 1:], d['lengths'])
    da = da_crit(cap.cpt_feats, cap.fc_feats.detach())
    pred2 = cap(T(s2s, 'captions'), T(s2s, 'cpt_words'), T(s2s, 'senti_words'), T(s2s, 'senti_labels'), 0.0,
                mode='seq2seq')
    l2 = xe_crit(pred2, T(s2s, 'captions')[:, 1:], s2s['lengths'])
    (xe + da + l2).backward()
    return pred, pred2, (float(xe.detach()), float(da.detach()), float(l2.detach()))


def test_tiny_train_iteration_full_grads_and_adam(golden):
    """train_xe.py inner step (xe + domain-align + seq2seq losses, backward, clamp, Adam): every
    gradient tensor and every post-step parameter vs the reference's."""
    g = golden('tiny')
    cap, c, st, w, d, s2s = make_captioner('tiny')
    optim, _, _ = cap.get_optim_criterion(4e-4)
    pred, pred2, losses = run_iteration(cap, d, s2s)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g['it/xe_logp'], atol=1e-4)
    np.testing.assert_allclose(pred2.detach().cpu().numpy(), g['it/s2s_logp'], atol=1e-4)
    np.testing.assert_allclose(losses, g['it/losses'], rtol=2e-5)
    grads = {}
    n = 0
    for k, q in cap.named_parameters():
        key = 'it/grad/' + k
        if key not in g.files:
            assert q.grad is None, k
            continue
        ref = g[key]
        grads[k] = q.grad.cpu().numpy().copy()
        np.testing.assert_allclose(grads[k], ref, atol=GRAD_RTOL * np.abs(ref).max() + 1e-7, err_msg=k)
        n += 1
    assert n == 32
    clip_gradient(optim, 0.1)
    optim.step()
    for k, q in cap.named_parameters():
        got, ref = q.detach().cpu().numpy(), g['it/adam/' + k]
        if k not in grads:
            np.testing.assert_array_equal(got, ref, err_msg=k)
            continue
        big = np.abs(g['it/grad/' + k]) > 1e-4      # Adam's first step amplifies noise on ~eps-sized grads
        np.testing.assert_allclose(got[big], ref[big], atol=3e-6, err_msg=k)
        np.testing.assert_allclose(got[~big], ref[~big], atol=8.2e-4, err_msg=k)   # sign flips: up to 2*lr


def test_tiny_dropout_train_mode_grads(golden):
    g = golden('tiny')
    cap, c, st, w, d, s2s = make_captioner('tiny', train=True)
    masks = {k: torch.from_numpy(g['drop/mask_' + k]) for k in ('fc', 'att', 'label')}
    for i in range(c['T']):
        masks['out%d' % i] = torch.from_numpy(g['drop/mask_out%d' % i])
    cap.zero_grad()
    pred = cap.forward_xe(T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'cpt_words'), T(d, 'captions'),
                          T(d, 'senti_labels'), 0.0, _masks=masks)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g['drop/xe_logp'], atol=2e-4)
    loss = XECriterion()(pred, T(d, 'captions')[:, 1:], d['lengths'])
    np.testing.assert_allclose(float(loss.detach()), g["drop/loss"][0], rtol=2e-5)
    loss.backward()
    for k, q in cap.named_parameters():
        key = 'drop/grad/' + k
        if key in g.files:
            ref = g[key]
            np.testing.assert_allclose(q.grad.cpu().numpy(), ref, atol=GRAD_RTOL * np.abs(ref).max() + 1e-7,
                                       err_msg=k)


def test_tiny_sampled_rollout_reinforce_grads(golden):
    """forward_rl(sample_max=0) with grad: replay the reference's draws, RewardCriterion, backward."""
    g = golden('tiny')
    cap, c, st, w, d, _ = make_captioner('tiny')
    a = (T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'cpt_words'), T(d, 'senti_words'), T(d, 'senti_labels'))
    cap.zero_grad()
    seq, lp, mk = cap.forward_rl(*a, c['T'], 0, _replay=torch.from_numpy(g['rl/sample_draws']).to(dev()))
    assert lp.requires_grad
    assert (seq.cpu().numpy() == g['rl/sample_seq']).all()
    np.testing.assert_allclose(lp.detach().cpu().numpy(), g['rl/sample_logprobs'], atol=1e-4)
    reward = torch.from_numpy(g['rl/sample_reward']).to(dev())
    loss = (-lp * mk * reward).sum() / mk.sum()       # RewardCriterion, self_critical/utils.py:173-177
    np.testing.assert_allclose(float(loss.detach()), g["rl/sample_rl_loss"][0], rtol=5e-5)
    loss.backward()
    n = 0
    for k, q in cap.named_parameters():
        key = 'rl/grad/' + k
        if key in g.files:
            ref = g[key]
            np.testing.assert_allclose(q.grad.cpu().numpy(), ref, atol=GRAD_RTOL * np.abs(ref).max() + 1e-7,
                                       err_msg=k)
            n += 1
    assert n >= 36


@pytest.mark.parametrize('name', ['cfg1', 'b128'])
def test_fullsize_train_iteration_digests(golden, name):
    """BASELINE.json configs[1] (B=128 XE forward+backward, 36x2048 feats, V=10k, T=20) and the B=4
    plumbing config: losses, gradient fingerprints of all 32 trained tensors, post-Adam parameters."""
    g = golden(name)
    cap, c, st, w, d, s2s = make_captioner(name)
    optim, _, _ = cap.get_optim_criterion(4e-4)
    pred, pred2, losses = run_iteration(cap, d, s2s)
    np.testing.assert_allclose(losses, g['it/losses'], rtol=3e-5)
    tgt = pred.detach().gather(2, T(d, 'captions')[:, 1:].unsqueeze(2)).squeeze(2).cpu().numpy()
    np.testing.assert_allclose(tgt, g['it/xe_logp_tgt'], atol=1e-4)
    n = 0
    for k, q in cap.named_parameters():
        key = 'it/gdig/' + k
        if key in g.files:
            assert_digest_close(digest(q.grad.cpu().numpy()), g[key], k, rel=2e-4)
            n += 1
    assert n == 32
    clip_gradient(optim, 0.1)
    optim.step()
    for k, q in cap.named_parameters():
        got, ref = digest(q.detach().cpu().numpy()), g['it/adig/' + k]
        # parameters move by at most lr=4e-4 per element; the fingerprint must agree to that scale.
        # Tensors whose true gradient is 0 (the alpha biases: softmax shift invariance) carry only
        # ~1e-9 rounding noise in the reference, which Adam's first step amplifies to ~lr.
        gkey = 'it/gdig/' + k
        if gkey in g.files and g[gkey][2] > 1e-6:
            assert abs(got[2] - ref[2]) <= 5e-5 * ref[2] + 1e-6, k    # noise-gradient elements may step the other way
        np.testing.assert_allclose(got[3:], ref[3:], atol=4.1e-4, err_msg=k)


def test_dv_and_dp_summed_after_the_sweep_equal_the_per_step_accumulation():
    """isc_attn_dv_from_alpha / isc_attn_dp_from_de (dV = sum_t alpha_t x dout_t and dP = sum_t d e_t w (1 - tanh^2) once,
    after the sweep) against isc_attn_scan_bwd accumulating dV and dP at every step in the sweep's order: bit-identical;
    the other outputs of the scan backward do not change when the two are left out of it.  With and without q2."""
    D_ = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    # (R = 196: the reference encoder's 14 x 14 grid - several region chunks; odd column counts are in
    # tests/test_gpu_train_sizes.py: the per-step kernel does not take them)
    for B, T, R, A, with_q2 in ((37, 7, 36, 512, False), (21, 20, 11, 512, True), (5, 3, 6, 64, False),
                                (9, 20, 196, 512, False), (3, 20, 196, 512, True), (4, 5, 50, 128, True)):
        Pm, Vm = torch.randn(B, R, A, generator=g).to(D_), torch.randn(B, R, A, generator=g).to(D_)
        w = (torch.randn(1, A, generator=g) * 0.3).to(D_)
        q = torch.randn(T, B, A, generator=g).to(D_)
        q2 = torch.randn(B, A, generator=g).to(D_) if with_q2 else None
        alpha = torch.softmax(torch.randn(B, T, R, generator=g), dim=-1).to(D_)
        dout = torch.randn(T, B, A, generator=g).to(D_)
        outs = {}
        for deferred in (False, True):
            dP = None if deferred else torch.empty(B, R, A, device=D_)
            dV = None if deferred else torch.empty(B, R, A, device=D_)
            de = torch.empty(T, B, R, device=D_)
            dq, dw = torch.empty(T, B, A, device=D_), torch.empty(B, A, device=D_)
            for i, t in enumerate(range(T - 1, -1, -1)):
                ops.attn_scan_bwd([ops.scan_bwd_problem(Pm, Vm, q[t], w, alpha[:, t], dout[t], dP, dV, dq[t], dw, i > 0,
                                                        q2=q2, de_out=de[t])], B)
            outs[deferred] = (dq, dw, de, dP, dV)
        dV2, dP2 = torch.empty(B, R, A, device=D_), torch.empty(B, R, A, device=D_)
        ops.attn_dv_from_alpha(alpha, dout, dV2)
        ops.attn_dp_from_de(Pm, q, w, outs[True][2], dP2, q2=q2)
        torch.cuda.synchronize()
        assert torch.equal(outs[False][4], dV2), (B, T, R, A)
        assert torch.equal(outs[False][3], dP2), (B, T, R, A)
        for k in range(3):
            assert torch.equal(outs[False][k], outs[True][k]), k
        ref = torch.einsum('btr,tbd->brd', alpha.double().cpu(), dout.double().cpu())
        np.testing.assert_allclose(dV2.cpu().numpy(), ref.float().numpy(), atol=2e-5)
        qq = q.double().cpu() + (q2.double().cpu() if with_q2 else 0.0)
        th = torch.tanh(Pm.double().cpu().unsqueeze(0) + qq.unsqueeze(2))                 # [T,B,R,A]
        refP = (outs[True][2].double().cpu().unsqueeze(-1) * w.double().cpu().view(1, 1, 1, A) * (1 - th * th)).sum(0)
        np.testing.assert_allclose(dP2.cpu().numpy(), refP.float().numpy(), atol=3e-5)


def test_reward_loss_kernels_vs_the_reference_formula():
    """isc_reward_loss_fwd / _bwd (RewardCriterion, self_critical/utils.py:169-177) against the formula in fp64, and
    through autograd against torch's own graph of the same expression."""
    from insenticap_model_amd import RewardCriterion
    g = torch.Generator().manual_seed(8)
    for B, Tn in ((1, 1), (6, 8), (512, 20), (1000, 33)):
        lp = -torch.rand(B, Tn, generator=g) * 5
        mk = (torch.rand(B, Tn, generator=g) > 0.3).float()
        mk[:, 0] = 1
        rw = torch.randn(B, Tn, generator=g)
        ref = (-lp.double() * mk.double() * rw.double()).sum() / mk.double().sum()
        a = lp.clone().to(dev()).requires_grad_(True)
        loss = RewardCriterion()(a, mk.to(dev()), rw.to(dev()))
        np.testing.assert_allclose(float(loss.detach()), float(ref), rtol=2e-6, atol=1e-7)
        (loss * 3.0).backward()
        b = lp.clone().requires_grad_(True)
        ((-b * mk * rw).sum() / mk.sum() * 3.0).backward()
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-6, atol=1e-9)
        again = RewardCriterion()(a.detach(), mk.to(dev()), rw.to(dev()))
        assert float(again) == float(loss.detach())               # fixed reduction order: bit-repeatable
    # python-number / broadcast rewards (Detector's 'senti' branch hands 0 + 0.4 * cls_reward, utils callers a scalar)
    a = lp.to(dev()).requires_grad_(True)
    np.testing.assert_allclose(float(RewardCriterion()(a, mk.to(dev()), 0.5).detach()),
                               float((-lp.double() * mk.double() * 0.5).sum() / mk.double().sum()), rtol=2e-6)


def test_grad_scale_kernel_picks_the_power_of_two():
    out = torch.zeros(4, device=dev())
    for vals, want in (([5e-5, -1e-9], 2.0 ** 11), ([0.0, 0.0], 1.0), ([3.0], 2.0 ** -5), ([0.124], 1.0),
                       ([0.0624], 2.0), ([float('nan'), 1e-3], 2.0 ** 6), ([float('inf')], 1.0), ([1e-30], 2.0 ** 60)):
        a = torch.tensor(vals, device=dev())
        ops.grad_scale([a[:1].contiguous(), None, a[1:].contiguous()], out)
        S, inv = out[:2].tolist()
        assert out[2:].view(torch.int32).tolist() == [0, 0]          # state words left zeroed
        assert S == want and inv == 1.0 / want, (vals, S, want)
        finite = [abs(v) for v in vals if np.isfinite(v) and v != 0]
        if finite and want not in (1.0, 2.0 ** 60) or vals == [0.124]:
            assert 2.0 ** -4 <= max(finite) * S < 2.0 ** -3
    big = torch.rand(3_000_000, device=dev()) * 1e-6                  # many workgroups: same answer as one
    big[1_234_567] = -7.0e-4
    ops.grad_scale([big, None, big[:5].contiguous()], out)
    assert out[:2].tolist() == [2.0 ** 7, 2.0 ** -7] and out[2:].view(torch.int32).tolist() == [0, 0]


def test_sparse_dlogp_handover_equals_the_dense_tensor_bit_for_bit():
    """XECriterion and the REINFORCE gather hand their gradient to the decode node as (token, weight) pairs; with the
    gradient scale off the result must equal the dense [B,T,V] route bit for bit (same kernel arithmetic), and a loss
    that also touches the log-probs directly (dense part) adds on top."""
    cap, c, st, w, d, s2s = make_captioner('tiny')
    cap.grad_scaling = False
    a = (T(d, 'fc_feats'), T(d, 'att_feats'), T(d, 'cpt_words'), T(d, 'captions'), T(d, 'senti_labels'))

    def grads(kind):
        cap.zero_grad()
        pred = cap(*a, 0.0, mode='xe')
        tgt = T(d, 'captions')[:, 1:]
        if kind == 'sparse':
            loss = XECriterion()(pred, tgt, d['lengths'])
        elif kind == 'dense':                                     # same loss by torch ops: dense d log-prob
            L = torch.tensor(d['lengths'], device=dev())
            mask = (torch.arange(pred.shape[1], device=dev())[None, :] < L[:, None]).float()
            loss = -(pred.gather(2, tgt.unsqueeze(2)).squeeze(2) * mask).sum() / mask.sum()
        else:                                                     # criterion + a dense term on the same log-probs
            loss = XECriterion()(pred, tgt, d['lengths']) + 1e-3 * pred[:, :, 5].sum()
        loss.backward()
        return {k: q.grad.clone() for k, q in cap.named_parameters() if q.grad is not None}
    gs, gd, gm = grads('sparse'), grads('dense'), grads('mixed')
    assert set(gs) == set(gd)
    for k in gs:
        assert torch.equal(gs[k], gd[k]), k
    cap.zero_grad()
    pred = cap(*a, 0.0, mode='xe')
    (1e-3 * pred[:, :, 5].sum()).backward()
    for k, q in cap.named_parameters():
        if q.grad is not None:
            ref = gs[k] + q.grad
            # (two sweeps on differently rounded f16 planes: the project-wide gradient bar, not bit equality)
            np.testing.assert_allclose(gm[k].cpu().numpy(), ref.cpu().numpy(), atol=GRAD_RTOL * float(ref.abs().max()) + 1e-9,
                                       err_msg=k)
    # the scaled sweep agrees with the unscaled one to rounding
    cap.grad_scaling = True
    g2 = grads('sparse')
    for k in gs:
        np.testing.assert_allclose(g2[k].cpu().numpy(), gs[k].cpu().numpy(),
                                   atol=GRAD_RTOL * float(gs[k].abs().max()) + 1e-10, err_msg=k)
