"""The ragged teacher-forced unroll (`Captioner.row_counts`, autograd._train_forward / _backward: step t runs on the rows
whose caption has not ended, the prefix [0, M_t) of a batch sorted by length as the reference's collates sort it,
dataloader.py:17,37,68,124) against the full unroll of the same batch: the three losses and all 32 gradients - the positions
it skips are the ones XECriterion masks (captioner.py:431-436), their gradient is exactly zero.  Tiny geometry and BASELINE
configs[1]'s (128 + 80 rows, V = 10k, T = 20, 36 x 2048), eval and train mode (same dropout draws), unsorted lengths and
device-side lengths leave the unroll as it is.  pytest -m gpu."""
import numpy as np
import pytest
import torch

from insenticap_model_amd import Captioner, ops, synth
from insenticap_model_amd.train import xe_forward_backward

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TINY = dict(V=64, st=synth.TINY_SETTINGS, R=6, T=8, B=8, S=4)
FULL = dict(V=10000, st=synth.DEFAULT_SETTINGS, R=36, T=20, B=128, S=80)
GRAD_TOL = 1e-4          # gradients within 1e-4 of the tensor's largest element


def make(cfg, ragged, seed=9):
    cap = Captioner(synth.make_idx2word(cfg['V']), synth.SENTIMENT_CATEGORIES, cfg['st'])
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(cfg['V'], cfg['st'], seed=seed).items()})
    cap.pair_unrolls = False
    cap.ragged_unroll = ragged
    return cap.to(DEV).eval()


def batch(cfg, seed, sort=True):
    d = synth.make_inputs(cfg['B'], cfg['V'], cfg['st'], regions=cfg['R'], seq_len=cfg['T'], seed=seed)
    s = synth.make_inputs(cfg['S'], cfg['V'], cfg['st'], regions=cfg['R'], seq_len=cfg['T'], seed=seed + 1000)
    if sort:
        d, s = synth.sort_by_length(d), synth.sort_by_length(s)
    t = lambda x: torch.from_numpy(x).to(DEV)      # noqa: E731
    fact = (t(d['fc_feats']), t(d['att_feats']), t(d['captions']), d['lengths'], t(d['cpt_words']))
    scs = (t(s['captions']), s['lengths'], t(s['cpt_words']), t(s['senti_words']), t(s['senti_labels']))
    return fact, t(d['senti_labels']), scs


def iteration(cap, fact, labels, scs, seed=None):
    optim, xc, dc = cap.get_optim_criterion(4e-4)
    if seed is not None:
        torch.manual_seed(seed)
    vec = xe_forward_backward(cap, optim, xc, dc, fact, labels, scs, 0.0, pair=False, overlap_unrolls=False)
    grads = {k: q.grad.detach().clone() for k, q in cap.named_parameters() if q.grad is not None}
    return vec.cpu().numpy(), grads


def assert_close_grads(ga, gb, tol=GRAD_TOL):
    assert ga.keys() == gb.keys() and len(ga) == 32
    for k in ga:
        gmax = float(gb[k].abs().max())
        err = float((ga[k] - gb[k]).abs().max())
        assert torch.isfinite(ga[k]).all(), k
        assert err <= tol * max(gmax, 1e-12), (k, err, gmax)


@pytest.mark.parametrize('cfg', [TINY, FULL], ids=['tiny', 'b128_v10k'])
@pytest.mark.parametrize('train', [False, True], ids=['eval', 'train'])
def test_ragged_unroll_equals_the_full_unroll(cfg, train):
    fact, labels, scs = batch(cfg, 90)
    assert fact[3][0] > fact[3][-1]                      # (ragged for real)
    a_cap, b_cap = make(cfg, True).train(train), make(cfg, False).train(train)
    h3s = ops._lib.load().isc_h3s_launches()
    a = iteration(a_cap, fact, labels, scs, seed=3)
    b = iteration(b_cap, fact, labels, scs, seed=3)
    assert ops._lib.load().isc_h3s_launches() >= h3s
    np.testing.assert_allclose(a[0], b[0], rtol=3e-6, atol=3e-6)
    assert_close_grads(a[1], b[1])
    assert ops.device_status() == 0


def test_ragged_unroll_after_garbage_in_the_allocator():
    """The rows a step skips must not leak whatever the caching allocator hands back: poison freed memory with NaN first."""
    cfg = TINY
    fact, labels, scs = batch(cfg, 91)
    b = iteration(make(cfg, False), fact, labels, scs)
    junk = [torch.full((1 << 20,), float('nan'), device=DEV) for _ in range(8)]
    del junk
    a = iteration(make(cfg, True), fact, labels, scs)
    np.testing.assert_allclose(a[0], b[0], rtol=3e-6, atol=3e-6)
    assert_close_grads(a[1], b[1])


def test_unsorted_or_device_lengths_keep_the_full_unroll():
    cfg = TINY
    cap = make(cfg, True)
    fact, labels, scs = batch(cfg, 92, sort=False)
    assert any(a < b for a, b in zip(fact[3], fact[3][1:]))
    with cap.row_counts(fact[3]):
        assert cap.__dict__['_row_counts'] is None
    with cap.row_counts(torch.tensor(sorted(fact[3], reverse=True), device=DEV)):
        assert cap.__dict__['_row_counts'] is None
    with cap.row_counts(sorted(fact[3], reverse=True)):
        counts = cap.__dict__['_row_counts']
        assert counts[0] == cfg['B'] and counts == sorted(counts, reverse=True) and counts[-1] >= 1
    assert cap.__dict__['_row_counts'] is None
    a = iteration(cap, fact, labels, scs)                # unsorted batch through the flag: the full unroll, same numbers
    b = iteration(make(cfg, False), fact, labels, scs)
    np.testing.assert_array_equal(a[0], b[0])


def test_policy_flag_takes_the_eager_step_off_the_merged_chain_and_auto_waits_for_large_batches():
    cfg = TINY
    fact, labels, scs = batch(cfg, 93)
    cap = make(cfg, True)
    cap.pair_unrolls = None                              # (the default: merged eager steps - unless the ragged form applies)
    assert cap.ragged_applies(fact[3]) and not cap.ragged_applies(torch.tensor(fact[3]))
    calls, orig = [], cap.row_counts
    cap.row_counts = lambda lengths, T=None: (calls.append(len(lengths)), orig(lengths, T))[1]
    optim, xc, dc = cap.get_optim_criterion(4e-4)
    vec = xe_forward_backward(cap, optim, xc, dc, fact, labels, scs, 0.0)
    assert calls == [cfg['B'], cfg['B'], cfg['S']]       # asked once, then one chain per unroll inside the contexts
    ref = iteration(make(cfg, False), fact, labels, scs)
    np.testing.assert_allclose(vec.cpu().numpy(), ref[0], rtol=3e-6, atol=3e-6)
    cap.ragged_unroll = 'auto'                           # 8 rows: far below RAGGED_AUTO_ROWS
    assert not cap.ragged_applies(fact[3])
    cap.RAGGED_AUTO_ROWS = 4                             # (instance override)
    assert cap.ragged_applies(fact[3])
    # scheduled sampling keeps the full unroll - and therefore the merged chain
    cap.ragged_unroll = True
    calls.clear()
    xe_forward_backward(cap.train(), optim, xc, dc, fact, labels, scs, 0.25)
    assert calls == []


@pytest.mark.parametrize('V', [64, 10000, 9999], ids=['v64', 'v10k', 'v_odd'])
def test_dlogits_of_a_position_with_zero_coefficients_is_zero_whatever_its_logits_hold(V):
    """isc_logsoftmax_bwd_raw: a row whose coefficients are all zero (behind its caption's end) is written as zeros
    without its logits or statistics being read - NaN there must not reach d logits; every other row keeps its bits."""
    B, T, n_tile, Vp = 6, 5, (V + 127) // 128, (V + 31) // 32 * 32
    g = torch.Generator().manual_seed(4)
    raw = torch.randn(T, B, V, generator=g).to(DEV)
    pm, ps = torch.empty(T, B, n_tile, device=DEV), torch.empty(T, B, n_tile, device=DEV)
    for k in range(n_tile):                      # tile statistics as the classifier's epilogue leaves them
        tile = raw[:, :, k * 128:(k + 1) * 128]
        pm[:, :, k] = tile.amax(-1)
        ps[:, :, k] = (tile - pm[:, :, k:k + 1]).exp().sum(-1)
    ids = torch.randint(0, V, (B, T), generator=g).to(DEV)
    coef = torch.randn(B, T, generator=g).to(DEV)
    dead = torch.zeros(B, T, dtype=torch.bool, device=DEV)
    dead[1, 3:] = dead[4, 1:] = dead[5, 4:] = True
    coef = torch.where(dead, torch.zeros_like(coef), coef).contiguous()

    def run(r, m, s):
        out = torch.full((T * B, Vp), 7.0, device=DEV)
        ops.logsoftmax_bwd_raw(r, V, B * V, B, T, V, m, s, B, [(ids, coef)], out, out_step_rows=B)
        return out.view(T, B, Vp)
    clean = run(raw, pm, ps)
    dead_tb = dead.t()
    raw2, pm2, ps2 = raw.clone(), pm.clone(), ps.clone()
    raw2[dead_tb], pm2[dead_tb], ps2[dead_tb] = float('nan'), float('inf'), 0.0
    poisoned = run(raw2, pm2, ps2)
    assert torch.equal(poisoned, clean)
    assert (clean[dead_tb] == 0).all() and (clean[~dead_tb][:, :V].abs().sum(-1) > 0).all() and (clean[:, :, V:] == 0).all()
    # the live rows against the definition: coef * (onehot - softmax)
    ref = -torch.softmax(raw.double(), -1) * coef.t().double().unsqueeze(-1)
    ref.scatter_add_(2, ids.t().unsqueeze(-1), coef.t().double().unsqueeze(-1))
    np.testing.assert_allclose(clean[:, :, :V].cpu().numpy(), ref.cpu().numpy(), atol=2e-6)
