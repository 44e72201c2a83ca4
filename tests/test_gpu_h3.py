"""Split-f16 GEMM path (gemm_h3_kernel, include/insenticap_hip.h: isc_set_h3_mode): fp32 operands split into two
f16 planes each, three f16 MFMAs per k-step, fp32 accumulators.  Checked here: it stays an fp32-accurate path - its
error against an fp64 contraction is no larger than that of the fp32 MFMA tiles on the same inputs - through every
fused epilogue (linear, LSTM cell, vocabulary statistics), on ragged shapes, K-segments and grouped launches, and the
auto mode takes it only for large launches."""
import numpy as np
import pytest
import torch

from insenticap_model_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore():
    yield
    ops.set_h3_mode(1)
    ops.set_tile_override(-1)


def dev():
    return torch.device('cuda:0')


def _rand(g, *shape, scale=1.0):
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _err(out, ref):
    d = (out.double().cpu() - ref).abs()
    return d.max().item(), d.pow(2).mean().sqrt().item()


@pytest.mark.parametrize('M,N,K1,K2', [(512, 256, 64, 0), (1000, 520, 512, 96), (777, 128, 2048, 0), (4096, 512, 512, 512)])
def test_linear_h3_vs_fp32_tiles(M, N, K1, K2):
    g = torch.Generator().manual_seed(M + N)
    x1, w1, b = _rand(g, M, K1), _rand(g, N, K1, scale=K1 ** -0.5), _rand(g, N)
    keep = (torch.rand(M, N, generator=g) > 0.5).to(torch.uint8)
    prior = _rand(g, M, N)
    ref = x1.double() @ w1.double().t() + b.double() + prior.double()
    segs = [(x1.to(dev()), w1.to(dev()))]
    if K2:
        x2, w2 = _rand(g, M, K2), _rand(g, N, K2, scale=K2 ** -0.5)
        ref = ref + x2.double() @ w2.double().t()
        segs.append((x2.to(dev()), w2.to(dev())))
    ref_pre = torch.relu(ref)
    ref_out = ref_pre * keep.double() * 2.0
    db, dkeep = b.to(dev()), keep.to(dev())
    errs = {}
    for mode in (0, 2):
        ops.set_h3_mode(mode)
        out = prior.clone().to(dev())
        pre = torch.full((M, N), float('nan'), device=dev())
        ops.linear_fwd([ops.linear_problem(segs, out, db, relu=True, keep_mask=dkeep,
                                           mask_scale=2.0, out_pre=pre, accumulate=True)])
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.cpu().numpy(), ref_out.float().numpy(), atol=3e-5, rtol=1e-5)
        np.testing.assert_allclose(pre.cpu().numpy(), ref_pre.float().numpy(), atol=3e-5, rtol=1e-5)
        errs[mode] = _err(pre, ref_pre)
    # fp32-accurate: no worse than the fp32 MFMA chain (both are dominated by the final fp32 rounding of the output)
    assert errs[2][1] <= errs[0][1] * 1.05 + 1e-9, errs
    assert errs[2][0] <= errs[0][0] * 1.5 + 1e-8, errs


def test_linear_grouped_shared_activations():
    """Three projections of one activation matrix in one launch (the roll-out's h2att / h2word / gate launch): the
    activation planes are built once and shared."""
    g = torch.Generator().manual_seed(11)
    M, K = 1500, 512
    x = _rand(g, M, K)
    ws = [_rand(g, n, K, scale=K ** -0.5) for n in (512, 512, 384)]
    bs = [_rand(g, n) for n in (512, 512, 384)]
    dx = x.to(dev())
    dws, dbs = [w.to(dev()) for w in ws], [b.to(dev()) for b in bs]
    outs = {}
    for mode in (0, 2):
        ops.set_h3_mode(mode)
        o = [torch.empty(M, w.shape[0], device=dev()) for w in ws]
        ops.linear_fwd([ops.linear_problem([(dx, w)], oo, b) for w, oo, b in zip(dws, o, dbs)])
        torch.cuda.synchronize()
        outs[mode] = o
    for w, b, o0, o2 in zip(ws, bs, outs[0], outs[2]):
        ref = x.double() @ w.double().t() + b.double()
        e0, e2 = _err(o0, ref), _err(o2, ref)
        assert e2[0] < 3e-6 and e2[1] <= e0[1] * 1.05 + 1e-9, (e0, e2)


@pytest.mark.parametrize('M,H,with_pre,with_tab', [(700, 64, True, False), (1024, 512, False, False),
                                                   (4096, 512, True, True)])
def test_lstm_h3(M, H, with_pre, with_tab):
    g = torch.Generator().manual_seed(M + H)
    ks = (H, 2 * H, 32)
    xs = [_rand(g, M, k) for k in ks]
    ws = [_rand(g, 4 * H, k, scale=(3 * k) ** -0.5) for k in ks]
    b_ih, b_hh, c0 = _rand(g, 4 * H), _rand(g, 4 * H), _rand(g, M, H)
    pre = _rand(g, M, 4 * H, scale=0.3) if with_pre else None
    z = sum(x.double() @ w.double().t() for x, w in zip(xs, ws)) + b_ih.double() + b_hh.double()
    kw = {}
    keep = []
    if with_pre:
        z = z + pre.double()
        kw['pre'] = pre.to(dev())
    if with_tab:
        tab = _rand(g, 50, 4 * H, scale=0.3)
        ids = torch.randint(0, 50, (M,), generator=g)
        z = z + tab.double()[ids]
        kw['tab'], kw['tab_ids'] = tab.to(dev()), ids.to(dev())
    i, f, gg, o = z.split(H, dim=1)
    c_ref = torch.sigmoid(f) * c0.double() + torch.sigmoid(i) * torch.tanh(gg)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    dsegs = [(x.to(dev()), w.to(dev())) for x, w in zip(xs, ws)]
    dargs = [v.to(dev()) for v in (b_ih, b_hh, c0)]
    errs = {}
    for mode in (0, 2):
        ops.set_h3_mode(mode)
        h, c = torch.empty(M, H, device=dev()), torch.empty(M, H, device=dev())
        gates = torch.empty(M, 4 * H, device=dev())
        ops.lstm_fwd(dsegs, dargs[0], dargs[1], dargs[2], h, c, gates_out=gates, **kw)
        torch.cuda.synchronize()
        np.testing.assert_allclose(h.cpu().numpy(), h_ref.float().numpy(), atol=2e-5)
        np.testing.assert_allclose(c.cpu().numpy(), c_ref.float().numpy(), atol=2e-5)
        errs[mode] = _err(h, h_ref) + _err(c, c_ref)
    # the cell's hardware exp / rcp dominate both; the contraction must not add to it
    assert errs[2][1] <= errs[0][1] * 1.1 + 1e-9 and errs[2][3] <= errs[0][3] * 1.1 + 1e-9, errs


@pytest.mark.parametrize('M,V,K', [(520, 1000, 64), (300, 10000, 512), (4096, 10000, 512), (2100, 9487, 512)])
def test_vocab_h3(M, V, K):
    g = torch.Generator().manual_seed(V + M)
    h, W, bias = _rand(g, M, K), _rand(g, V, K, scale=4 * K ** -0.5), _rand(g, V)
    logits_ref = h.double() @ W.double().t() + bias.double()
    lse_ref = torch.logsumexp(logits_ref, 1)
    nt = (V + 127) // 128
    dh, dW, dbias = h.to(dev()), W.to(dev()), bias.to(dev())
    errs, args = {}, {}
    for mode in (0, 2):
        ops.set_h3_mode(mode)
        pm, ps = torch.empty(M, nt, device=dev()), torch.empty(M, nt, device=dev())
        pi = torch.empty(M, nt, device=dev(), dtype=torch.int32)
        logits = torch.empty(M, V, device=dev())
        ops.vocab_fwd(dh, dW, dbias, pm, ps, pi, logits)
        torch.cuda.synchronize()
        np.testing.assert_allclose(logits.cpu().numpy(), logits_ref.float().numpy(), atol=3e-5, rtol=1e-5)
        mx = pm.max(1).values
        lse = mx + torch.log((ps * torch.exp(pm - mx[:, None])).sum(1))
        np.testing.assert_allclose(lse.cpu().numpy(), lse_ref.float().numpy(), atol=3e-5, rtol=1e-5)
        col = pm.argmax(1)
        arg = pi.gather(1, col[:, None]).squeeze(1).long()
        assert torch.equal(arg, logits.argmax(1))           # statistics consistent with the logits written
        errs[mode] = _err(logits, logits_ref)
        args[mode] = arg.cpu()
    assert errs[2][1] <= errs[0][1] * 1.05 + 1e-9 and errs[2][0] <= errs[0][0] * 1.5 + 1e-8, errs
    # arg-max against fp64: the split path may only differ where fp64 itself has a near-tie
    ref_arg = logits_ref.argmax(1)
    for mode in (0, 2):
        bad = (args[mode] != ref_arg).nonzero().flatten()
        for r in bad.tolist():
            top2 = logits_ref[r].topk(2).values
            assert (top2[0] - top2[1]).item() < 1e-5, (mode, r, top2)


def test_h3_wide_dynamic_range_and_tiny_values():
    """Values far below the f16 normal range keep their fp32 accuracy (the lo plane carries the residual scaled by
    2^11), and large-but-legal magnitudes do not overflow."""
    g = torch.Generator().manual_seed(3)
    M, N, K = 2048, 1024, 256          # large enough not to take the split-K route
    x = _rand(g, M, K) * torch.pow(10.0, torch.randint(-7, 4, (M, K), generator=g).float())
    w = _rand(g, N, K) * torch.pow(10.0, torch.randint(-7, 1, (N, K), generator=g).float())
    ref = x.double() @ w.double().t()
    scale = (x.double().abs() @ w.double().abs().t())         # error is relative to sum |a||b|
    dx, dw = x.to(dev()), w.to(dev())
    rel = {}
    for mode in (0, 2):
        ops.set_h3_mode(mode)
        out = torch.empty(M, N, device=dev())
        ops.linear_fwd([ops.linear_problem([(dx, dw)], out)])
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        rel[mode] = ((out.double().cpu() - ref).abs() / scale).max().item()
    assert rel[2] < 2e-6 and rel[2] <= rel[0] * 1.5 + 1e-9, rel


def test_auto_mode_only_takes_large_launches():
    """mode 1 (default): a small launch stays bit-identical to the fp32 tiles, a large one matches the forced path."""
    g = torch.Generator().manual_seed(9)

    def run(M, N, K, mode):
        gg = torch.Generator().manual_seed(M)
        x, w = _rand(gg, M, K).to(dev()), _rand(gg, N, K, scale=K ** -0.5).to(dev())
        ops.set_h3_mode(mode)
        out = torch.empty(M, N, device=dev())
        ops.linear_fwd([ops.linear_problem([(x, w)], out)])
        torch.cuda.synchronize()
        return out.cpu()
    assert torch.equal(run(256, 512, 128, 1), run(256, 512, 128, 0))
    big_auto, big_force, big_off = run(4096, 1024, 256, 1), run(4096, 1024, 256, 2), run(4096, 1024, 256, 0)
    assert torch.equal(big_auto, big_force)
    assert not torch.equal(big_auto, big_off)                  # a different summation order, same accuracy class
    assert (big_auto - big_off).abs().max().item() < 5e-6


def test_linear_with_more_activation_rows_than_any_workspace():
    """40000 x 1024 activations (their planes would exceed the 128 MB workspace): activations without producer planes
    are read as fp32 rows and split in registers after the fragment read - one launch, no plane copy, whatever the
    row count - with accumulate / keep-mask / pre-activation outputs in play."""
    g = torch.Generator().manual_seed(21)
    M, N, K = 40000, 256, 1024
    x, w, b = _rand(g, M, K), _rand(g, N, K, scale=K ** -0.5), _rand(g, N)
    keep = (torch.rand(M, N, generator=g) > 0.3).to(torch.uint8)
    prior = _rand(g, M, N)
    ref_pre = torch.relu(x.double() @ w.double().t() + b.double() + prior.double())
    ref_out = ref_pre * keep.double() * 1.5
    dx, dw, db, dkeep = x.to(dev()), w.to(dev()), b.to(dev()), keep.to(dev())
    before = ops._lib.load().isc_h3_launches()
    out = prior.clone().to(dev())
    pre = torch.full((M, N), float('nan'), device=dev())
    ops.linear_fwd([ops.linear_problem([(dx, dw)], out, db, relu=True, keep_mask=dkeep, mask_scale=1.5,
                                       out_pre=pre, accumulate=True)])
    torch.cuda.synchronize()
    assert ops._lib.load().isc_h3_launches() - before == 1         # one launch: nothing is chunked or copied
    np.testing.assert_allclose(pre.cpu().numpy(), ref_pre.float().numpy(), atol=3e-5, rtol=1e-5)
    np.testing.assert_allclose(out.cpu().numpy(), ref_out.float().numpy(), atol=3e-5, rtol=1e-5)


def test_weights_scope_reuses_planes_and_ends_cleanly():
    """Inside ops.h3_weights_scope the weight planes are built once and reused (bit-identical results, and only the
    first launch carries the weight split); after the scope a changed weight is picked up again."""
    g = torch.Generator().manual_seed(31)
    M, N, K = 4096, 1024, 256
    x, w = _rand(g, M, K).to(dev()), _rand(g, N, K, scale=K ** -0.5).to(dev())

    def run():
        out = torch.empty(M, N, device=dev())
        ops.linear_fwd([ops.linear_problem([(x, w)], out)])
        return out
    plain = run()
    with ops.h3_weights_scope(dev()):
        a, b = run(), run()
        with ops.h3_weights_scope(dev()):          # nested: ignored
            c = run()
    torch.cuda.synchronize()
    assert torch.equal(plain, a) and torch.equal(a, b) and torch.equal(a, c)
    w.mul_(2.0)                                     # outside any scope: the next launch splits the new values
    doubled = run()
    with ops.h3_weights_scope(dev()):
        doubled_scoped = run()
    torch.cuda.synchronize()
    assert torch.equal(doubled_scoped, doubled)
    np.testing.assert_allclose(doubled.cpu().numpy(), (plain * 2).cpu().numpy(), rtol=2e-6, atol=1e-6)


def test_rollout_state_planes_are_bit_identical_to_resplitting():
    """The roll-out lets the LSTM epilogues write the f16 planes of h next to h (isc_step_plan.h1_hi ...); consumers
    then read those instead of splitting h again.  Same values by construction -> identical tokens and log-probs,
    on the whole-step entry point and on the per-op path (bench.py's kernel timer), with fewer split launches."""
    import numpy as np
    from conftest import case_setup
    from insenticap_model_amd import Captioner, synth
    c, st, w, _, _ = case_setup('cfg1')
    cap = Captioner(synth.make_idx2word(c['V']), synth.SENTIMENT_CATEGORIES, st)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    cap.to(dev()).eval()
    B, Tn = 1024, 12
    d = synth.make_inputs(B, c['V'], st, regions=36, seq_len=Tn, seed=99)
    a = [torch.from_numpy(np.asarray(d[k])).to(dev()) for k in ('fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')]
    ops.set_h3_mode(2)
    outs = {}
    for planes in (True, False):
        cap.state_planes = planes
        with torch.no_grad():
            seq, lp, mk = cap(*a, Tn, 1, mode='rl')
        outs[planes] = (seq.cpu(), lp.cpu())
    cap.state_planes = True
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    # per-op path with the timer armed on a middle step: the planes stay consistent across the two paths
    ops.TIMER.arm_step = 5
    try:
        with torch.no_grad():
            seq, lp, mk = cap(*a, Tn, 1, mode='rl')
    finally:
        ops.TIMER.arm_step = None
        ops.TIMER.records.clear()
    assert torch.equal(seq.cpu(), outs[True][0]) and torch.equal(lp.cpu(), outs[True][1])


def test_sentiment_word_tables_match_per_caption_features():
    """Eval-mode roll-out and beam search serve the sentiment words from two vocabulary-sized tables through the
    scan's gather mode (isc_scan_problem.row_ids); the reference's per-caption [B,M,.] features give the same result."""
    import numpy as np
    from conftest import case_setup
    from insenticap_model_amd import Captioner, synth
    c, st, w, _, _ = case_setup('cfg1')
    cap = Captioner(synth.make_idx2word(c['V']), synth.SENTIMENT_CATEGORIES, st)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    cap.to(dev()).eval()
    B, Tn = 320, 12
    d = synth.make_inputs(B, c['V'], st, regions=36, seq_len=Tn, seed=17)
    a = [torch.from_numpy(np.asarray(d[k])).to(dev()) for k in ('fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')]
    outs = {}
    for tables in (True, False):
        cap.words_table = tables
        with torch.no_grad():
            seq, lp, mk = cap(*a, Tn, 1, mode='rl')
            sw = cap.senti_weights.cpu()
            caps, scores, _ = cap.sample_batch(a[0][:8], a[1][:8], a[3][:8], a[4][:8], 3, 1, Tn)
        outs[tables] = (seq.cpu(), lp.cpu(), sw, caps, scores)
    cap.words_table = True
    assert torch.equal(outs[True][0], outs[False][0])
    np.testing.assert_allclose(outs[True][1].numpy(), outs[False][1].numpy(), atol=1e-5)
    np.testing.assert_allclose(outs[True][2].numpy(), outs[False][2].numpy(), atol=1e-5)
    assert outs[True][3] == outs[False][3]
    np.testing.assert_allclose(np.asarray(outs[True][4]), np.asarray(outs[False][4]), atol=1e-4)


def test_freed_captioners_planes_are_never_resumed_by_the_next_model():
    """Round-2 advisor finding: the suspended weights scope is process-global per stream and was keyed only on
    (data_ptr, version) of the parameters.  A checkpoint-evaluation loop frees one Captioner and builds the next with
    the same shapes: the caching allocator hands back the same addresses with the same version counters, and the
    second model would decode with the FIRST model's f16 planes.  The key now carries a per-instance nonce (renewed by
    .to() and load_state_dict), and a freed instance's suspended scopes are forgotten."""
    import gc
    from insenticap_model_amd import Captioner, synth
    V, st = 512, synth.DEFAULT_SETTINGS
    d = synth.make_inputs(256, V, st, regions=8, seq_len=6, seed=3)
    a = [torch.from_numpy(np.asarray(d[k])).to(dev()) for k in ('fc_feats', 'att_feats', 'cpt_words', 'senti_words',
                                                                 'senti_labels')]

    def decode(seed, mode):
        ops.set_h3_mode(mode)
        cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
        cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=seed).items()})
        cap.to(dev()).eval()
        with torch.no_grad():
            seq, lp, _ = cap(*a, 6, 1, mode='rl')
            seq2, lp2, _ = cap(*a, 6, 1, mode='rl')          # second call: resumes ITS OWN suspended scope
        assert torch.equal(lp, lp2)
        ptrs = sorted(q.data_ptr() for q in cap.parameters())
        key = cap._weights_key()
        out = (seq.cpu(), lp.cpu())
        del cap
        gc.collect()
        return out, ptrs, key

    (s1, l1), p1, k1 = decode(1, 2)
    (s2, l2), p2, k2 = decode(2, 2)             # same shapes, other weights, (very likely) the same addresses
    (s2x, l2x), _, _ = decode(2, 0)             # the exact-fp32 engine never uses planes
    assert k1[0] != k2[0]                       # distinct nonces whatever the allocator did
    assert not torch.equal(l1, l2)
    np.testing.assert_allclose(l2.numpy(), l2x.numpy(), atol=2e-5)
    if p1 == p2:                                # the hazard was real on this run: same storages, same versions
        assert k1[1:] == k2[1:]
    # load_state_dict and .to() renew the nonce of a live instance
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
    n0 = cap._weights_key()[0]
    cap.load_state_dict(cap.state_dict())
    n1 = cap._weights_key()[0]
    cap.to(dev())
    assert len({n0, n1, cap._weights_key()[0]}) == 3


def test_weights_scope_gives_its_depth_back_when_begin_fails():
    """A scope whose isc_h3_weights_begin raises must not leave the stream's nesting depth at 1: every later scope on
    the stream would otherwise be taken for a nested one and silently never open."""
    import ctypes
    from insenticap_model_amd import _lib
    lib = _lib.load()
    real = lib.isc_h3_weights_begin
    key = (0, torch.cuda.current_stream(0).cuda_stream)
    before = ops.h3_weights_scope._depth.get(key, 0)

    class Boom:
        def __call__(self, *a):
            return -1                          # ISC_E_NULL
    try:
        lib.isc_h3_weights_begin = Boom()
        with pytest.raises(_lib.HipLibraryError):
            with ops.h3_weights_scope(dev()):
                pass
    finally:
        lib.isc_h3_weights_begin = real
    assert ops.h3_weights_scope._depth.get(key, 0) == before
    with ops.h3_weights_scope(dev()) as sc:     # and the next scope opens normally
        assert sc.opened


def test_optimizer_step_refreshes_the_suspended_weight_planes():
    """FusedClampAdam.step() re-splits the weights behind the stream's suspended weights scope in place and in few
    batched launches (isc_h3_weights_refresh), so the next sweep RESUMES its scope: its results equal those of a scope
    built from scratch on the updated weights bit for bit, and the split-launch count of the second iteration drops."""
    from insenticap_model_amd import Captioner, XECriterion, synth
    from insenticap_model_amd.train import xe_train_step
    V, st = 512, synth.DEFAULT_SETTINGS
    d = synth.make_inputs(32, V, st, regions=8, seq_len=6, seed=3)
    s2 = synth.make_inputs(16, V, st, regions=8, seq_len=6, seed=4)
    t = torch.from_numpy
    fact = (None, t(d['fc_feats']), t(d['att_feats']), (t(d['captions']), d['lengths']), t(d['cpt_words']))
    scs = ((t(s2['captions']), s2['lengths']), t(s2['cpt_words']), t(s2['senti_words']), t(s2['senti_labels']))

    def run(refresh):
        cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
        cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=7).items()})
        cap.to(dev()).eval()
        optim, xe_crit, da_crit = cap.get_optim_criterion(4e-4)
        optim.refresh_weight_planes = refresh
        outs = []
        for _ in range(3):
            o = xe_train_step(cap, optim, xe_crit, da_crit, fact, t(d['senti_labels']), scs, 0.0, 0.1,
                              overlap_unrolls=False)
            outs.append(float(o['all_loss']))
        torch.cuda.synchronize()
        return outs, {k: v.detach().clone() for k, v in cap.state_dict().items()}
    ops.set_h3_mode(2)
    l1, p1 = run(True)
    l0, p0 = run(False)
    assert l1 == l0
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def _set_ksplit(on):
    from insenticap_model_amd import _lib
    return _lib.load().isc_set_h3_ksplit(int(on))


@pytest.mark.parametrize('M,N,K1,K2', [(4608, 512, 2048, 0), (4160, 512, 4096, 2048), (1100, 384, 8192, 0)])
def test_linear_long_k_on_few_large_tiles_is_cut_into_k_slices(M, N, K1, K2):
    """isc_set_h3_ksplit: a linear launch of < 200 large tiles over >= 64 k-blocks contracts in slices of K (raw partial
    tiles in slabs, fixed-order reduce + epilogue): same result as the one-slice launch up to fp32 summation order, the
    fp64 bound of the other split-f16 tests, every epilogue feature (bias, accumulate, ReLU, mask, pre-mask output)."""
    g = torch.Generator().manual_seed(M + K1)
    x1, w1, b = _rand(g, M, K1), _rand(g, N, K1, scale=K1 ** -0.5), _rand(g, N)
    keep = (torch.rand(M, N, generator=g) > 0.5).to(torch.uint8)
    prior = _rand(g, M, N)
    ref = x1.double() @ w1.double().t() + b.double() + prior.double()
    segs = [(x1.to(dev()), w1.to(dev()))]
    if K2:
        x2, w2 = _rand(g, M, K2), _rand(g, N, K2, scale=K2 ** -0.5)
        ref = ref + x2.double() @ w2.double().t()
        segs.append((x2.to(dev()), w2.to(dev())))
    ref_pre = torch.relu(ref)
    ref_out = ref_pre * keep.double() * 2.0
    db, dkeep = b.to(dev()), keep.to(dev())
    ops.set_h3_mode(2)
    res = {}
    try:
        for on in (1, 0):
            _set_ksplit(2 * on)              # (2: forward launches too - by default only backward dX launches are cut)
            out = prior.clone().to(dev())
            pre = torch.full((M, N), float('nan'), device=dev())
            ops.linear_fwd([ops.linear_problem(segs, out, db, relu=True, keep_mask=dkeep, mask_scale=2.0, out_pre=pre,
                                               accumulate=True)])
            torch.cuda.synchronize()
            np.testing.assert_allclose(out.cpu().numpy(), ref_out.float().numpy(), atol=3e-5, rtol=1e-5)
            np.testing.assert_allclose(pre.cpu().numpy(), ref_pre.float().numpy(), atol=3e-5, rtol=1e-5)
            res[on] = (pre.clone(), _err(pre, ref_pre))
    finally:
        _set_ksplit(1)
    assert res[1][1][1] <= res[0][1][1] * 1.05 + 1e-9, (res[1][1], res[0][1])
    np.testing.assert_allclose(res[1][0].cpu().numpy(), res[0][0].cpu().numpy(), atol=8e-6, rtol=2e-6)   # fp32 summation order
    assert ops.device_status() == 0


def test_nn_backward_contraction_over_the_vocabulary_in_k_slices():
    """dX = dY W with K = 9984 vocabulary rows and 4160 = 20 x (128 + 80) time-stacked rows (the classifier's input
    gradient of a merged training iteration): K slices on the large tile vs the one-slice launch vs fp64."""
    g = torch.Generator().manual_seed(77)
    M, N, K = 4160, 512, 9984
    dy, w = _rand(g, M, K, scale=1e-2), _rand(g, K, N, scale=K ** -0.5)
    ref = dy.double() @ w.double()
    ddy, dw = dy.to(dev()), w.to(dev())
    ops.set_h3_mode(2)
    outs = {}
    try:
        for on in (1, 0):
            _set_ksplit(on)
            out = torch.empty(M, N, device=dev())
            ops.gemm_bwd([ops.gemm_problem([(ddy, dw)], out, ops.NN)], ops.NN)
            torch.cuda.synchronize()
            outs[on] = out
            np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), atol=2e-6, rtol=1e-5)
    finally:
        _set_ksplit(1)
    np.testing.assert_allclose(outs[1].cpu().numpy(), outs[0].cpu().numpy(), atol=5e-7, rtol=2e-6)


@pytest.mark.parametrize('K,launches', [(4160, 1), (22080, 3)], ids=['one_pass', 'k_chunks'])
def test_dw_contractions_that_share_their_dy_split_it_once(K, launches):
    """isc_gemm_bwd, TN layout: several dW = dY^T X problems over ONE dY (the weight gradients of the layers that consumed
    the same pre-activation gradient) - the transposing split of dY once, one skinny launch for the group - against the
    same problems issued one by one and against fp64; an `accumulate` problem in the group adds to what is there."""
    g = torch.Generator().manual_seed(5)
    M = 2048               # (K = 22 080 = 20 steps x 1104 rows: the planes exceed the 128 MB workspace - three chunks of K)
    dy = _rand(g, K, M, scale=1e-2).to(dev())
    xs = [_rand(g, K, n).to(dev()) for n in (512, 512, 384)]
    prior = _rand(g, M, 384).to(dev())
    from insenticap_model_amd import _lib
    n0 = _lib.load().isc_h3_launches()
    outs = [torch.empty(M, x.shape[1], device=dev()) for x in xs]
    outs[2].copy_(prior)
    ops.gemm_bwd([ops.gemm_problem([(dy, x)], o, ops.TN, accumulate=(i == 2)) for i, (x, o) in enumerate(zip(xs, outs))],
                 ops.TN)
    torch.cuda.synchronize()
    assert _lib.load().isc_h3_launches() - n0 == launches              # one GEMM launch (per K chunk) for the three problems
    single = [torch.empty(M, x.shape[1], device=dev()) for x in xs]
    single[2].copy_(prior)
    for i, (x, o) in enumerate(zip(xs, single)):
        ops.gemm_bwd([ops.gemm_problem([(dy, x)], o, ops.TN, accumulate=(i == 2))], ops.TN)
    torch.cuda.synchronize()
    for i, (a, b, x) in enumerate(zip(outs, single, xs)):
        ref = dy.double().cpu().t() @ x.double().cpu() + (prior.double().cpu() if i == 2 else 0.0)
        np.testing.assert_allclose(a.cpu().numpy(), ref.float().numpy(), atol=6e-6, rtol=1e-5)
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-6, rtol=2e-6)
