"""Decode rows (csrc/rows.hip, isc_rows_step_fwd / isc_beam_select): the inference decode step on at most 8 rows.

Kernel by kernel against fp64 (classifier statistics, logits, per-tile masked candidates on ragged vocabularies and row
counts 1..8; the whole step against the general kernels of isc_step_fwd on the same plan, with and without a state
re-ordering index), the one-launch top-k + merge against isc_beam_topk + isc_beam_merge on the same logits, and end to
end: beam searches (one image x beam 5 / 3, two images x beam 4, tiny and reference sizes) and a 4-caption greedy
roll-out on this path against the same calls on the kernels it replaces.  pytest -m gpu."""
import numpy as np
import pytest
import torch

from insenticap_model_amd import Captioner, _lib, ops, synth

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _rand(g, *shape, scale=1.0):
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _n():
    return ops._lib.load().isc_rows_launches()


class _region_walk_scan:
    """The general step with its gated scan on attn_scan_gate_kernel (isc_set_rows_scan_max(0)): "the kernels this path
    replaces" of the comparisons below; the default hands steps of up to 256 rows to the rows scan kernel as well."""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = ops.set_rows_scan_max(0) if self.on else None

    def __exit__(self, *exc):
        if self.on:
            ops.set_rows_scan_max(self.prev)
        return False


def _ext(V, beam=0, last=None, cons=0, special=1, cand=None):
    x = _lib.RowsExt()
    x.stats_tile, x.beam = ops.rows_stats_tile(V), beam
    x.pad_id, x.sos_id, x.unk_id, x.mask_special, x.decoding_constraint = 0, 1, 3, special, cons
    if last is not None:
        x.last_word = last.data_ptr()
    if cand is not None:
        x.cand_val, x.cand_idx = cand[0].data_ptr(), cand[1].data_ptr()
    return x


@pytest.mark.parametrize('M', [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize('V,K', [(10000, 512), (9487, 512), (130, 64), (64, 32), (4100, 96)])
def test_classifier_statistics_logits_and_tile_candidates(M, V, K):
    g = torch.Generator().manual_seed(V + 7 * M)
    h, W, bias = _rand(g, M, K), _rand(g, V, K, scale=4 * K ** -0.5), _rand(g, V)
    ref = h.double() @ W.double().t() + bias.double()
    tw = ops.rows_stats_tile(V)
    nt = (V + tw - 1) // tw
    pm, ps = torch.empty(M, nt, device=DEV), torch.empty(M, nt, device=DEV)
    pi = torch.empty(M, nt, device=DEV, dtype=torch.int32)
    lg = torch.full((M, V), float('nan'), device=DEV)
    cv = torch.full((M, nt, 8), float('nan'), device=DEV)
    ci = torch.full((M, nt, 8), -7, device=DEV, dtype=torch.int32)
    last = torch.randint(4, V, (M,), generator=g).to(DEV)
    x = _ext(V, beam=5, last=last, cons=1, cand=(cv, ci))
    n0 = _n()
    ops.rows_vocab_fwd(h.to(DEV), W.to(DEV), bias.to(DEV), pm, ps, pi, x, lg)
    torch.cuda.synchronize()
    assert _n() == n0 + 1
    np.testing.assert_allclose(lg.cpu().numpy(), ref.float().numpy(), atol=3e-6, rtol=2e-6)
    mx = pm.max(1).values
    lse = mx + torch.log((ps * torch.exp(pm - mx[:, None])).sum(1))
    np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(ref, 1).float().numpy(), atol=5e-6, rtol=2e-6)
    # statistics and candidates are exactly those of the stored logits
    pad = torch.full((M, nt * tw), float('-inf'))
    pad[:, :V] = lg.cpu()
    t = pad.view(M, nt, tw)
    assert torch.equal(pm.cpu(), t.max(2).values)
    assert torch.equal(pi.cpu().long(), t.argmax(2) + torch.arange(nt)[None, :] * tw)
    masked = pad.clone()
    masked[:, [0, 1, 3]] = float('-inf')
    masked[torch.arange(M), last.cpu()] = float('-inf')
    mt = masked.view(M, nt, tw)
    top = mt.topk(8, dim=2)
    assert torch.equal(cv.cpu(), top.values)
    finite = torch.isfinite(top.values)
    # ids: exact wherever the value is unique inside its tile
    vals = top.values
    uniq = finite.clone()
    uniq[:, :, 1:] &= vals[:, :, 1:] != vals[:, :, :-1]
    uniq[:, :, :-1] &= vals[:, :, :-1] != vals[:, :, 1:]
    assert torch.equal(ci.cpu().long()[uniq], (top.indices + torch.arange(nt)[None, :, None] * tw)[uniq])
    # repeatable
    cv2, ci2 = torch.empty_like(cv), torch.empty_like(ci)
    ops.rows_vocab_fwd(h.to(DEV), W.to(DEV), bias.to(DEV), pm, ps, pi, _ext(V, 5, last, 1, cand=(cv2, ci2)), lg)
    torch.cuda.synchronize()
    assert torch.equal(cv, cv2) and torch.equal(ci, ci2)


def _captioner(V=10000, st=synth.DEFAULT_SETTINGS, seed=0):
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=seed).items()})
    return cap.to(DEV).eval()


def _inputs(B, V, st, R=36, T=20, seed=11):
    d = synth.make_inputs(B, V, st, regions=R, seq_len=T, seed=seed)
    return {k: torch.from_numpy(np.asarray(v)).to(DEV) for k, v in d.items()}


@pytest.mark.parametrize('rows,V,st,R', [(5, 10000, synth.DEFAULT_SETTINGS, 36), (3, 10000, synth.DEFAULT_SETTINGS, 36),
                                         (8, 10000, synth.DEFAULT_SETTINGS, 36), (5, 10000, synth.DEFAULT_SETTINGS, 196),
                                         (1, 10000, synth.DEFAULT_SETTINGS, 6), (5, 256, synth.TINY_SETTINGS, 12),
                                         (2, 64, synth.TINY_SETTINGS, 1)])
def test_whole_step_against_the_general_kernels(rows, V, st, R):
    """isc_rows_step_fwd vs isc_step_fwd on the same plan (random non-zero state); then with a re-ordering index
    against the general step on explicitly gathered state."""
    cap = _captioner(V, st)
    d = _inputs(rows, V, st, R)
    p = cap._p()
    H = st['rnn_hid_dim']
    with torch.no_grad(), ops.h3_weights_scope(DEV):
        P = cap._prologue(p, 'rl', d['fc_feats'], d['att_feats'], d['cpt_words'], d['senti_words'], d['senti_labels'],
                          want_table='build', words_table=True, gate_rows=rows)
        assert cap._rows_step_ok(rows, P)
        g = torch.Generator().manual_seed(rows)
        h0 = (_rand(g, 2, rows, H, scale=0.5).to(DEV), )
        c0 = (_rand(g, 2, rows, H, scale=0.5).to(DEV), )
        tok = torch.randint(4, V, (rows,), generator=g).to(DEV)
        perm = torch.randint(0, rows, (rows,), generator=g).to(DEV)
        outs = {}
        for name, src in (('general', None), ('rows', None), ('general_perm', perm), ('rows_perm', perm)):
            rows_path = name.startswith('rows')
            tw = ops.rows_stats_tile(V) if rows_path else 128
            ws = cap._alloc_step_ws(rows, P, tw)
            hc, cc = h0[0], c0[0]
            if src is not None and not rows_path:
                hc, cc = hc[:, src].contiguous(), cc[:, src].contiguous()
            hn, cn = torch.empty_like(hc), torch.empty_like(cc)
            aC, aS, bG = torch.zeros(rows, R, device=DEV), torch.zeros(rows, P.Mw, device=DEV), torch.zeros(rows, 1, device=DEV)
            x = None
            if rows_path:
                x = _ext(V)
                if src is not None:
                    x.src_row = src.data_ptr()
            n0 = _n()
            with _region_walk_scan(not rows_path):
                cap._step(p, P, ws, None, hc, cc, hn, cn, aC, aS, bG, tok=tok, rows_ext=x)
            torch.cuda.synchronize()
            assert _n() - n0 == (5 if rows_path else 0)
            mx = ws['pmax'].max(1).values
            lse = mx + torch.log((ws['psum'] * torch.exp(ws['pmax'] - mx[:, None])).sum(1))
            arg = ws['pidx'].gather(1, ws['pmax'].argmax(1)[:, None]).squeeze(1)
            outs[name] = [t.cpu() for t in (hn, cn, ws['f'], aC, aS, bG, lse, mx, arg)]
    for a, b in (('general', 'rows'), ('general_perm', 'rows_perm')):
        for i, (x, y) in enumerate(zip(outs[a][:8], outs[b][:8])):
            np.testing.assert_allclose(y.numpy(), x.numpy(), atol=2e-5, rtol=2e-5, err_msg='%s output %d' % (b, i))
        same = outs[a][8] == outs[b][8]
        assert same.float().mean() >= 0.5 or rows < 3


def test_select_equals_topk_then_merge():
    """isc_beam_select on the few-row classifier's outputs vs isc_beam_topk + isc_beam_merge on the same logits: same
    ids everywhere, log-probs and scores to fp32 rounding of the normaliser, same parents, words, lengths, counters."""
    g = torch.Generator().manual_seed(3)
    n_img, beam, T, V, K = 2, 4, 6, 10000, 512
    rows = n_img * beam
    W, bias = _rand(g, V, K, scale=4 * K ** -0.5).to(DEV), _rand(g, V).to(DEV)
    tw = ops.rows_stats_tile(V)
    nt, nt128 = (V + tw - 1) // tw, (V + 127) // 128
    eos = 2
    for t in (0, 1, 2):
        h = _rand(g, rows, K).to(DEV)
        last = torch.randint(4, V, (rows,), generator=g)
        if t == 2:
            last[1] = eos                                   # an ended candidate is carried
            last[4:8] = eos                                 # image 1: all ended -> done latches
        last = last.to(DEV)
        score = (_rand(g, rows).double() * 3).to(DEV)
        words = torch.randint(4, V, (rows, T), generator=g).to(DEV)
        length = torch.full((rows,), t, dtype=torch.int32, device=DEV)
        # --- rows path
        pm, ps = torch.empty(rows, nt, device=DEV), torch.empty(rows, nt, device=DEV)
        pi = torch.empty(rows, nt, device=DEV, dtype=torch.int32)
        lg = torch.empty(rows, V, device=DEV)
        cv, ci = torch.empty(rows, nt, 8, device=DEV), torch.empty(rows, nt, 8, device=DEV, dtype=torch.int32)
        ops.rows_vocab_fwd(h, W, bias, pm, ps, pi, _ext(V, beam, last, 1, cand=(cv, ci)), lg)
        a = _lib.BeamSelectArgs()
        a.n_img, a.beam, a.T, a.t, a.n_tile, a.V, a.eos_id = n_img, beam, T, t, nt, V, eos
        out = dict(score=torch.zeros(rows, dtype=torch.float64, device=DEV), last=torch.zeros(rows, dtype=torch.int64, device=DEV),
                   words=torch.zeros(rows, T, dtype=torch.int64, device=DEV), length=torch.zeros(rows, dtype=torch.int32, device=DEV),
                   done=torch.zeros(n_img, dtype=torch.int32, device=DEV), src=torch.zeros(rows, dtype=torch.int64, device=DEV),
                   live=torch.zeros(T + 1, dtype=torch.int32, device=DEV), tv=torch.zeros(rows, beam, device=DEV),
                   ti=torch.zeros(rows, beam, dtype=torch.int64, device=DEV))
        a.part_max, a.part_sum, a.cand_val, a.cand_idx = pm.data_ptr(), ps.data_ptr(), cv.data_ptr(), ci.data_ptr()
        a.score_in, a.score_out, a.last_in, a.last_out = score.data_ptr(), out['score'].data_ptr(), last.data_ptr(), out['last'].data_ptr()
        a.words_in, a.words_out, a.len_in, a.len_out = words.data_ptr(), out['words'].data_ptr(), length.data_ptr(), out['length'].data_ptr()
        a.done, a.src_row, a.live = out['done'].data_ptr(), out['src'].data_ptr(), out['live'].data_ptr()
        a.top_val, a.top_idx = out['tv'].data_ptr(), out['ti'].data_ptr()
        ops.beam_select(a)
        # --- three-launch path on the same logits
        pm2, ps2 = torch.empty(rows, nt128, device=DEV), torch.empty(rows, nt128, device=DEV)
        t3 = torch.full((rows, nt128 * 128), float('-inf'), device=DEV)
        t3[:, :V] = lg
        t3 = t3.view(rows, nt128, 128)
        pm2.copy_(t3.max(2).values)
        ps2.copy_(torch.exp(t3 - pm2[:, :, None]).sum(2))
        tv2, ti2 = torch.empty(rows, beam, device=DEV), torch.empty(rows, beam, dtype=torch.int64, device=DEV)
        ops.beam_topk(lg, pm2, ps2, last, beam, 0, 1, 3, True, 1, tv2, ti2)
        m = _lib.BeamMergeArgs()
        ref = dict(score=torch.zeros(rows, dtype=torch.float64, device=DEV), last=torch.zeros(rows, dtype=torch.int64, device=DEV),
                   words=torch.zeros(rows, T, dtype=torch.int64, device=DEV), length=torch.zeros(rows, dtype=torch.int32, device=DEV),
                   done=torch.zeros(n_img, dtype=torch.int32, device=DEV), gather=torch.zeros(rows, dtype=torch.int64, device=DEV),
                   live=torch.zeros(T + 1, dtype=torch.int32, device=DEV))
        m.n_img, m.beam, m.T, m.t, m.eos_id = n_img, beam, T, t, eos
        m.top_val, m.top_idx = tv2.data_ptr(), ti2.data_ptr()
        m.score_in, m.score_out, m.last_in, m.last_out = score.data_ptr(), ref['score'].data_ptr(), last.data_ptr(), ref['last'].data_ptr()
        m.words_in, m.words_out, m.len_in, m.len_out = words.data_ptr(), ref['words'].data_ptr(), length.data_ptr(), ref['length'].data_ptr()
        m.done, m.gather, m.live = ref['done'].data_ptr(), ref['gather'].data_ptr(), ref['live'].data_ptr()
        ops.beam_merge(m)
        torch.cuda.synchronize()
        assert torch.equal(out['ti'], ti2), t
        np.testing.assert_allclose(out['tv'].cpu().numpy(), tv2.cpu().numpy(), atol=2e-6)
        assert torch.equal(out['last'], ref['last']) and torch.equal(out['words'], ref['words'])
        assert torch.equal(out['length'], ref['length']) and torch.equal(out['done'], ref['done'])
        assert torch.equal(out['live'], ref['live'])
        assert torch.equal(out['src'], ref['gather'] % rows)
        np.testing.assert_allclose(out['score'].cpu().numpy(), ref['score'].cpu().numpy(), atol=1e-5)


@pytest.mark.parametrize('V,st,n_img,beam,R', [(10000, synth.DEFAULT_SETTINGS, 1, 5, 36), (10000, synth.DEFAULT_SETTINGS, 1, 3, 36),
                                               (10000, synth.DEFAULT_SETTINGS, 2, 4, 36), (10000, synth.DEFAULT_SETTINGS, 1, 5, 196),
                                               (256, synth.TINY_SETTINGS, 1, 5, 12), (64, synth.TINY_SETTINGS, 2, 3, 5),
                                               # (beam 6 .. 8: the select's lists move up past their fifth place; 1, 2: its smallest workgroups)
                                               (10000, synth.DEFAULT_SETTINGS, 1, 8, 36), (10000, synth.DEFAULT_SETTINGS, 1, 6, 36),
                                               (256, synth.TINY_SETTINGS, 1, 7, 12), (256, synth.TINY_SETTINGS, 1, 1, 12),
                                               (256, synth.TINY_SETTINGS, 4, 2, 12)])
def test_beam_search_on_this_path_equals_the_general_path(V, st, n_img, beam, R):
    cap = _captioner(V, st, seed=2)
    cap.enable_beam_graphs(False)
    d = _inputs(n_img, V, st, R, seed=5)
    res = {}
    for on in (True, False):
        cap.rows_step = on
        n0 = _n()
        with _region_walk_scan(not on):
            out = cap.sample_batch(d['fc_feats'], d['att_feats'], d['senti_words'], d['senti_labels'], beam, 1, 20)
        torch.cuda.synchronize()
        launched = _n() - n0
        # (the live-image counter is read every fourth step: a search that ends early has enqueued up to three more)
        assert (launched % 5 == 0 and launched >= 5 * cap.last_beam_steps) if on else launched == 0
        res[on] = out
    (_, s1, i1), (_, s0, i0) = res[True], res[False]
    for a, b, sa, sb in zip(i1, i0, s1, s0):
        if a != b:          # a near-tie between engines may reorder candidates: then the scores must tie too
            assert abs(sa[0] - sb[0]) < 1e-3, (a, b, sa, sb)
        else:
            np.testing.assert_allclose(sa, sb, atol=2e-4)
    assert sum(a == b for a, b in zip(i1, i0)) >= (len(i1) + 1) // 2


def test_a_vocabulary_beyond_the_tile_lists_takes_the_general_kernels():
    """Round-4 advisor finding: the few-row classifier keeps statistics per <= 64 columns and isc_beam_select holds 256 tile
    lists, so V > 16384 does not fit this path - and the path was chosen without looking at V: a single-image beam search
    raised ISC_E_SHAPE instead of running on the general kernels.  Now the gate (Captioner._rows_vocab_ok,
    isc_rows_step_supported) sends such a vocabulary to the general kernels: same captions as with the path switched off,
    no rows launch, also for a small greedy roll-out and from the default beam graphs."""
    V, st = 20000, synth.DEFAULT_SETTINGS
    cap = _captioner(V, st, seed=4)
    assert not cap._rows_vocab_ok() and _captioner(16384 - 384, synth.TINY_SETTINGS)._rows_vocab_ok()
    d = _inputs(2, V, st, 36, seed=6)
    n0 = _n()
    outs = []
    with _region_walk_scan(True):                      # (the general step's own use of the row scan kernel off: it counts too)
        for graphs in (False, True, True):
            cap.enable_beam_graphs(graphs)
            outs.append(cap.sample(d['fc_feats'][0], d['att_feats'][0], d['senti_words'][0], d['senti_labels'][0:1], 5, 1, 20))
        with torch.no_grad():
            seq = cap(d['fc_feats'], d['att_feats'], d['cpt_words'], d['senti_words'], d['senti_labels'], 20, 1, mode='rl')[0]
        torch.cuda.synchronize()
    assert _n() == n0                                  # nothing went to the few-row kernels
    cap.rows_step = False
    cap.enable_beam_graphs(False)
    ref = cap.sample(d['fc_feats'][0], d['att_feats'][0], d['senti_words'][0], d['senti_labels'][0:1], 5, 1, 20)
    with torch.no_grad():
        seq0 = cap(d['fc_feats'], d['att_feats'], d['cpt_words'], d['senti_words'], d['senti_labels'], 20, 1, mode='rl')[0]
    for o in outs:
        assert o[0] == ref[0]
        np.testing.assert_allclose(o[1], ref[1], atol=1e-5)
    assert torch.equal(seq, seq0)
    plan = cap._make_plan(cap._p(), cap._prologue(cap._p(), 'beam', d['fc_feats'][:1], d['att_feats'][:1], None,
                                                 d['senti_words'][:1], d['senti_labels'][:1], want_table='build',
                                                 gate_rows=5), 5)
    assert not ops.rows_step_supported(plan)           # the library says so too


def test_beam_graphs_replay_equals_eager_on_this_path():
    cap = _captioner(10000, synth.DEFAULT_SETTINGS, seed=4)
    d = _inputs(1, 10000, synth.DEFAULT_SETTINGS, 36, seed=6)
    cap.enable_beam_graphs(False)
    ref = cap.sample_batch(d['fc_feats'], d['att_feats'], d['senti_words'], d['senti_labels'], 5, 1, 20)
    cap.enable_beam_graphs(True)
    for _ in range(3):      # eager (first sight), capture, replay
        out = cap.sample_batch(d['fc_feats'], d['att_feats'], d['senti_words'], d['senti_labels'], 5, 1, 20)
        assert out[2] == ref[2]
        np.testing.assert_allclose(out[1], ref[1], atol=1e-6)


@pytest.mark.parametrize('B', [1, 2, 3, 4])
def test_finalize_folded_into_the_next_steps_first_launch_equals_the_finalize_launches(B):
    """Few-row greedy roll-outs: step t's isc_rollout_finalize rides on step t + 1's att-LSTM launch
    (isc_rows_ext.fin_prev; cap.rows_fused_finalize = False keeps one finalize launch per step).  Same fold, same
    arithmetic: tokens, log-probs, masks, raw tokens and the executed-step counters are bit-identical - also when rows
    end early (an <EOS> placed where a row emits it) and when every row has ended (the reference's early break)."""
    cap = _captioner(10000, synth.DEFAULT_SETTINGS, seed=3)
    cap.enable_rollout_graphs(False)
    big = _inputs(160, 10000, synth.DEFAULT_SETTINGS, 36, seed=18)
    with torch.no_grad():               # (160 x 20 token-steps: the token table is built, and cached for the small calls)
        cap(big['fc_feats'], big['att_feats'], big['cpt_words'], big['senti_words'], big['senti_labels'], 20, 1, mode='rl')
    d = _inputs(B, 10000, synth.DEFAULT_SETTINGS, 36, seed=17)
    args = (d['fc_feats'], d['att_feats'], d['cpt_words'], d['senti_words'], d['senti_labels'], 20, 1)
    fin = ops._lib.load().isc_rollout_finalize_launches

    def run(fused):
        cap.rows_fused_finalize = fused
        n0, f0 = _n(), fin()
        with torch.no_grad():
            seq, lp, mk = cap(*args, mode='rl')
        torch.cuda.synchronize()
        assert _n() - n0 == 5 * 20
        assert fin() - f0 == (1 if fused else 20)
        return seq.cpu(), lp.cpu(), mk.cpu(), cap.cont_weights.cpu(), cap.senti_weights.cpu()
    eos0 = cap.eos_id
    try:
        base = run(False)
        cases = [eos0, int(base[0][0, 2]), int(base[0][B - 1, 7])]      # no early end / row 0 ends at step 2 / a later one
        for eos in cases:
            cap.eos_id = eos
            a, b = run(True), run(False)
            for x, y in zip(a, b):
                assert torch.equal(x, y), eos
            if eos == cases[1]:         # row 0 ends where it first emits that token (at step 2 at the latest)
                first = int((base[0][0] == eos).nonzero()[0])
                assert first <= 2 and float(a[2][0, first + 1:].sum()) == 0.0 and float(a[2][0, :first + 1].sum()) == first + 1
    finally:
        cap.eos_id = eos0
        cap.rows_fused_finalize = True


def test_few_row_beam_graph_follows_in_place_changes_of_the_prologue_weights():
    """A few-row search's graph keeps the f16 planes of its prologue's weights (built once in front of the capture, not
    re-split by every replay): a weight of the prologue changed in place must not be served from the old planes - the
    graph key holds those weights' versions, the next calls run eagerly and capture anew."""
    cap = _captioner(10000, synth.DEFAULT_SETTINGS, seed=4)
    d = _inputs(1, 10000, synth.DEFAULT_SETTINGS, 36, seed=6)
    args = (d['fc_feats'], d['att_feats'], d['senti_words'], d['senti_labels'], 5, 1, 20)
    for _ in range(3):                  # eager, capture, replay
        before = cap.sample_batch(*args)
    h3 = ops._lib.load().isc_h3s_launches
    n0 = h3()
    cap.sample_batch(*args)
    split_free = h3() - n0              # (a replay enqueues through the graph: the library's counters stand still)
    assert split_free == 0
    with torch.no_grad():
        for q in (cap.att_embed[0].weight, cap.att2att[0].weight, cap.attention.cont2att.weight):
            q.mul_(-0.75)               # in place: the version counters move
    cap.enable_beam_graphs(False)
    ref = cap.sample_batch(*args)
    cap.enable_beam_graphs(True)
    assert ref[2] != before[2] or not np.allclose(ref[1], before[1], atol=1e-3)     # the change matters
    for _ in range(3):
        out = cap.sample_batch(*args)
        assert out[2] == ref[2]
        np.testing.assert_allclose(out[1], ref[1], atol=1e-6)


def test_copy_multi_moves_every_byte():
    """isc_copy_multi: up to 8 copies of mixed sizes / dtypes / alignments in one launch (graph replays stage their inputs
    with it); more pairs go out in groups."""
    g = torch.Generator().manual_seed(3)
    shapes = [(2048,), (36, 2048), (10,), (1,), (3, 5, 7), (16385,), (1, 1), (4097, 3), (129,), (65536 + 3,)]
    dts = [torch.float32, torch.float32, torch.int64, torch.int64, torch.float16, torch.uint8, torch.int32, torch.float32,
           torch.uint8, torch.uint8]
    srcs, dsts = [], []
    for sh, dt in zip(shapes, dts):
        x = (torch.rand(*sh, generator=g) * 200 - 100)
        srcs.append(x.to(dt).to(DEV))
        dsts.append(torch.zeros(sh, dtype=dt, device=DEV))
    # an unaligned pair: views one byte into their buffers
    raw_s, raw_d = torch.arange(0, 5001, dtype=torch.int64).to(torch.uint8).to(DEV), torch.zeros(5001, dtype=torch.uint8, device=DEV)
    srcs.append(raw_s[1:]); dsts.append(raw_d[1:])
    ops.copy_multi(dsts, srcs)
    torch.cuda.synchronize()
    for a, b in zip(dsts, srcs):
        assert torch.equal(a, b)
    assert int(raw_d[0]) == 0
    with pytest.raises(ValueError):
        ops.copy_multi([dsts[0]], [srcs[1]])


@pytest.mark.parametrize('B', [1, 4, 8])
def test_small_greedy_rollout_on_this_path_equals_the_general_path(B):
    cap = _captioner(10000, synth.DEFAULT_SETTINGS, seed=1)
    cap.enable_rollout_graphs(False)
    d = _inputs(B, 10000, synth.DEFAULT_SETTINGS, 36, seed=9)
    res = {}
    for on in (True, False):
        cap.rows_step = on
        n0 = _n()
        with torch.no_grad(), _region_walk_scan(not on):
            seq, lp, mk = cap(d['fc_feats'], d['att_feats'], d['cpt_words'], d['senti_words'], d['senti_labels'], 20, 1, mode='rl')
        torch.cuda.synchronize()
        assert ((_n() - n0) == 5 * 20) if on else (_n() == n0)
        res[on] = (seq.cpu(), lp.cpu(), mk.cpu(), cap.cont_weights.cpu())
    a, b = res[True], res[False]
    same = (a[0] == b[0]).all(1)
    assert same.float().mean() >= 0.75
    assert float((a[1][same] - b[1][same]).abs().max()) < 1e-4
    assert torch.equal(a[2][same], b[2][same])
    np.testing.assert_allclose(a[3][same].numpy(), b[3][same].numpy(), atol=1e-5)
    # replayed draws (forced tokens) read the stored logits on this path as well
    with torch.no_grad():
        cap.rows_step = True
        rep = cap.forward_rl(d['fc_feats'], d['att_feats'], d['cpt_words'], d['senti_words'], d['senti_labels'], 20, 0,
                             _replay=res[False][0].to(DEV))
        cap.rows_step = False
        rep0 = cap.forward_rl(d['fc_feats'], d['att_feats'], d['cpt_words'], d['senti_words'], d['senti_labels'], 20, 0,
                              _replay=res[False][0].to(DEV))
    assert torch.equal(rep[0], rep0[0])
    np.testing.assert_allclose(rep[1].cpu().numpy(), rep0[1].cpu().numpy(), atol=1e-4)


@pytest.mark.parametrize('B,R', [(9, 36), (128, 36), (200, 36), (256, 36), (64, 6), (100, 196), (12, 49)])
def test_gated_scan_of_larger_steps_on_the_row_kernel_equals_the_region_walk(B, R):
    """isc_step_fwd's gated scan for up to isc_set_rows_scan_max rows runs on rows_scan_gate_kernel (one 1024-thread
    workgroup per row, f and its f16 planes written for the MFMA lang-LSTM): the same greedy roll-out with it and on
    attn_scan_gate_kernel - tokens, log-probs, attention weights."""
    cap = _captioner(10000, synth.DEFAULT_SETTINGS, seed=2)
    cap.enable_rollout_graphs(False)
    d = _inputs(B, 10000, synth.DEFAULT_SETTINGS, R, seed=13)
    res = {}
    prev = ops.set_rows_scan_max(-1)
    assert prev == 256
    try:
        for limit in (256, 0):
            ops.set_rows_scan_max(limit)
            n0 = _n()
            with torch.no_grad():
                seq, lp, mk = cap(d['fc_feats'], d['att_feats'], d['cpt_words'], d['senti_words'], d['senti_labels'], 20, 1,
                                  mode='rl')
            torch.cuda.synchronize()
            assert (_n() - n0 == 20) if limit else (_n() == n0)
            res[limit] = (seq.cpu(), lp.cpu(), mk.cpu(), cap.cont_weights.cpu(), cap.senti_weights.cpu())
    finally:
        ops.set_rows_scan_max(prev)
    a, b = res[256], res[0]
    same = (a[0] == b[0]).all(1)
    assert same.float().mean() >= 0.9
    assert float((a[1][same] - b[1][same]).abs().max()) < 1e-4
    assert torch.equal(a[2][same], b[2][same])
    np.testing.assert_allclose(a[3][same].numpy(), b[3][same].numpy(), atol=1e-5)
    np.testing.assert_allclose(a[4][same].numpy(), b[4][same].numpy(), atol=1e-5)
