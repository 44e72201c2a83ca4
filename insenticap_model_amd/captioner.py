"""MI355X-native `Captioner`: same constructor, attributes, call modes and 40-tensor
`state_dict` as the reference class (/root/reference/models/captioner.py:120-424), with the
decode arithmetic executed by libinsenticap_hip.so instead of stock torch ops.

The nn.Module containers below exist only to own the parameters under the reference's
names (checkpoint compatibility); their `forward`s are never called.
PyTorch is used for device memory, streams and (in train mode) random numbers only.
"""
import ctypes as C
import itertools
import weakref

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import RolloutStep


class ContentAttention(nn.Module):
    def __init__(self, settings):
        super().__init__()
        self.h2att = nn.Linear(settings['rnn_hid_dim'], settings['att_hid_dim'])
        self.att_alpha = nn.Linear(settings['att_hid_dim'], 1)


class SentiAttention(nn.Module):
    def __init__(self, settings):
        super().__init__()
        self.h2word = nn.Linear(settings['rnn_hid_dim'], settings['att_hid_dim'])
        self.label2word = nn.Linear(settings['word_emb_dim'], settings['att_hid_dim'])
        self.word_alpha = nn.Linear(settings['att_hid_dim'], 1)


class Attention(nn.Module):
    def __init__(self, settings):
        super().__init__()
        self.cont_att = ContentAttention(settings)
        self.senti_att = SentiAttention(settings)
        self.h2att = nn.Linear(settings['rnn_hid_dim'], settings['att_hid_dim'])
        self.cont2att = nn.Linear(settings['feat_emb_dim'], settings['att_hid_dim'])
        self.senti2att = nn.Linear(settings['feat_emb_dim'], settings['att_hid_dim'])
        self.att_alpha = nn.Linear(settings['att_hid_dim'], 1)


class _Pro:
    """Step-invariant tensors of one call (captioner.py:198-214 / 247-261 / 294-315)."""
    fc_e = None      # [B,E]   what the att-LSTM sees (after dropout)
    att_e3 = None    # [B,R,E] embedded regions (after dropout)
    att_p3 = None    # [B,R,A] att2att projection
    words_e3 = None  # [B,M,W] embedded sentiment words
    words_p3 = None  # [B,M,A]
    words_ids = None  # gather mode: [B,M] int64 ids (leading <PAD> included); words_e3 / words_p3 are then the
                      # [V,W] / [V,A] tables relu(Emb) and senti2att(relu(Emb)) shared by all rows
    label_e = None   # [B,W]   sentiment-label embedding (added to every xt)
    label_w = None   # [B,A]   label2word(label_e): step-invariant term of the senti attention
    pre1 = None      # [B,4H]  fc_e W_fc^T + label_e W_x^T + b_ih + b_hh: step-invariant att-LSTM input
    tab = None       # [V,4H]  relu(Emb) W_x^T (inference only): replaces the word-embedding K-segment
    gate_Gc = None   # [B*R,A] att_e through cont2att.weight (few-row inference: isc_attn_scan_gate_fwd)
    gate_Gs = None   # [B*M,A] / [V,A] words_e through senti2att.weight
    B = R = Mw = 0
    # kept for the backward pass: raw inputs, ids and dropout keep-masks
    x_fc = x_att = cmean = cpt = cpt_ids = label_ids = sw_ids = None
    m_fc = m_att = m_cpt = m_label = m_words = None
    sc = 1.0


_INSTANCE_NONCE = itertools.count(1)


def _forget_instance(nonce):
    """A Captioner was freed: its suspended weights scopes (f16 planes keyed on its weights) must never be resumed."""
    try:
        ops.h3_weights_scope.forget(lambda wk: wk and wk[0][0] == 'captioner' and wk[0][1] <= nonce[0])
    except Exception:       # noqa: BLE001 - interpreter shutdown
        pass


class Captioner(nn.Module):
    def __init__(self, idx2word, sentiment_categories, settings):
        super().__init__()
        self.idx2word = idx2word
        self.pad_id = idx2word.index('<PAD>')
        self.unk_id = idx2word.index('<UNK>')
        self.sos_id = idx2word.index('<SOS>') if '<SOS>' in idx2word else self.pad_id
        # the reference guards <EOS> with the presence of '<SOS>' (captioner.py:128)
        self.eos_id = idx2word.index('<EOS>') if '<SOS>' in idx2word else self.pad_id
        self.neu_idx = sentiment_categories.index('neutral')
        self.vocab_size = len(idx2word)
        self.settings = dict(settings)
        W, F, FA = settings['word_emb_dim'], settings['fc_feat_dim'], settings['att_feat_dim']
        E, H, A = settings['feat_emb_dim'], settings['rnn_hid_dim'], settings['att_hid_dim']
        for name, v in (('word_emb_dim', W), ('fc_feat_dim', F), ('att_feat_dim', FA),
                        ('feat_emb_dim', E), ('rnn_hid_dim', H), ('att_hid_dim', A)):
            if v % 32:
                raise ValueError('%s=%d: the HIP kernels need dimensions that are multiples of 32' % (name, v))
        if W != E:
            # the gate mixes v_hat [E] with e_hat [W] elementwise (captioner.py:117 and the
            # "TODO now: word_emb_dim == feat_emb_dim" at :157)
            raise ValueError('word_emb_dim must equal feat_emb_dim')
        self.drop = nn.Dropout(settings['dropout_p'])
        self.word_embed = nn.Sequential(nn.Embedding(self.vocab_size, W, padding_idx=self.pad_id), nn.ReLU())
        self.senti_label_embed = nn.Sequential(nn.Embedding(len(sentiment_categories), W), nn.ReLU())
        self.fc_embed = nn.Sequential(nn.Linear(F, E), nn.ReLU())
        self.cpt2fc = nn.Sequential(nn.Linear(W, E), nn.ReLU())
        self.att_embed = nn.Sequential(nn.Linear(FA, E), nn.ReLU())
        self.att_lstm = nn.LSTMCell(H + E + W, H)
        self.att2att = nn.Sequential(nn.Linear(E, A), nn.ReLU())
        self.senti2att = nn.Sequential(nn.Linear(W, A), nn.ReLU())
        self.attention = Attention(settings)
        self.lang_lstm = nn.LSTMCell(H + E, H)
        self.classifier = nn.Linear(H, self.vocab_size)
        self.fc_feats = self.cpt_feats = self.s2s_cpt_feats = None
        # the XE and seq2seq unrolls of one training iteration through ONE step chain (autograd_pair): None = where it is
        # faster (eager steps: yes; inside HIP graphs: two branches - autograd_pair.use_pair), True / False = always / never
        self.pair_unrolls = None
        self.cont_weights, self.senti_weights, self.cont_senti_weights = [], [], []
        # identity of this instance's weights in the process-wide caches keyed on weight values (ops.h3_weights_scope):
        # a one-element list so that the finalizer below sees renewals
        self._wnonce = [next(_INSTANCE_NONCE)]
        weakref.finalize(self, _forget_instance, self._wnonce)
        # Beam searches are served from captured HIP graphs by default (round 3): a search geometry seen twice is
        # captured on its second call and replayed from then on - bit-identical results, 2.3 instead of 2.6 ms for a
        # single-image beam-5 search.  enable_beam_graphs(False) turns it off (320 MB of graph buffers per captioner).
        self._beam_graphs, self._beam_graphs_max = {}, 4
        # ... and so are eval-mode greedy roll-outs of at most ROLLOUT_GRAPH_MAX_ROWS captions (enable_rollout_graphs):
        # at those sizes a roll-out is ~130 dependent small launches and the host is the slower side (B = 4: single call
        # 1.90 -> 1.40 ms, B = 128: 1.99 -> 1.62; at 512 rows the device is, and the graph's per-replay weight split
        # costs more than it saves: eager).
        self._rollout_graphs, self._rollout_graphs_max = {}, 4

    # ------------------------------------------------------------------ plumbing
    def _p(self):
        p = {k: v.detach() for k, v in self.named_parameters()}
        if not p['classifier.weight'].is_cuda:
            raise _lib.HipLibraryError('Captioner parameters are on the CPU: call .to("cuda") - '
                                       'this implementation has no CPU path')
        ops.register_status_words(p['classifier.weight'].device)      # numerics flags of this device (once)
        return p

    @property
    def _dev(self):
        return self.classifier.weight.device

    def _new(self, *shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self._dev)

    def _zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(shape, dtype=dtype, device=self._dev)

    def _zeros_many(self, *specs):
        """Zero tensors for (shape, dtype) specs carved out of ONE zeroed byte arena (256-byte aligned views): one fill
        launch instead of one per tensor - a roll-out starts from a dozen of them, 4-7 % of a small batch's time."""
        sizes = []
        for shape, dtype in specs:
            n = 1
            for d in shape:
                n *= int(d)
            sizes.append(n * torch.empty(0, dtype=dtype).element_size())
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 255) & ~255
        # (a float fill: the same kernel and rate as torch.zeros of a float tensor, whatever the views' types)
        arena = torch.zeros(max(total, 256) // 4, dtype=torch.float32, device=self._dev).view(torch.uint8)
        return [arena[o:o + n].view(dtype).view(*shape) for (shape, dtype), o, n in zip(specs, offs, sizes)]

    def init_hidden(self, bsz):
        H = self.att_lstm.hidden_size
        return (self._zeros(2, bsz, H), self._zeros(2, bsz, H))

    def _f32(self, x):
        ops.require_device(x)
        return x.contiguous() if x.dtype == torch.float32 else x.float().contiguous()

    def _ids(self, x):
        ops.require_device(x)
        return x.long().contiguous()

    def _mask_source(self, masks):
        """Returns f(key, *shape) -> (uint8 keep-mask or None, scale). `masks`: explicit dict
        (tests replay the reference's masks); otherwise drawn with torch's RNG in train mode."""
        p_drop = self.drop.p
        scale = 1.0 / (1.0 - p_drop) if p_drop < 1.0 else 0.0

        pre = {}

        def draw(shape):
            # keep with probability 1 - p: ONE launch (uniform draw + compare + byte store; torch's generator, so it is
            # graph-safe and replays draw afresh) instead of rand / >= / to(uint8)
            return torch.empty(shape, dtype=torch.uint8, device=self._dev).bernoulli_(1.0 - p_drop)

        def f(key, *shape):
            if masks is not None:
                m = masks.get(key)
                if m is None:
                    return None, 1.0
                return m.to(device=self._dev, dtype=torch.uint8).reshape(shape).contiguous(), scale
            if not self.training or p_drop == 0.0:
                return None, 1.0
            if key in pre:
                return pre[key], scale
            return draw(shape), scale

        def predraw(prefix, n, *shape):
            """Draw the masks `prefix0 .. prefix<n-1>` of one unroll in a single tensor (3 launches instead of 3n)."""
            if masks is not None or not self.training or p_drop == 0.0:
                return
            m = draw((n,) + tuple(shape))
            for i in range(n):
                pre['%s%d' % (prefix, i)] = m[i]
        f.predraw = predraw
        return f

    # ------------------------------------------------------------------ prologue
    GATE_FUSED_MAX_ROWS = 768      # decode rows up to which scans + gate sum + gate mix run as ONE launch (inference)
    ROLLOUT_GRAPH_MAX_ROWS = 256   # greedy eval roll-outs up to this many captions are served from HIP graphs

    def _prologue(self, p, mode, fc=None, att=None, cpt_words=None, senti_words=None, senti_labels=None,
                  masks=None, want_table=False, words_table=False, gate_rows=0, pre1_out=None):
        """want_table: False | 'cached' (use the embedding table only if already built) | 'build'.
        words_table: serve the sentiment words from the vocabulary-sized tables (no dropout on them, no autograd).
        gate_rows: number of decode rows of an INFERENCE call (0: training / not applicable): up to
        GATE_FUSED_MAX_ROWS the scans' features are also carried through the gate's projections here, once, so that
        every decode step runs scans + gate sum + gate mix as one launch (isc_attn_scan_gate_fwd)."""
        P = _Pro()
        st = self.settings
        E, A, Wd = st['feat_emb_dim'], st['att_hid_dim'], st['word_emb_dim']
        mask_for = self._mask_source(masks)
        first = []   # small independent problems that share one launch
        if mode != 'seq2seq':
            fc = self._f32(fc)
            B = fc.shape[0]
            att = self._f32(att).reshape(B, -1, att.shape[-1])
            R = att.shape[1]
            P.B, P.R = B, R
            m, sc = mask_for('fc', B, E)
            P.x_fc, P.m_fc, P.sc = fc, m, sc
            P.fc_e = self._new(B, E)
            self.fc_feats = self._new(B, E) if m is not None else P.fc_e
            first.append(ops.linear_problem([(fc, p['fc_embed.0.weight'])], P.fc_e, p['fc_embed.0.bias'],
                                            relu=True, keep_mask=m, mask_scale=sc,
                                            out_pre=self.fc_feats if m is not None else None))
        else:
            B = cpt_words.shape[0]
            P.B = B
        if cpt_words is not None:
            cmean = self._new(B, Wd)
            P.cpt_ids = self._ids(cpt_words)
            ops.embed_relu_mean_fwd(p['word_embed.0.weight'], P.cpt_ids, cmean)
            cpt = self._new(B, E)
            P.cmean, P.cpt = cmean, cpt
            if mode == 'seq2seq':
                m, sc = mask_for('cpt', B, E)
                P.m_cpt, P.sc = m, sc
                self.cpt_feats = self._new(B, E) if m is not None else cpt
                first.append(ops.linear_problem([(cmean, p['cpt2fc.0.weight'])], cpt, p['cpt2fc.0.bias'],
                                                relu=True, keep_mask=m, mask_scale=sc,
                                                out_pre=self.cpt_feats if m is not None else None))
                P.fc_e = cpt          # captioner.py:250-251: fc_feats := dropout(cpt_feats)
            else:
                self.cpt_feats = cpt
                first.append(ops.linear_problem([(cmean, p['cpt2fc.0.weight'])], cpt, p['cpt2fc.0.bias'],
                                                relu=True))
        ops.linear_fwd(first)
        if senti_labels is not None:
            P.label_e = self._new(B, Wd)
            P.label_ids = self._ids(senti_labels).reshape(-1)
            ops.embed_relu_fwd(p['senti_label_embed.0.weight'], P.label_ids, P.label_e)
            m, sc = mask_for('label', B, Wd)
            P.m_label, P.sc = m, sc
            if m is not None:
                P.label_e = P.label_e * (m.float() * sc)   # [B,W] elementwise, train mode only
            if senti_words is not None:
                P.label_w = self._new(B, A)
                ops.linear_fwd([ops.linear_problem(
                    [(P.label_e, p['attention.senti_att.label2word.weight'])], P.label_w,
                    p['attention.senti_att.label2word.bias'])])
        second = []
        if mode != 'seq2seq':
            m, sc = mask_for('att', B * R, E)
            P.x_att, P.m_att, P.sc = att.reshape(B * R, -1), m, sc
            att_e = self._new(B * R, E)
            ops.linear_fwd([ops.linear_problem([(att.reshape(B * R, -1), p['att_embed.0.weight'])], att_e,
                                               p['att_embed.0.bias'], relu=True, keep_mask=m, mask_scale=sc)])
            att_p = self._new(B * R, A)
            second.append(ops.linear_problem([(att_e, p['att2att.0.weight'])], att_p,
                                             p['att2att.0.bias'], relu=True))
            P.att_e3, P.att_p3 = att_e.view(B, R, E), att_p.view(B, R, A)
        if senti_words is not None:
            sw = self._ids(senti_words).reshape(B, -1)
            P.Mw = sw.shape[1] + 1
            m, sc = mask_for('words', B * P.Mw, Wd)
            P.sw_ids, P.m_words, P.sc = sw, m, sc
        if senti_words is not None and words_table and m is None:
            # captioner.py:307-312 without dropout: word_embed(id) and senti2att(word_embed(id)) are functions of the
            # id alone -> two [V,.] tables (41 MB, cache-resident) instead of [B,M,.] copies re-read every step
            P.words_e3, P.words_p3 = self._senti_tables(p)
            P.words_ids = torch.cat([sw.new_full((B, 1), self.pad_id), sw], dim=1).contiguous()
        elif senti_words is not None:
            words_e = self._new(B * P.Mw, Wd)
            ops.embed_senti_words_fwd(p['word_embed.0.weight'], sw, self.pad_id, words_e, m, sc)
            words_p = self._new(B * P.Mw, A)
            second.append(ops.linear_problem([(words_e, p['senti2att.0.weight'])], words_p,
                                             p['senti2att.0.bias'], relu=True))
            P.words_e3, P.words_p3 = words_e.view(B, P.Mw, Wd), words_p.view(B, P.Mw, A)
        if second:
            ops.linear_fwd(second)
        if (0 < gate_rows <= self.GATE_FUSED_MAX_ROWS and getattr(self, 'gate_fused', True) and masks is None
                and P.att_e3 is not None and P.words_e3 is not None and A == E == Wd and A <= 1024):
            # G_c = att_e cont2att.weight^T per region; G_s = words_e senti2att.weight^T per word (or per vocabulary
            # entry in table mode) - no bias: the biases are added once, with the h-term, inside the kernel
            P.gate_Gc = self._new(B * R, A)
            gl = [ops.linear_problem([(P.att_e3.view(B * R, E), p['attention.cont2att.weight'])], P.gate_Gc)]
            if P.words_ids is not None:
                P.gate_Gs = self._gate_senti_table(p, P.words_e3)
            else:
                P.gate_Gs = self._new(B * P.Mw, A)
                gl.append(ops.linear_problem([(P.words_e3.view(B * P.Mw, Wd), p['attention.senti2att.weight'])],
                                             P.gate_Gs))
            ops.linear_fwd(gl)
        # Hoist the step-invariant inputs of the att-LSTM (captioner.py:174: cat[h_lang, fc, xt] with
        # xt = relu(Emb[it]) + label_e): fc_e W_fc^T + label_e W_x^T + b_ih + b_hh is computed once.
        H = st['rnn_hid_dim']
        Wih = p['att_lstm.weight_ih']
        segs = [(P.fc_e, Wih[:, H:H + E])]
        if P.label_e is not None:
            segs.append((P.label_e, Wih[:, H + E:]))
        # (pre1_out: the caller's [B,4H] row block of a larger buffer - the merged unroll of two sibling calls keeps
        # both calls' rows in one [B1+B2,4H] tensor)
        P.pre1 = self._new(B, 4 * H) if pre1_out is None else pre1_out
        ops.linear_fwd([ops.linear_problem(segs, P.pre1, p['att_lstm.bias_ih'], p['att_lstm.bias_hh'])])
        if want_table:
            P.tab = self._embedding_table(p, build=(want_table == 'build'))
        return P

    def _senti_tables(self, p):
        """(relu(Emb) [V,W], relu(senti2att(relu(Emb))) [V,A]), cached until the embedding or senti2att change."""
        emb, W2, b2 = p['word_embed.0.weight'], p['senti2att.0.weight'], p['senti2att.0.bias']
        key = (emb.data_ptr(), emb._version, W2.data_ptr(), W2._version, b2.data_ptr(), b2._version, ops.WEIGHT_EPOCH,
               ops.h3_mode() == 0)          # (a table built by one GEMM engine is not handed to a call on the other)
        cached = self._table_cache('_senti_tab_cache', key)
        if cached is not None:
            return cached
        V, Wd, A = self.vocab_size, self.settings['word_emb_dim'], self.settings['att_hid_dim']
        act = self._new(V, Wd)
        ops.embed_relu_fwd(emb, torch.arange(V, dtype=torch.int64, device=self._dev), act)
        proj = self._new(V, A)
        ops.linear_fwd([ops.linear_problem([(act, W2)], proj, b2, relu=True)])
        self._table_cache('_senti_tab_cache', key, (act, proj))
        return act, proj

    def _table_cache(self, name, key, value=None):
        """Weight-derived tables, ONE entry per GEMM engine (the last element of `key`: exact-fp32 engine or not): a call
        served on the other engine - an out-of-domain input, `--h3-mode 0` - builds its own table and leaves the first
        engine's in place (a single slot made every switch back rebuild, or miss, the 82 MB token table)."""
        slots = self.__dict__.setdefault(name, {})
        if not isinstance(slots, dict):
            slots = self.__dict__[name] = {}
        if value is None:
            e = slots.get(key[-1])
            return e[1] if e is not None and e[0] == key else None
        slots[key[-1]] = (key, value)
        return value

    def _gate_senti_table(self, p, act):
        """relu(Emb) attention.senti2att.weight^T [V,A]: the sentiment-word table carried through the gate's projection
        (isc_attn_scan_gate_fwd), cached until the embedding or that weight change."""
        emb, Wg = p['word_embed.0.weight'], p['attention.senti2att.weight']
        key = (emb.data_ptr(), emb._version, Wg.data_ptr(), Wg._version, ops.WEIGHT_EPOCH, ops.h3_mode() == 0)
        cached = self._table_cache('_gate_tab_cache', key)
        if cached is not None:
            return cached
        tab = self._new(act.shape[0], Wg.shape[0])
        ops.linear_fwd([ops.linear_problem([(act, Wg)], tab)])
        return self._table_cache('_gate_tab_cache', key, tab)

    def _embedding_table(self, p, build):
        """relu(Emb) W_x^T [V,4H], cached until the embedding or the att-LSTM weights change
        (tensor version counters). Only used without autograd; costs V*4H*W*2 flop (21 GFLOP) once."""
        emb, Wih = p['word_embed.0.weight'], p['att_lstm.weight_ih']
        key = (emb.data_ptr(), emb._version, Wih.data_ptr(), Wih._version, ops.WEIGHT_EPOCH, ops.h3_mode() == 0)
        cached = self._table_cache('_tab_cache', key)
        if cached is not None:
            return cached
        if not build:
            return None
        st = self.settings
        H, E, Wd, V = st['rnn_hid_dim'], st['feat_emb_dim'], st['word_emb_dim'], self.vocab_size
        act = self._new(V, Wd)
        ops.embed_relu_fwd(emb, torch.arange(V, dtype=torch.int64, device=self._dev), act)
        tab = self._new(V, 4 * H)
        ops.linear_fwd([ops.linear_problem([(act, Wih[:, H + E:])], tab)])
        return self._table_cache('_tab_cache', key, tab)

    # ------------------------------------------------------------------ one decode step
    ROWS_STEP_MAX = 8              # decode rows up to which an inference step runs on the few-row kernels (csrc/rows.hip)

    def _rows_step_ok(self, rows, P):
        """The conditions of isc_rows_step_supported that are known before a plan exists (both attentions with the
        pre-projected gate rows, equal widths <= 512); `self.rows_step = False` keeps the general kernels."""
        st = self.settings
        E, A, Wd, H = st['feat_emb_dim'], st['att_hid_dim'], st['word_emb_dim'], st['rnn_hid_dim']
        return (getattr(self, 'rows_step', True) and 0 < rows <= self.ROWS_STEP_MAX and not ops.TIMER.armed
                and self._rows_vocab_ok()
                and ops.TIMER.arm_step is None and getattr(P, 'gate_Gc', None) is not None
                and getattr(P, 'gate_Gs', None) is not None and A == E == Wd and A <= 512 and A % 4 == 0 and H % 4 == 0
                and H <= 1024)

    def _rows_vocab_ok(self):
        """The few-row classifier keeps statistics per isc_rows_stats_tile(V) <= 64 columns and isc_beam_select holds at
        most 256 tile lists (64 lanes x 4): vocabularies beyond 16384 words take the general kernels."""
        V = self.vocab_size
        tw = ops.rows_stats_tile(V)
        return tw > 0 and (V + tw - 1) // tw <= 256

    def _alloc_step_ws(self, rows, P, stats_tile=128):
        st = self.settings
        E, A = st['feat_emb_dim'], st['att_hid_dim']
        has_cont, has_senti = P.att_e3 is not None, P.words_e3 is not None
        ws = {}
        if has_cont:
            ws['qa'], ws['v'] = self._new(rows, A), self._new(rows, E)
        if has_senti:
            ws['qw'], ws['s'] = self._new(rows, A), self._new(rows, E)
        if has_cont and has_senti:
            ws['z'], ws['f'] = self._new(rows, A), self._new(rows, E)
        n_tile = (self.vocab_size + stats_tile - 1) // stats_tile
        ws['pmax'] = self._new(rows, n_tile)
        ws['psum'] = self._new(rows, n_tile)
        ws['pidx'] = self._new(rows, n_tile, dtype=torch.int32)
        return ws

    def _make_plan(self, p, P, rows):
        """isc_step_plan with everything that does not change between steps filled in."""
        st = self.settings
        pl = _lib.StepPlan()
        pl.rows, pl.H, pl.E, pl.A = rows, st['rnn_hid_dim'], st['feat_emb_dim'], st['att_hid_dim']
        pl.W, pl.V, pl.R, pl.Mw = st['word_emb_dim'], self.vocab_size, P.R, P.Mw
        for field, key in (('Wih1', 'att_lstm.weight_ih'), ('Whh1', 'att_lstm.weight_hh'),
                           ('Wih2', 'lang_lstm.weight_ih'), ('Whh2', 'lang_lstm.weight_hh'),
                           ('b_ih2', 'lang_lstm.bias_ih'), ('b_hh2', 'lang_lstm.bias_hh'),
                           ('W_h2att', 'attention.cont_att.h2att.weight'), ('b_h2att', 'attention.cont_att.h2att.bias'),
                           ('w_alpha_c', 'attention.cont_att.att_alpha.weight'),
                           ('b_alpha_c', 'attention.cont_att.att_alpha.bias'),
                           ('W_h2word', 'attention.senti_att.h2word.weight'),
                           ('b_h2word', 'attention.senti_att.h2word.bias'),
                           ('w_alpha_s', 'attention.senti_att.word_alpha.weight'),
                           ('b_alpha_s', 'attention.senti_att.word_alpha.bias'),
                           ('W_gh', 'attention.h2att.weight'), ('b_gh', 'attention.h2att.bias'),
                           ('W_gc', 'attention.cont2att.weight'), ('b_gc', 'attention.cont2att.bias'),
                           ('W_gs', 'attention.senti2att.weight'), ('b_gs', 'attention.senti2att.bias'),
                           ('w_gate', 'attention.att_alpha.weight'), ('b_gate', 'attention.att_alpha.bias'),
                           ('W_cls', 'classifier.weight'), ('b_cls', 'classifier.bias')):
            t = p[key]
            assert t.is_contiguous()
            setattr(pl, field, t.data_ptr())
        for field, t in (('pre1', P.pre1), ('tab', P.tab), ('att_p', P.att_p3), ('att_e', P.att_e3),
                         ('words_p', P.words_p3), ('words_e', P.words_e3), ('label_w', P.label_w)):
            if t is not None:
                assert t.is_contiguous()
                setattr(pl, field, t.data_ptr())
        if P.words_ids is not None:
            pl.words_ids, pl.words_ids_ld = P.words_ids.data_ptr(), P.words_ids.stride(0)
        ws = ops.splitk_ws(self._dev)
        pl.splitk_ws, pl.splitk_ws_floats = ws.data_ptr(), ws.numel()
        return pl

    def _step(self, p, P, ws, xt, h_cur, c_cur, h_nxt, c_nxt, alpha_c=None, alpha_s=None, beta=None,
              logits=None, out_mask=None, out_scale=1.0, save=None, tok=None, normalize=False,
              hp_cur=None, hp_nxt=None, rows_ext=None):
        """forward_step (captioner.py:168-186) on `rows` sequences: ONE library call (isc_step_fwd; with `rows_ext`, an
        isc_rows_ext, isc_rows_step_fwd: the few-row kernels, `ws` then holds statistics per rows_ext.stats_tile columns)
        that enqueues every kernel of the step. h/c arguments are indexable pairs (0 = att-LSTM,
        1 = lang-LSTM) of [rows,H] tensors; `save` (training) holds 'g1','g2' [rows,4H] and 'hdrop'
        [rows,H] buffers kept for the backward pass; `xt` = relu(Emb[token]) or None when P.tab
        serves the token ids `tok`; normalize=True turns `logits` into log-probs in place.
        hp_cur / hp_nxt (optional, both or none): [2 layers, 2 planes, rows, H] f16 split-f16 planes of h_cur /
        h_nxt - read for the state's GEMM segments, written by the LSTM epilogues (isc_step_plan.h1_hi ...).
        While bench.py has the kernel timer armed the per-kernel path below runs instead."""
        if ops.TIMER.armed:
            return self._step_py(p, P, ws, xt, h_cur, c_cur, h_nxt, c_nxt, alpha_c, alpha_s, beta, logits,
                                 out_mask, out_scale, save, tok, normalize, hp_cur, hp_nxt)
        save = save or {}
        pl = ws.get('_plan')
        if pl is None:
            pl = ws['_plan'] = self._make_plan(p, P, h_cur[0].shape[0])
        ptr = ops.ptr
        pl.xt = ptr(xt)
        if tok is not None and P.tab is not None:
            pl.tok, pl.tok_stride = tok.data_ptr(), tok.stride(0)
        pl.h1_prev, pl.h2_prev, pl.c1_prev, pl.c2_prev = (h_cur[0].data_ptr(), h_cur[1].data_ptr(),
                                                          c_cur[0].data_ptr(), c_cur[1].data_ptr())
        pl.h1, pl.h2, pl.c1, pl.c2 = (h_nxt[0].data_ptr(), h_nxt[1].data_ptr(), c_nxt[0].data_ptr(),
                                      c_nxt[1].data_ptr())
        if hp_cur is not None:
            pl.h1_prev_hi, pl.h1_prev_lo = ops.planes_ptrs(hp_cur[0])
            pl.h2_prev_hi, pl.h2_prev_lo = ops.planes_ptrs(hp_cur[1])
            pl.h1_hi, pl.h1_lo = ops.planes_ptrs(hp_nxt[0])
            pl.h2_hi, pl.h2_lo = ops.planes_ptrs(hp_nxt[1])
            for k in ('v', 's', 'f'):                 # plane workspace of the step's own intermediates
                pp = ws.get(k + 'p')
                hi, lo = (None, None) if pp is None else ops.planes_ptrs(pp)
                setattr(pl, k + '_hi', hi)
                setattr(pl, k + '_lo', lo)
        else:
            for k in ('h1_prev_hi', 'h1_prev_lo', 'h2_prev_hi', 'h2_prev_lo', 'h1_hi', 'h1_lo', 'h2_hi', 'h2_lo',
                      'v_hi', 'v_lo', 's_hi', 's_lo', 'f_hi', 'f_lo'):
                setattr(pl, k, None)
        pl.g1, pl.g2 = ptr(save.get('g1')), ptr(save.get('g2'))
        for k in ('qa', 'v', 'qw', 's', 'z', 'f'):
            setattr(pl, k, ptr(ws.get(k)))
        pl.alpha_c, pl.alpha_c_ld = (alpha_c.data_ptr(), alpha_c.stride(0)) if alpha_c is not None else (None, 0)
        pl.alpha_s, pl.alpha_s_ld = (alpha_s.data_ptr(), alpha_s.stride(0)) if alpha_s is not None else (None, 0)
        pl.beta, pl.beta_ld = (beta.data_ptr(), beta.stride(0)) if beta is not None else (None, 0)
        hdrop = None
        if out_mask is not None:
            hdrop = save['hdrop'] if 'hdrop' in save else self._new(h_cur[0].shape[0], pl.H)
        pl.out_mask, pl.out_scale, pl.hdrop = ptr(out_mask), out_scale, ptr(hdrop)
        pl.logits, pl.ld_logits = (logits.data_ptr(), logits.stride(0)) if logits is not None else (None, 0)
        pl.apply_logsoftmax = int(normalize)
        pl.pmax, pl.psum, pl.pidx = ptr(ws.get('pmax')), ptr(ws.get('psum')), ptr(ws.get('pidx'))   # None: no classifier
        pl.gate_Gc, pl.gate_Gs = ptr(getattr(P, 'gate_Gc', None)), ptr(getattr(P, 'gate_Gs', None))
        if rows_ext is not None:
            ops.rows_step_fwd(pl, rows_ext)
        else:
            ops.step_fwd(pl)

    def _step_py(self, p, P, ws, xt, h_cur, c_cur, h_nxt, c_nxt, alpha_c=None, alpha_s=None, beta=None,
                 logits=None, out_mask=None, out_scale=1.0, save=None, tok=None, normalize=False,
                 hp_cur=None, hp_nxt=None):
        """forward_step (captioner.py:168-186) on `rows` sequences. h/c arguments are indexable
        pairs (0 = att-LSTM, 1 = lang-LSTM) of [rows,H] tensors; reads *_cur, writes *_nxt, the
        vocabulary tile statistics and (optionally) raw logits. `save` (training): dict with
        'g1','g2' [rows,4H] gate buffers and 'hdrop' [rows,H] kept for the backward pass.
        `xt` = relu(Emb[token]) (the label term lives in P.pre1); with P.tab the token ids `tok`
        are enough and xt may be None."""
        save = save or {}
        st = self.settings
        E, H = st['feat_emb_dim'], st['rnn_hid_dim']
        Wih, Whh = p['att_lstm.weight_ih'], p['att_lstm.weight_hh']
        rows = h_cur[0].shape[0]
        # att-LSTM over cat[h_lang_prev, fc, xt] (captioner.py:174) without materialising the cat; the
        # fc / label / bias terms come pre-summed (P.pre1), the word term from the table when present
        pc = (lambda l: None) if hp_cur is None else (lambda l: hp_cur[l])      # planes of h_cur[l] / h_nxt[l]
        pn = (lambda l: None) if hp_nxt is None else (lambda l: hp_nxt[l])
        wp = (lambda k: None) if hp_cur is None else (lambda k: ws.get(k + 'p'))   # planes of v / s / f
        segs = [(h_cur[1], Wih[:, 0:H], pc(1)), (h_cur[0], Whh, pc(0))]
        if P.tab is None:
            segs.insert(1, (xt, Wih[:, H + E:]))
        ops.lstm_fwd(segs, None, None, c_cur[0], h_nxt[0], c_nxt[0], gates_out=save.get('g1'),
                     pre=P.pre1, tab=P.tab, tab_ids=tok if P.tab is not None else None, h_planes=pn(0))
        h1 = h_nxt[0]
        has_cont, has_senti = P.att_e3 is not None, P.words_e3 is not None
        probs, scans = [], []
        if has_cont:
            probs.append(ops.linear_problem([(h1, p['attention.cont_att.h2att.weight'], pn(0))], ws['qa'],
                                            p['attention.cont_att.h2att.bias']))
            scans.append(ops.scan_problem(P.att_p3, P.att_e3, ws['qa'],
                                          p['attention.cont_att.att_alpha.weight'],
                                          p['attention.cont_att.att_alpha.bias'], ws['v'], alpha_c,
                                          out_planes=wp('v')))
        if has_senti:
            probs.append(ops.linear_problem([(h1, p['attention.senti_att.h2word.weight'], pn(0))], ws['qw'],
                                            p['attention.senti_att.h2word.bias']))
            scans.append(ops.scan_problem(P.words_p3, P.words_e3, ws['qw'],
                                          p['attention.senti_att.word_alpha.weight'],
                                          p['attention.senti_att.word_alpha.bias'], ws['s'], alpha_s,
                                          q2=P.label_w, out_planes=wp('s'), row_ids=P.words_ids))
        gate = has_cont and has_senti
        if gate:   # the h2att(h1) term of the gate rides in the same launch as the two projections
            probs.append(ops.linear_problem([(h1, p['attention.h2att.weight'], pn(0))], ws['z'],
                                            p['attention.h2att.bias']))
        ops.linear_fwd(probs)
        fused_gate = gate and getattr(P, 'gate_Gc', None) is not None and getattr(P, 'gate_Gs', None) is not None
        if fused_gate:       # few rows, inference: scans + gate sum + gate mix in one launch (as the step plan does)
            ops.attn_scan_gate_fwd(scans, (P.gate_Gc, P.gate_Gs), ws['z'], p['attention.cont2att.bias'],
                                   p['attention.senti2att.bias'], p['attention.att_alpha.weight'],
                                   p['attention.att_alpha.bias'], ws['f'], beta, f_planes=wp('f'))
            feat, featp = ws['f'], wp('f')
        else:
            ops.attn_scan_fwd(scans, rows)
        if fused_gate:
            pass
        elif gate:
            # z = cont2att(v) + senti2att(s) + h2att(h1) (captioner.py:107-110): add the v / s terms
            ops.linear_fwd([ops.linear_problem(
                [(ws['v'], p['attention.cont2att.weight'], wp('v')),
                 (ws['s'], p['attention.senti2att.weight'], wp('s'))],
                ws['z'], p['attention.cont2att.bias'], p['attention.senti2att.bias'], accumulate=True)])
            ops.gate_mix_fwd(ws['z'], p['attention.att_alpha.weight'], p['attention.att_alpha.bias'],
                             ws['v'], ws['s'], ws['f'], beta, out_planes=wp('f'))
            feat, featp = ws['f'], wp('f')
        else:
            feat, featp = (ws['v'], wp('v')) if has_cont else (ws['s'], wp('s'))
        Wih2, Whh2 = p['lang_lstm.weight_ih'], p['lang_lstm.weight_hh']
        hdrop = None
        if out_mask is not None:
            hdrop = save['hdrop'] if 'hdrop' in save else self._new(rows, H)
        ops.lstm_fwd([(feat, Wih2[:, 0:E], featp), (h1, Wih2[:, E:E + H], pn(0)), (h_cur[1], Whh2, pc(1))],
                     p['lang_lstm.bias_ih'], p['lang_lstm.bias_hh'], c_cur[1], h_nxt[1], c_nxt[1],
                     gates_out=save.get('g2'), h_keep_mask=out_mask, mask_scale=out_scale, hdrop_out=hdrop,
                     h_planes=pn(1))
        if ws.get('pmax') is None:            # the caller projects all steps at once after its unroll
            return
        ops.vocab_fwd(hdrop if hdrop is not None else h_nxt[1], p['classifier.weight'],
                      p['classifier.bias'], ws['pmax'], ws['psum'], ws['pidx'], logits,
                      h_planes=None if hdrop is not None else pn(1))
        if normalize:
            ops.logsoftmax_apply(logits, ws['pmax'], ws['psum'])

    def _set_weights(self, aC, aS, bG, steps):
        """attention._get_weights (captioner.py:83-94): per-step weights concatenated along dim 1.
        `steps` may be the roll-out's device-side `alive` counter vector instead of an int: the number of
        executed steps (the reference's early `break`, captioner.py:343-344) is then read from the device
        only if somebody looks at the weights - the roll-out itself stays free of host syncs."""
        self._weights_pending = (aC, aS, bG, steps)

    def _resolve_weights(self):
        pend = self.__dict__.get('_weights_pending')
        if pend is None:
            return
        aC, aS, bG, steps = pend
        self._weights_pending = None
        if torch.is_tensor(steps):                       # alive[t+1] == 0: no row unfinished after step t
            alive_h = steps.cpu().tolist()
            T = len(alive_h) - 1
            steps = next((t + 1 for t in range(T) if alive_h[t + 1] == 0), T)
        B = (aC if aC is not None else aS).shape[0]
        self._weights = (aC[:, :steps].reshape(B, -1) if aC is not None else [],
                         aS[:, :steps].reshape(B, -1) if aS is not None else [],
                         bG[:, :steps] if bG is not None else [])

    def _weight_get(self, i):
        self._resolve_weights()
        return self.__dict__.get('_weights', ([], [], []))[i]

    def _weight_set(self, i, value):
        self._resolve_weights()
        w = list(self.__dict__.get('_weights', ([], [], [])))
        w[i] = value
        self._weights = tuple(w)

    cont_weights = property(lambda self: self._weight_get(0), lambda self, v: self._weight_set(0, v))
    senti_weights = property(lambda self: self._weight_get(1), lambda self, v: self._weight_set(1, v))
    cont_senti_weights = property(lambda self: self._weight_get(2), lambda self, v: self._weight_set(2, v))

    # ------------------------------------------------------------------ modes
    def forward(self, *args, **kwargs):
        mode = kwargs.pop('mode', 'xe')
        return getattr(self, 'forward_' + mode)(*args, **kwargs)

    def _needs_grad(self):
        return torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters())

    def _teacher_forced(self, p, P, tokens_in, ss_prob, masks):
        """Unroll feeding tokens_in[:, i] (with scheduled sampling in train mode,
        captioner.py:218-234). Returns log-probs [B,T,V]."""
        B, T = tokens_in.shape
        V, Wd = self.vocab_size, self.settings['word_emb_dim']
        R, Mw = P.R, P.Mw
        h, c = [self._zeros(2, B, self.att_lstm.hidden_size) for _ in range(2)], \
               [self._zeros(2, B, self.att_lstm.hidden_size) for _ in range(2)]
        ws = self._alloc_step_ws(B, P)
        out = self._new(B, T, V)
        aC = self._new(B, T, R) if P.att_e3 is not None else None
        aS = self._new(B, T, Mw) if P.words_e3 is not None else None
        bG = self._new(B, T) if (aC is not None and aS is not None) else None
        xt = self._new(B, Wd)
        mask_for = self._mask_source(masks)
        emb = p['word_embed.0.weight']
        mask_for.predraw('out', T, B, self.att_lstm.hidden_size)
        with ops.h3_weights_scope(self._dev, key=self._weights_key()):      # frozen weights for the whole unroll
            for i in range(T):
                it = tokens_in[:, i]
                if self.training and i >= 1 and ss_prob > 0.0:       # scheduled sampling, on the device (no host test)
                    u = torch.rand(2, B, device=self._dev)
                    drawn = torch.empty(B, dtype=torch.int64, device=self._dev)
                    ops.sched_sample(out[:, i - 1], ws['pmax'], ws['psum'], ws['pidx'], u[0], u[1], ss_prob, it, drawn)
                    it = drawn
                ops.embed_relu_fwd(emb, it.contiguous(), xt)
                om, osc = mask_for('out%d' % i, B, self.att_lstm.hidden_size)
                cur, nxt = i & 1, (i + 1) & 1
                logits = out[:, i]
                self._step(p, P, ws, xt, h[cur], c[cur], h[nxt], c[nxt],
                           aC[:, i] if aC is not None else None, aS[:, i] if aS is not None else None,
                           bG[:, i:i + 1] if bG is not None else None, logits, om, osc, normalize=True)
        self._set_weights(aC, aS, bG, T)
        return out

    def token_logprobs(self, on=True):
        """`with captioner.token_logprobs():` - teacher-forced calls WITH gradients (`forward_xe`, `forward_seq2seq`,
        `forward_xe_seq2seq`) return log p(target) [B,T] (targets = captions[:, 1:], what XECriterion gathers:
        captioner.py:427-440) instead of the [B,T,V] log-probs, and `XECriterion` takes that tensor.  The [B,T,V] tensor
        is then never written or read in the iteration - forward or backward (0.56 ms of 15.6 at B = 1024) - and the loss
        and every gradient keep their bits: log p comes from the raw logits by the expression the tensor would have been
        stored with.  The package's own training steps (train.py, train_graph.py, Detector.forward) run inside it; the
        reference's call surface (a caller that wants `pred`) is untouched outside."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            prev = self.__dict__.get('_token_logprobs')
            self.__dict__['_token_logprobs'] = bool(on) and getattr(self, 'fused_criteria', True)
            try:
                yield self
            finally:
                self.__dict__['_token_logprobs'] = prev
        return scope()

    # `ragged_unroll = 'auto'`: from this many rows on the eager step with the ragged unroll beats both the merged eager step
    # and the graph-served iteration (13.5-13.9 ms against 15.8 / 15.0 at 1024 + 80 rows; at 512 rows it is a tie with the
    # eager steps, 9.2-10.1 against 9.5-9.9, and behind the graphs' 8.7): tools/ragged_probe.py, bench.py
    # xe_train*.ragged_eager_ms_per_iter
    RAGGED_AUTO_ROWS = 1024

    def ragged_applies(self, lengths):
        """Whether `row_counts(lengths)` would shorten the unroll: the flag is on, `lengths` is a host list sorted longest
        first with at least one caption shorter than the longest (train.xe_forward_backward then runs one chain per unroll
        instead of the merged chain, whose two row blocks are not one prefix)."""
        with self.row_counts(lengths):
            c = self.__dict__.get('_row_counts')
        return c is not None and c[-1] < c[0]

    def row_counts(self, lengths, T=None):
        """`with captioner.row_counts(lengths):` - the teacher-forced unroll WITH gradients inside (one `forward_xe` or
        `forward_seq2seq` call under `token_logprobs()`, scheduled sampling off) runs step t on the rows whose caption has
        not ended yet.  `lengths`: XECriterion's host list (captioner.py:431-436: row i counts for steps < lengths[i]),
        sorted longest first as the reference's collates hand it over (dataloader.py:17,37,68,124) - anything else (a device
        tensor, an unsorted list) leaves the unroll as it is.  Loss and gradients are the full unroll's: positions behind
        a caption's end are masked by the criterion, so nothing reads them and their gradient is exactly zero."""
        import contextlib
        counts = None
        mode = getattr(self, 'ragged_unroll', False)      # False | True | 'auto' (batches of >= RAGGED_AUTO_ROWS rows)
        if isinstance(lengths, (list, tuple)) and len(lengths) > 0 and mode and (
                mode != 'auto' or len(lengths) >= self.RAGGED_AUTO_ROWS):
            ls = [int(x) for x in lengths]
            steps = max(ls) if T is None else T
            if all(a >= b for a, b in zip(ls, ls[1:])) and ls[-1] >= 1:
                counts = [sum(1 for x in ls if x > t) for t in range(steps)]

        @contextlib.contextmanager
        def scope():
            prev = self.__dict__.get('_row_counts')
            self.__dict__['_row_counts'] = counts
            try:
                yield self
            finally:
                self.__dict__['_row_counts'] = prev
        return scope()

    def forward_xe(self, fc_feats, att_feats, cpt_words, captions, senti_labels, ss_prob=0.0, _masks=None, _targets=None):
        """(_targets: test hook for token_logprobs() - the criterion's targets when `captions` carries replayed FED tokens
        instead of the ground truth; default captions[:, 1:].)"""
        if self._needs_grad():
            from .autograd import xe_with_grad
            return xe_with_grad(self, 'xe', fc_feats, att_feats, cpt_words, None, captions, senti_labels,
                                ss_prob, _masks, _targets)
        p = self._p()
        P = self._prologue(p, 'xe', fc_feats, att_feats, cpt_words, None, senti_labels, _masks)
        return self._teacher_forced(p, P, self._ids(captions)[:, :-1], ss_prob, _masks)

    def forward_seq2seq(self, senti_captions, cpt_words, senti_words, senti_labels, ss_prob=0.0, _masks=None,
                        _targets=None):
        if self._needs_grad():
            from .autograd import xe_with_grad
            return xe_with_grad(self, 'seq2seq', None, None, cpt_words, senti_words, senti_captions,
                                senti_labels, ss_prob, _masks, _targets)
        p = self._p()
        P = self._prologue(p, 'seq2seq', None, None, cpt_words, senti_words, senti_labels, _masks)
        return self._teacher_forced(p, P, self._ids(senti_captions)[:, :-1], ss_prob, _masks)

    def forward_xe_seq2seq(self, fc_feats, att_feats, cpt_words, captions, senti_labels, ss_prob,
                           s_captions, s_cpt_words, s_senti_words, s_senti_labels, s_ss_prob=None,
                           _masks=None, _s_masks=None, _targets=None, _s_targets=None):
        """`forward_xe(...)` and `forward_seq2seq(...)` of ONE training iteration (train_xe.py:160-181,
        models/decoder.py:138-157) as one call: returns (pred, pred2), leaves `fc_feats` / `cpt_feats` as the XE call
        leaves them (the domain-align loss reads them right after it, train_xe.py:163) and the seq2seq call's
        `cpt_feats` in `s2s_cpt_feats`.  With gradients the two unrolls share one step chain (autograd_pair: both LSTM
        cells, the classifier and every backward contraction once over the rows of both calls); without, or when
        `self.pair_unrolls` is False, it is exactly the two calls in the reference's order."""
        s_ss_prob = ss_prob if s_ss_prob is None else s_ss_prob
        from .autograd_pair import pair_applicable, pair_with_grad
        if self._needs_grad() and self.pair_unrolls is not False and pair_applicable(self, _masks, _s_masks):
            pred, pred2, self.s2s_cpt_feats = pair_with_grad(
                self, fc_feats, att_feats, cpt_words, captions, senti_labels, ss_prob, s_captions, s_cpt_words,
                s_senti_words, s_senti_labels, s_ss_prob, _masks, _s_masks, _targets, _s_targets)
            return pred, pred2
        pred = self.forward_xe(fc_feats, att_feats, cpt_words, captions, senti_labels, ss_prob, _masks=_masks,
                               _targets=_targets)
        keep = (self.fc_feats, self.cpt_feats)
        pred2 = self.forward_seq2seq(s_captions, s_cpt_words, s_senti_words, s_senti_labels, s_ss_prob, _masks=_s_masks,
                                     _targets=_s_targets)
        self.s2s_cpt_feats = self.cpt_feats
        self.fc_feats, self.cpt_feats = keep
        return pred, pred2

    def forward_rl(self, fc_feats, att_feats, cpt_words, senti_words, senti_labels, max_seq_len, sample_max,
                   _replay=None, _masks=None):
        """Greedy (`sample_max=1`) or sampled roll-out (captioner.py:290-349) with the whole T-step loop
        enqueued without a host sync; `_replay` [B,T] forces the raw draws (parity tests)."""
        if not sample_max and self._needs_grad():
            from .autograd import rollout_with_grad
            return rollout_with_grad(self, fc_feats, att_feats, cpt_words, senti_words, senti_labels,
                                     max_seq_len, _replay, _masks)
        if (sample_max and self.__dict__.get('_rollout_graphs') is not None and not self._needs_grad()
                and not self.training and _replay is None and _masks is None and ops.TIMER.arm_step is None
                and fc_feats.shape[0] <= self.ROLLOUT_GRAPH_MAX_ROWS and ops.graphs_allowed_here()
                and fc_feats.is_cuda and not torch.cuda.is_current_stream_capturing()   # (a training graph's capture runs its greedy baseline
                                                                        # here: it IS being captured - no graph inside a graph)
                and self._features_in_domain(fc_feats, att_feats)):     # (beyond the domain: eager, exact engine)
            return self._graphed_rollout(fc_feats, att_feats, cpt_words, senti_words, senti_labels, max_seq_len)
        return self._rollout(fc_feats, att_feats, cpt_words, senti_words, senti_labels, max_seq_len,
                             sample_max, _replay, _masks)[:3]

    # ------------------------------------------------------------------ hipGraph replay of the greedy roll-out
    def enable_rollout_graphs(self, on=True, max_graphs=4):
        """Serve eval-mode greedy roll-outs (forward_rl, sample_max=1) from captured HIP graphs.  Below a few
        hundred captions a roll-out is ~260 dependent small launches: the host needs 110-140 us per decode step
        to enqueue them, the device ~105 us to run them.  The roll-out has no host read, so the whole T-step loop
        (prologue included) captures into ONE graph per input geometry; a call then costs one input copy and one
        graph launch.  Inputs must keep their shapes to hit the cache; a graph belongs to the weight VALUES it was
        captured under (it contains no weight-split launches: the f16 planes stay on the captioner's roll-out stream) -
        after any weight change the next call runs eagerly and the one after captures anew."""
        self._rollout_graphs = {} if on else None
        self._rollout_graphs_max = max_graphs
        self._graphs_explicit = bool(on)

    def _rollout_stream(self):
        """The stream every roll-out graph of this captioner is warmed up and captured on.  Everything the library keeps
        per stream - split-K workspace, f16 weight-plane buffer, weights-scope slot - is then the graphs' own: the
        planes a warm-up run builds there are still there, at the same addresses, when a replay reads them."""
        st = self.__dict__.get('_ro_stream')
        if st is None or st.device != self._dev:
            st = self.__dict__['_ro_stream'] = ops.private_stream(self._dev)
            idx = self._dev.index if self._dev.index is not None else torch.cuda.current_device()
            weakref.finalize(self, ops.release_stream_state, idx, st.cuda_stream)
        return st

    def _graphed_rollout(self, fc_feats, att_feats, cpt_words, senti_words, senti_labels, T):
        ins = [self._f32(fc_feats), self._f32(att_feats), cpt_words, senti_words, senti_labels]
        # keyed on the weights' VALUES (every parameter's storage and version, the fused optimizer's epoch): a graph
        # holds no weight-split launches - the planes its warm-up run left on the roll-out stream are valid exactly
        # as long as this key is.  After a weight change the first call runs eagerly there (rebuilding the planes),
        # the second captures anew.
        key = (tuple((tuple(x.shape), x.dtype) for x in ins), T, self._weights_key(), torch.cuda.current_device())
        cache = self._rollout_graphs
        st = self._rollout_stream()
        idx = self._dev.index if self._dev.index is not None else torch.cuda.current_device()
        skey = ((idx, st.cuda_stream),)
        cur = torch.cuda.current_stream(self._dev)

        def on_stream(fn):                      # fn() on the roll-out stream, ordered after / before the caller's
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                out = fn()
            cur.wait_stream(st)
            return out
        entry = cache.get(key)
        if isinstance(entry, tuple) and entry[4] != ops.h3_weights_scope.cold_begins(skey):
            entry = cache[key] = 'seen'         # the stream's scope was rebuilt since the capture: addresses may differ
        if entry is None:                       # first sight: run eagerly (warms kernels, builds tables and planes)
            while len(cache) >= self._rollout_graphs_max:
                self._graph_evicted(cache.pop(next(iter(cache))), 'roll-out')    # least recently used
                if self._rollout_graphs is None:
                    return self._rollout(*ins, T, 1, None, None)[:3]
            cache[key] = 'seen'
            outs = on_stream(lambda: self._rollout(*ins, T, 1, None, None)[:3])
            for o in outs:
                o.record_stream(cur)
            return outs
        cache[key] = cache.pop(key)             # LRU order
        if entry == 'seen':
            static = [x.clone() for x in ins]
            graph = torch.cuda.CUDAGraph()

            def capture():
                with ops.graph_capture(graph, stream=st):
                    o = self._rollout(*static, T, 1, None, None)[:3]
                    return o, self.__dict__.get('_weights_pending')
            outs, pending = on_stream(capture)
            entry = cache[key] = (graph, static, outs, pending, ops.h3_weights_scope.cold_begins(skey))
        graph, static, outs, pending = entry[:4]
        ops.stage_inputs(static, ins)            # inputs -> static buffers, one launch
        graph.replay()
        self._weights_pending = pending          # attention weights of this replay (resolved lazily, as always)
        fresh = [torch.empty_like(o) for o in outs]       # the caller's own copies of the graph's outputs: one launch
        ops.copy_multi(fresh, outs)
        return tuple(fresh)

    GRAPH_THRASH_LIMIT = 8

    def _graph_evicted(self, entry, what):
        """A CAPTURED graph fell out of a cache of `max_graphs` geometries.  Graph serving that is on by default is for
        callers with a few recurring geometries; one whose batch size wanders would pay a capture (~0.1 s: device-wide
        synchronisation, allocator sweep) every few calls.  After GRAPH_THRASH_LIMIT such evictions both default caches
        switch themselves off for this captioner (eager from then on, same results); enable_*_graphs(True, max_graphs=n)
        switches them back on with room for the caller's geometries."""
        if isinstance(entry, str):
            return
        n = self.__dict__['_graph_evictions'] = self.__dict__.get('_graph_evictions', 0) + 1
        if n == self.GRAPH_THRASH_LIMIT and not self.__dict__.get('_graphs_explicit', False):
            import warnings
            warnings.warn('insenticap_model_amd: %d captured %s / beam graphs evicted - input geometries keep changing; '
                          'default graph serving is off for this captioner (enable_rollout_graphs / enable_beam_graphs '
                          'with a larger max_graphs to keep it)' % (n, what))
            self._rollout_graphs = None
            self._beam_graphs = None

    def enable_beam_graphs(self, on=True, max_graphs=4):
        """Serve beam searches (sample / sample_batch, device-side merge) from captured HIP graphs: the prologue and
        steps 0-3 in one graph, every further four steps in another, the live-image counter read between them
        (beam.py: _graphed_search).  Inputs must keep their shapes to hit the cache; weights may change in place
        (a change of the embedding / att-LSTM / senti2att weights re-captures: the cached tables depend on them)."""
        self._beam_graphs = {} if on else None
        self._beam_graphs_max = max_graphs
        self._graphs_explicit = bool(on)

    def _weights_key(self):
        """Identifies the current parameter VALUES of THIS instance (its nonce + storage pointers + version counters +
        the epoch bumped by the fused optimizer, which writes behind torch's back): equal keys => the f16 weight planes
        of an earlier call are still valid (ops.h3_weights_scope(key=...)).  The nonce matters: the caching allocator
        hands a freed model's addresses to the next one of the same shapes, with equal version counters."""
        return (('captioner', self._wnonce[0]),) + tuple((q.data_ptr(), q._version) for q in self.parameters()) + \
            (ops.WEIGHT_EPOCH,)

    def _renew_weights_nonce(self):
        self._wnonce[0] = next(_INSTANCE_NONCE)

    def _apply(self, fn, *args, **kwargs):               # .to() / .cuda() / .float(): new storages
        out = super()._apply(fn, *args, **kwargs)
        self._renew_weights_nonce()
        return out

    def load_state_dict(self, *args, **kwargs):          # new values (copied in place)
        out = super().load_state_dict(*args, **kwargs)
        self._renew_weights_nonce()
        return out

    def _graph_buffers(self):
        """The split-K workspace (128 MB) and weight-plane buffer (192 MB) that every HIP graph captured for this
        captioner - roll-out and beam graphs alike - launches into: 320 MB per captioner, not per cached graph."""
        gb = self.__dict__.get('_graph_bufs')
        if gb is None or gb[0].device != self._dev:
            gb = self.__dict__['_graph_bufs'] = (
                torch.empty(ops.splitk_ws_floats(), dtype=torch.float32, device=self._dev),
                torch.empty(ops.h3w_bytes(), dtype=torch.uint8, device=self._dev))
        return gb

    def _rollout(self, fc_feats, att_feats, cpt_words, senti_words, senti_labels, T, sample_max, replay,
                 masks):
        self._p()                                  # raises on CPU parameters before anything touches the device
        if not self._features_in_domain(fc_feats, att_feats):
            # features beyond the split-f16 domain: the reference decodes whatever its encoder produced
            # (captioner.py:198-214, 294-315) - so does this call, on the exact-fp32 engine
            with ops.exact_fp32_engine(), ops.h3_weights_scope(self._dev, key=self._weights_key()):
                return self._rollout_impl(fc_feats, att_feats, cpt_words, senti_words, senti_labels, T, sample_max,
                                          replay, masks)
        # frozen weights for prologue + loop: split them once per call - or, with unchanged weights, once per run of calls
        with ops.h3_weights_scope(self._dev, key=self._weights_key()):
            return self._rollout_impl(fc_feats, att_feats, cpt_words, senti_words, senti_labels, T, sample_max,
                                      replay, masks)

    # ------------------------------------------------------------------ operand domain of the split-f16 engine
    SPLIT_F16_MAX = 65504.0

    def _features_in_domain(self, *feats):
        """True when the caller's feature tensors may go through the split-f16 GEMM engine (every |x| < 65504, no NaN /
        inf - those are decoded as they are by the exact engine too, as the reference does).  With `numerics_checks`
        off, the engine off, inside a graph capture or while an owner does its own checking (`_domain_check_off`:
        Detector.forward looks at the numerics flags at its own synchronisation point instead) no check.  One reduction
        over the features and ONE host read per NEW feature tensor: the verdict is remembered per tensor OBJECT and
        version (a loop over one batch pays once; a roll-out of 16384 captions reads 4.8 GB = 0.8 ms of its 37 ms)."""
        feats = [x for x in feats if x is not None and torch.is_tensor(x) and x.is_floating_point()]
        if (not getattr(self, 'numerics_checks', True) or self.__dict__.get('_domain_check_off') or not feats
                or not all(x.is_cuda for x in feats)          # (CPU tensors: the call itself raises - no CPU path)
                or ops.h3_mode() == 0 or torch.cuda.is_current_stream_capturing()):
            return True
        memo = self.__dict__.setdefault('_domain_memo', {})
        verdict, fresh = True, []
        for x in feats:
            e = memo.get(id(x))
            if e is not None and e[0]() is x and e[1] == x._version:
                verdict = verdict and e[2]
            else:
                fresh.append(x)
        if fresh:
            # max |x| by ONE reduction pass per tensor (no |x| copy of a 4.8 GB feature batch); NaN propagates
            worst = torch.stack([torch.linalg.vector_norm(x.detach().reshape(-1), float('inf')).float()
                                 for x in fresh]).amax()
            v = float(worst)                 # (NaN compares False: NaN / inf inputs also take the exact engine)
            ok = bool(v < self.SPLIT_F16_MAX)
            for k in [k for k, e in memo.items() if e[0]() is None]:
                del memo[k]
            while len(memo) >= 16:
                memo.pop(next(iter(memo)))
            for x in fresh:
                memo[id(x)] = (weakref.ref(x), x._version, ok)
            verdict = verdict and ok
            if not ok:
                self._warn_out_of_domain('max |x| = %g' % v)
        return verdict

    def _known_out_of_domain(self, *feats, remember=False):
        """Feature tensors a search was already redone for (Captioner.sample_batch): keyed by storage address, size and
        version - `sample` hands in views, whose object identity changes from call to call.  A stale entry (another
        tensor at the same address) only costs the fast engine for that call, never a wrong result."""
        bad = self.__dict__.setdefault('_domain_bad', {})
        keys = [(x.data_ptr(), x.numel(), x._version) for x in feats
                if x is not None and torch.is_tensor(x) and x.is_floating_point()]
        if remember:
            while len(bad) >= 16:
                bad.pop(next(iter(bad)))
            for k in keys:
                bad[k] = True
            return True
        return any(k in bad for k in keys)

    def _warn_out_of_domain(self, what):
        if not self.__dict__.get('_domain_warned'):
            import warnings
            self.__dict__['_domain_warned'] = True
            warnings.warn('insenticap_model_amd: features beyond the split-f16 operand domain |x| < 65504 (%s): the call '
                          'runs on the exact-fp32 GEMM engine (~2x slower; results as the reference\'s). Scale the '
                          'features to keep the fast engine.' % what)

    def _rollout_impl(self, fc_feats, att_feats, cpt_words, senti_words, senti_labels, T, sample_max, replay,
                      masks):
        p = self._p()
        arm = ops.TIMER.arm_step          # bench.py: time the kernels of ONE step (-1: the prologue)
        ops.TIMER.armed, ops.TIMER.phase = (arm == -1), 'prologue'
        # frozen weights + enough token-steps: the relu(Emb) W_x^T table pays for itself (21 GFLOP once)
        n_rows = fc_feats.shape[0]
        want = False
        if not torch.is_grad_enabled() or not any(q.requires_grad for q in self.parameters()):
            want = 'build' if n_rows * T >= self.vocab_size // 4 else 'cached'
        P = self._prologue(p, 'rl', fc_feats, att_feats, cpt_words, senti_words, senti_labels, masks,
                           want_table=want, words_table=bool(want) and getattr(self, 'words_table', True),
                           gate_rows=n_rows if want else 0)
        ops.TIMER.armed, ops.TIMER.phase = False, 'step'
        B, V = P.B, self.vocab_size
        H, Wd = self.att_lstm.hidden_size, self.settings['word_emb_dim']
        # a handful of captions, no sampling: the few-row kernels (statistics per isc_rows_stats_tile columns)
        rows_ext = None
        if (sample_max or replay is not None) and masks is None and not self.training and self._rows_step_ok(B, P):
            rows_ext = _lib.RowsExt()
            rows_ext.stats_tile = ops.rows_stats_tile(V)
        planes = rows_ext is None and getattr(self, 'state_planes', True)
        # Everything that starts at zero comes out of TWO zeroed arenas (two fill launches, was fifteen): the initial
        # state (the buffers step 0 writes need no zeros) with its f16 planes (zero state = zero planes) - freed with
        # the call -, and what the caller may keep: the outputs, the device-side counters, the attention weights
        f32, i64 = torch.float32, torch.int64
        h0, c0, hp0 = self._zeros_many(((2, B, H), f32), ((2, B, H), f32),
                                       ((2, 2, B, H) if planes else (0,), torch.float16))
        seq, raw, seq_logprobs, seq_masks, alive, aC, aS, bG, unf = self._zeros_many(
            ((B, T), i64), ((B, T), i64), ((B, T), f32), ((B, T), f32), ((T + 1,), torch.int32), ((B, T, P.R), f32),
            ((B, T, P.Mw), f32), ((B, T), f32), ((T + 1, B), torch.int32))
        h, c = [h0, self._new(2, B, H)], [c0, self._new(2, B, H)]
        # split-f16 planes of the state ([layer, hi|lo, B, H] f16): the LSTM epilogues write them next to h, so the
        # state's GEMM segments are never split again
        hp = [hp0, torch.empty(2, 2, B, H, dtype=torch.float16, device=self._dev)] if planes else [None, None]
        ws = self._alloc_step_ws(B, P, rows_ext.stats_tile if rows_ext is not None else 128)
        if hp[0] is not None:
            for k in ('v', 's', 'f'):
                if k in ws:
                    ws[k + 'p'] = torch.empty((2,) + tuple(ws[k].shape), dtype=torch.float16, device=self._dev)
        # the rows' unfinished flags: one array that every finalize updates in place - except on the few-row greedy path,
        # where step t + 1's first launch does step t's finalize and reads unf[t] while it writes unf[t + 1]
        unf[0:1].fill_(1)
        unfinished = unf[0]
        alive[0:1].fill_(B)                     # a fill kernel (a scalar assignment would be a pageable H2D copy)
        use_tab = P.tab is not None
        xt = [None, None] if use_tab else [self._new(B, Wd) for _ in range(2)]
        emb = p['word_embed.0.weight']
        sos = torch.full((B,), self.sos_id, dtype=torch.int64, device=self._dev)
        if not use_tab:
            ops.embed_relu_fwd(emb, sos, xt[0])
        need_logits = (not sample_max)
        logits = self._new(B, V) if need_logits else None
        forced = sample_u = None
        if not sample_max:
            if replay is not None:
                forced = self._ids(replay)
            else:
                sample_u = torch.rand(B, T, device=self._dev)
        mask_for = self._mask_source(masks)
        rs = RolloutStep()
        rs.B, rs.V, rs.T, rs.n_tile, rs.W = B, V, T, ws['pmax'].shape[1], Wd
        rs.part_max, rs.part_sum, rs.part_idx = ws['pmax'].data_ptr(), ws['psum'].data_ptr(), ws['pidx'].data_ptr()
        rs.logits, rs.ld_logits = ops.ptr(logits), V
        rs.forced, rs.sample_u = ops.ptr(forced), ops.ptr(sample_u)
        rs.eos_id = self.eos_id
        rs.seq, rs.seq_logprobs, rs.seq_masks = seq.data_ptr(), seq_logprobs.data_ptr(), seq_masks.data_ptr()
        rs.unfinished, rs.alive, rs.raw_tokens = unfinished.data_ptr(), alive.data_ptr(), raw.data_ptr()
        rs.emb, rs.xt_add = emb.data_ptr(), None      # the label term lives in P.pre1
        # up to four rows, arg-max decoding, token table: step t's finalize rides on step t + 1's att-LSTM launch
        # (isc_rows_ext.fin_prev) - T - 1 launches fewer per roll-out; only the last step's runs on its own
        fuse = (rows_ext is not None and sample_max and replay is None and use_tab and arm is None and B <= 4
                and getattr(self, 'rows_fused_finalize', True))
        rs_prev = RolloutStep.from_buffer_copy(rs) if fuse else None
        with ops.h3_weights_scope(self._dev):      # frozen weights for the whole loop: split them once, not per step
            for t in range(T):
                cur, nxt = t & 1, (t + 1) & 1
                om, osc = mask_for('out%d' % t, B, H)
                ops.TIMER.armed = (arm == t)
                if fuse:
                    if t > 0:
                        rs_prev.t, rs_prev.unfinished = t - 1, unf[t - 1].data_ptr()
                        rows_ext.fin_prev, rows_ext.fin_unfinished_out = C.addressof(rs_prev), unf[t].data_ptr()
                    else:
                        rows_ext.fin_prev = rows_ext.fin_unfinished_out = None
                # token fed at step t: <SOS>, then seq[:, t-1] (= it * unfinished, written by finalize)
                self._step(p, P, ws, xt[cur], h[cur], c[cur], h[nxt], c[nxt], aC[:, t], aS[:, t], bG[:, t:t + 1],
                           logits, om, osc, tok=(sos if t == 0 else seq[:, t - 1]) if use_tab else None,
                           hp_cur=hp[cur], hp_nxt=hp[nxt], rows_ext=rows_ext)
                if fuse and t + 1 < T:
                    continue
                rs.t = t
                rs.unfinished = unf[t].data_ptr() if fuse else unfinished.data_ptr()
                rs.xt_next = None if use_tab else xt[nxt].data_ptr()
                ops.rollout_finalize(rs)
        ops.TIMER.armed = False
        # no host read: the executed-step count stays on the device (`alive`) until someone needs it
        self._set_weights(aC, aS, bG, alive)
        return seq, seq_logprobs, seq_masks, raw, alive

    # ------------------------------------------------------------------ beam search
    def sample(self, fc_feat, att_feat, senti_words=None, senti_label=None,
               beam_size=3, decoding_constraint=1, max_seq_len=16):
        """Beam search for ONE image (captioner.py:351-420): returns (captions, scores)."""
        caps, scores, _ = self.sample_batch(
            fc_feat.reshape(1, -1), att_feat.reshape(1, -1, att_feat.shape[-1]),
            None if senti_words is None else senti_words.reshape(1, -1),
            None if senti_label is None else senti_label.reshape(1),
            beam_size, decoding_constraint, max_seq_len)
        return caps[0], scores[0]

    @torch.no_grad()
    def sample_batch(self, fc_feats, att_feats, senti_words=None, senti_labels=None,
                     beam_size=3, decoding_constraint=1, max_seq_len=16):
        """Beam search for I images at once: every step runs ONE batched decode step over I*beam rows
        and one device top-k; candidate bookkeeping follows the reference exactly (fp64 score sums,
        stable ordering, ended beams carried, captioner.py:378-411).
        Returns (captions[I][beam], scores[I][beam], id_sequences[I][beam])."""
        from .beam import beam_search_batch, replay_if_captured
        if self.training:
            self.eval()
        checks = getattr(self, 'numerics_checks', True)

        def on_exact_engine():
            # features beyond the split-f16 domain: the reference's sample() decodes whatever the encoder produced
            # (captioner.py:357-376) - this search runs eagerly on the exact-fp32 engine
            self._p()
            with ops.exact_fp32_engine(), ops.h3_weights_scope(self._dev, key=self._weights_key()):
                res = beam_search_batch(self, fc_feats, att_feats, senti_words, senti_labels, beam_size,
                                        decoding_constraint, max_seq_len, graphs=False)
            ops.device_status(reset=True)          # (NaN / inf features flag there as well: decoded as they are)
            return res
        # A search ends with a host read of its results, so the numerics flags are visible right behind it: no pass over
        # the features in front of the call (that pass + its host read were 6 % of a one-image search) - a flagged search
        # is redone on the exact engine, and the verdict on these feature tensors is remembered.
        if checks and ops.h3_mode() != 0 and self._known_out_of_domain(fc_feats, att_feats):
            return on_exact_engine()
        out = replay_if_captured(self, fc_feats, att_feats, senti_words, senti_labels, beam_size, decoding_constraint,
                                 max_seq_len)
        if out is None:
            self._p()                              # raises on CPU parameters before anything touches the device
            with ops.h3_weights_scope(self._dev, key=self._weights_key()):     # prologue + search: weights split once
                out = beam_search_batch(self, fc_feats, att_feats, senti_words, senti_labels, beam_size,
                                        decoding_constraint, max_seq_len)
        if checks:                                 # the host has the results, i.e. has waited: read the status too
            if ops.h3_mode() != 0 and ops.device_status(reset=True):
                self._known_out_of_domain(fc_feats, att_feats, remember=True)
                self._warn_out_of_domain('non-finite values in a beam search')
                return on_exact_engine()
            ops.check_numerics('Captioner.sample')
        return out

    def get_optim_criterion(self, lr, weight_decay=0):
        from .optim import FusedClampAdam   # a torch.optim.Adam subclass: same state_dict layout
        return FusedClampAdam(self.parameters(), lr=lr, weight_decay=weight_decay), \
            XECriterion(), nn.MSELoss()  # xe, domain align


class XECriterion(nn.Module):
    """Masked NLL with a global token mean (captioner.py:427-440)."""

    def forward(self, pred, target, lengths):
        # (the reference slices the targets to max(lengths) and masks by row: a `pred` WIDER than the longest caption -
        # captions padded to a fixed width so that every batch has one geometry - gives the same loss; narrower cannot)
        max_len = max(lengths)
        if pred.size(1) < max_len:
            raise ValueError('pred.size(1)=%d is shorter than max(lengths)=%d' % (pred.size(1), max_len))
        if pred.requires_grad:
            from .autograd import xe_criterion_with_grad
            return xe_criterion_with_grad(pred, target, lengths)
        ops.require_device(pred, target)
        out2 = torch.empty(2, dtype=torch.float32, device=pred.device)
        ln = ops.upload(lengths, torch.int32, pred.device)
        ops.xe_loss_fwd(pred.contiguous(), target.long().contiguous(), ln, out2)
        return out2[0] / out2[1]
