"""Batched beam search behind `Captioner.sample` (reference: captioner.py:351-420).

Default: the candidate bookkeeping runs on the device too (`isc_beam_merge`, one workgroup per image, fp64 score sums,
stable descending selection in insertion order), so a step is decode + top-k + merge + state gather with no host read;
the host looks at the "images still searching" counter every fourth step only.  `cap.beam_device_merge = False`
selects the host-side merges below (the same rules in Python / numpy; the CPU tests compare the two host forms, the
GPU tests compare them with the device merge).

The reference decodes one image at a time and runs one batch-1 `forward_step` per live beam
with 2*beam device->host scalar reads each.  Here all I images advance together: one decode
step over I*beam rows + one device top-k per time step and ONE host read of the [I*beam, beam]
(value, id) pairs.  The candidate bookkeeping is the reference's, kept on the host on purpose:
scores are Python floats (fp64 sums of fp32 log-probs, :404-406) and the per-step selection is
a stable descending sort over candidates in insertion order (:409), which fixes tie order.
"""
import numpy as np
import torch

from . import ops


class _ListMerge:
    """Candidate bookkeeping of captioner.py:378-411, image by image in plain Python (fastest for one or two
    images).  step() fills `last_out` [rows] and `gather_out` [rows] (source row of every new row: parent, or
    parent + rows when the candidate is carried and keeps its old state) and returns False when every image is done."""

    def __init__(self, n_img, beam, sos_id, eos_id):
        self.n_img, self.beam, self.eos = n_img, beam, eos_id
        self.cands = [[(0.0, sos_id, [])] for _ in range(n_img)]
        self.done = [False] * n_img

    def step(self, t, ti, tv, last_out, gather_out):
        beam, rows = self.beam, self.n_img * self.beam
        ti, tv = ti.tolist(), tv.tolist()
        any_live = False
        for i in range(self.n_img):
            base = i * beam
            if self.done[i]:
                for k in range(beam):
                    gather_out[base + k] = base + k + rows
                continue
            tmp = []                      # (score, last, words, src_row, was_stepped)
            all_ended = True
            for k, (score, lw, words) in enumerate(self.cands[i]):
                row = base + k
                if t > 0 and lw == self.eos:
                    tmp.append((score, lw, words, row, False))
                    continue
                all_ended = False
                for j in range(beam):
                    w = ti[row][j]
                    tmp.append((score + tv[row][j], w, words + [w], row, True))
            tmp = sorted(tmp, key=lambda x: x[0], reverse=True)[:beam]   # stable, as the reference
            self.cands[i] = [(s, lw, words) for (s, lw, words, _, _) in tmp]
            for k, (_, lw, _, src, st) in enumerate(tmp):
                gather_out[base + k] = src if st else src + rows
                last_out[base + k] = lw
            for k in range(len(tmp), beam):          # t == 0 with beam > candidates never happens
                gather_out[base + k] = base + k + rows
            if all_ended:
                self.done[i] = True
            else:
                any_live = True
        return any_live

    def result(self):
        return [[(s, words) for (s, _, words) in c] for c in self.cands]


class _VectorMerge:
    """The same bookkeeping for many images at once in numpy: fp64 score sums, candidates laid out in insertion
    order (parents in rank order, children by rank) and a STABLE descending sort, so ties resolve exactly as the
    reference's `sorted(..., reverse=True)`.  64 images cost one set of array operations instead of 64 Python loops."""

    def __init__(self, n_img, beam, T, sos_id, eos_id):
        self.n, self.beam, self.eos = n_img, beam, eos_id
        self.scores = np.full((n_img, beam), -np.inf)
        self.scores[:, 0] = 0.0
        self.ncand = 1                                      # candidates per image: 1 at t = 0, then `beam`
        self.last = np.full((n_img, beam), sos_id, dtype=np.int64)
        self.words = np.zeros((n_img, beam, T), dtype=np.int64)
        self.length = np.zeros((n_img, beam), dtype=np.int64)
        self.done = np.zeros(n_img, dtype=bool)
        self.img = np.arange(n_img)[:, None]

    def step(self, t, ti, tv, last_out, gather_out):
        n, beam = self.n, self.beam
        rows = n * beam
        valid = np.arange(beam)[None, :] < self.ncand                      # [1,beam]
        ended = (self.last == self.eos) & (t > 0) & valid                  # carried candidates
        live = valid & ~ended
        all_ended = ~live.any(axis=1)
        newly_done = all_ended & ~self.done
        active = ~self.done & ~all_ended
        # candidate table [n, beam(parent), beam(child)] in insertion order
        cand = np.full((n, beam, beam), -np.inf)
        child = self.scores[:, :, None] + tv.reshape(n, beam, beam).astype(np.float64)
        cand[live] = child[live]
        cand[ended, 0] = self.scores[ended]
        flat = cand.reshape(n, beam * beam)
        order = np.argsort(-flat, axis=1, kind='stable')[:, :beam]
        pk, cj = order // beam, order % beam
        new_scores = np.take_along_axis(flat, order, axis=1)
        carried = ended[self.img, pk]
        tok = ti.reshape(n, beam, beam)[self.img, pk, cj]
        new_last = np.where(carried, self.last[self.img, pk], tok)
        new_words = self.words[self.img, pk]
        new_len = self.length[self.img, pk]
        add = ~carried
        ii, kk = np.nonzero(add & active[:, None])
        new_words[ii, kk, new_len[ii, kk]] = tok[ii, kk]
        new_len = new_len + add
        upd = active                                                        # frozen images keep everything
        self.scores[upd] = new_scores[upd]
        self.last[upd] = new_last[upd]
        self.words[upd] = new_words[upd]
        self.length[upd] = new_len[upd]
        self.done |= newly_done
        base = (np.arange(n) * beam)[:, None]
        src = base + pk
        gat = np.where(carried, src + rows, src)
        ident = base + np.arange(beam)[None, :] + rows
        gather_out[:] = np.where(upd[:, None], gat, ident).reshape(rows)
        last_out[:] = self.last.reshape(rows)
        if t == 0:
            self.ncand = beam
        return bool(active.any())

    def result(self):
        out = []
        for i in range(self.n):
            k = self.ncand
            out.append([(float(self.scores[i, j]), self.words[i, j, :self.length[i, j]].tolist()) for j in range(k)])
        return out


def beam_search_batch(cap, fc_feats, att_feats, senti_words, senti_labels, beam, decoding_constraint, T, graphs=True):
    if (graphs and cap.__dict__.get('_beam_graphs') is not None and getattr(cap, 'beam_device_merge', True) and beam <= 8
            and ops.TIMER.arm_step is None and ops.graphs_allowed_here()):
        return _graphed_search(cap, fc_feats, att_feats, senti_words, senti_labels, beam, decoding_constraint, T)
    return _search(cap, fc_feats, att_feats, senti_words, senti_labels, beam, decoding_constraint, T)


def _search(cap, fc_feats, att_feats, senti_words, senti_labels, beam, decoding_constraint, T):
    s = _Search(cap, fc_feats, att_feats, senti_words, senti_labels, beam, decoding_constraint, T)
    if s.device_merge:
        with ops.h3_weights_scope(cap._dev):
            for t in range(T):
                s.step(t)
                if ((t & 3) == 3 or t == T - 1) and s.all_done(t):     # the one host read, every fourth step
                    break
        return s.finish()
    return s.run_host_merge()


CHUNK = 4          # decode steps per captured graph = steps between two looks at the live-image counter


FEW_ROW_PLANE_BYTES = 24 << 20


def _few_rows(cap, n_img, beam):
    """One image's beam on at most ROWS_STEP_MAX rows: the few-row step (csrc/rows.hip)."""
    return (n_img * beam <= cap.ROWS_STEP_MAX and beam <= 8 and getattr(cap, 'rows_step', True)
            and getattr(cap, 'beam_device_merge', True) and cap._rows_vocab_ok())


def _graph_key(cap, ins, beam, decoding_constraint, T):
    # (the cached tables a captured graph points at are functions of these weights: token table, sentiment-word tables,
    # and - gated scan - the sentiment-word table through attention.senti2att)
    # ... and - few-row searches - the prologue's weight planes, which such a graph keeps instead of re-splitting them
    versions = tuple(q._version for q in (cap.word_embed[0].weight, cap.att_lstm.weight_ih, cap.senti2att[0].weight,
                                          cap.senti2att[0].bias, cap.attention.senti2att.weight, cap.att_embed[0].weight,
                                          cap.att2att[0].weight, cap.attention.cont2att.weight))
    return (tuple(None if x is None else (tuple(x.shape), x.dtype) for x in ins), beam, decoding_constraint, T, versions,
            ops.WEIGHT_EPOCH, cap.eos_id, torch.cuda.current_device(), bool(getattr(cap, 'beam_step_gate', True)))


def _replay(cap, entry, ins, T):
    graphs, static, search = entry[:3]
    ops.stage_inputs(static, ins)           # inputs -> the graphs' static buffers, one launch
    for g, t1 in graphs:
        g.replay()
        cap.last_beam_steps = t1
        if t1 < T and search.all_done(t1 - 1):
            break
    return search.finish()


def replay_if_captured(cap, fc_feats, att_feats, senti_words, senti_labels, beam, decoding_constraint, T):
    """Captioner.sample_batch's short cut: a search whose graphs exist is one key look-up, the input copies, the replay
    and the read-back - none of the per-call set-up of the eager path (parameter dictionary, weights scope, mode walk:
    ~100 us of host time in front of a 1 ms search).  None when there is nothing to replay."""
    cache = cap.__dict__.get('_beam_graphs')
    if (not cache or not getattr(cap, 'beam_device_merge', True) or beam > 8 or ops.TIMER.arm_step is not None
            or not ops.graphs_allowed_here()):
        return None
    if not (fc_feats.is_cuda and att_feats.is_cuda):
        return None
    ins = [cap._f32(fc_feats), cap._f32(att_feats), senti_words, senti_labels]
    key = _graph_key(cap, ins, beam, decoding_constraint, T)
    entry = cache.get(key)
    if not isinstance(entry, tuple):
        return None
    cache[key] = cache.pop(key)             # LRU order
    return _replay(cap, entry, ins, T)


def _graphed_search(cap, fc_feats, att_feats, senti_words, senti_labels, beam, decoding_constraint, T):
    """The search served from captured HIP graphs (Captioner.enable_beam_graphs).  A single-image beam-5 step is
    ~14 small launches behind four FFI calls and the host needs longer to enqueue them than the device to run them;
    the loop has no host read except the live-image counter every CHUNK steps, so the prologue + steps 0..3 capture
    into one graph and every further CHUNK steps into another (one pool, buffers shared): a search costs one input copy,
    <= ceil(T / CHUNK) graph launches with one counter read between them, and the read-back."""
    dev = cap._dev
    ins = [cap._f32(fc_feats), cap._f32(att_feats), senti_words, senti_labels]
    key = _graph_key(cap, ins, beam, decoding_constraint, T)
    cache = cap._beam_graphs
    entry = cache.get(key)
    if entry is None:                       # first sight: run eagerly (builds the cached tables, warms the kernels)
        while len(cache) >= cap._beam_graphs_max:
            cap._graph_evicted(cache.pop(next(iter(cache))), 'beam')    # least recently used
            if cap._beam_graphs is None:
                return _search(cap, *ins, beam, decoding_constraint, T)
        cache[key] = 'seen'
        return _search(cap, *ins, beam, decoding_constraint, T)
    cache[key] = cache.pop(key)             # LRU order
    if entry == 'seen':
        static = [None if x is None else x.clone() for x in ins]
        ws, wp = cap._graph_buffers()          # ONE workspace / plane-buffer pair for all graphs of this captioner
        few = _few_rows(cap, ins[0].shape[0], beam)
        if few:
            # ... except the planes of a few-row search: its graph is ONE replay of ~1 ms, and the three split launches
            # of its prologue's weights (att_embed, att2att / senti2att, the gate projections: 8 MB of planes) are 2 % of
            # it.  They are built once, in front of the capture, into a buffer of this graph's own (the shared one is
            # re-planned by every capture), and the graph key holds those weights' versions.
            wp = torch.empty(FEW_ROW_PLANE_BYTES, dtype=torch.uint8, device=dev)
        stream = cap.__dict__.get('_beam_stream')        # one capture stream per captioner, held for itself
        if stream is None or stream.device != dev:
            stream = cap.__dict__['_beam_stream'] = ops.private_stream(dev)
        pool = torch.cuda.graph_pool_handle()
        graphs = []
        # (few-row searches are ONE graph: its last node copies the results into this pinned buffer)
        pinned = torch.empty(_Search.result_bytes(ins[0].shape[0], beam, T), dtype=torch.uint8).pin_memory() if few else None
        torch.cuda.synchronize()
        with ops.capture_buffers(ws, wp), torch.cuda.stream(stream):
            scope = ops.h3_weights_scope(dev)      # ONE scope over all the captures (its planes live in `wp`,
            scope.__enter__()                      # are split inside graph 0 and read by the later graphs)
            try:
                if few:                            # (the eager prologue leaves the planes in the scope)
                    _Search(cap, *static, beam, decoding_constraint, T)
                search, t0 = None, 0
                while t0 < T:
                    g = torch.cuda.CUDAGraph()
                    with ops.graph_capture(g, pool=pool, stream=stream):
                        if search is None:
                            search = _Search(cap, *static, beam, decoding_constraint, T)
                        # few-row searches end themselves on the device (isc_rows_ext.live_in): ONE graph, no counter
                        # read in between; the general kernels run CHUNK steps per graph with a look at the counter after
                        # each - their steps are gated as well (isc_set_stream_gate), so the up to CHUNK - 1 steps a graph
                        # runs past the end of the search cost a third of a step each.  (ONE gated graph for the whole
                        # search was measured: the same 3.4 ms for 64 images x 20 steps - the looks at the counter hide
                        # behind the device's queue - and slower for searches that end early, whose every remaining step
                        # still costs its launches: tools/beam64_probe.py)
                        t1 = T if search.rows_mode else min(t0 + CHUNK, T)
                        for t in range(t0, t1):
                            search.step(t)
                        if search.rows_mode and t1 == T and pinned is not None:
                            search.stage_result(pinned)
                    graphs.append((g, t1))
                    t0 = t1
            finally:
                scope.__exit__(None, None, None)
        entry = cache[key] = (graphs, static, search, ws, wp, pool, pinned)
    return _replay(cap, entry, ins, T)


class _Search:
    """State of one batched search: the prologue, the per-row copies, the device-side candidate table; step(t)
    enqueues decode step t (preceded by the state re-ordering that step t - 1 decided)."""

    def __init__(self, cap, fc_feats, att_feats, senti_words, senti_labels, beam, decoding_constraint, T):
        p = cap._p()
        n_img = fc_feats.shape[0]
        # few rows (one image's beam): the image's sentiment-word features as per-image tensors - the vocabulary-sized
        # tables would put an id -> row index chain in front of every step's sentiment scan, and there is one image
        few = _few_rows(cap, fc_feats.shape[0], beam)
        P = cap._prologue(p, 'beam', fc_feats, att_feats, None, senti_words,
                          senti_labels if senti_words is not None else None, want_table='build',
                          words_table=getattr(cap, 'words_table', True) and not few, gate_rows=fc_feats.shape[0] * beam)
        dev = cap._dev
        H, Wd, V = cap.att_lstm.hidden_size, cap.settings['word_emb_dim'], cap.vocab_size
        rows = n_img * beam
        self.device_merge = getattr(cap, 'beam_device_merge', True) and beam <= 8
        # <= 8 rows (one image's beam): the few-row step (csrc/rows.hip) - state re-ordering as an index on its loads,
        # per-tile candidates out of the classifier, top-k + merge in one launch (isc_beam_select): six launches per step.
        # The rows of an image share its step-invariant tensors there (isc_rows_ext.row_div): nothing is expanded.
        self.rows_mode = self.device_merge and cap._rows_step_ok(rows, P)
        # the general kernels under the device-side merge: a step past the search's end skips its heavy launches on the
        # device (ops.stream_gate; `beam_step_gate = False` on the captioner restores the ungated steps for A/B runs)
        self.gated = self.device_merge and not self.rows_mode and getattr(cap, 'beam_step_gate', True)
        if self.rows_mode:
            Pb = P
        else:
            # expand the step-invariant tensors to one copy per beam row (row = img*beam + k)
            rep = torch.arange(n_img, device=dev).repeat_interleave(beam)

            def expand(x):
                return None if x is None else x.index_select(0, rep).contiguous()
            Pb = type(P)()
            Pb.B, Pb.R, Pb.Mw = rows, P.R, P.Mw
            for name in ('fc_e', 'att_e3', 'att_p3', 'words_e3', 'words_p3', 'label_e', 'label_w', 'pre1'):
                if P.words_ids is not None and name in ('words_e3', 'words_p3'):
                    setattr(Pb, name, getattr(P, name))           # shared [V,.] tables: only the ids are per row
                else:
                    setattr(Pb, name, expand(getattr(P, name)))
            Pb.words_ids = expand(P.words_ids)
            Pb.tab = P.tab
            # gated scan (few rows): per-region projections follow the regions, the word table is shared
            gc, gs = getattr(P, 'gate_Gc', None), getattr(P, 'gate_Gs', None)
            if gc is not None:
                Pb.gate_Gc = expand(gc.view(n_img, P.R, -1)).view(rows * P.R, -1)
                Pb.gate_Gs = gs if P.words_ids is not None else expand(gs.view(n_img, P.Mw, -1)).view(rows * P.Mw, -1)
        self.cap, self.p, self.Pb = cap, p, Pb
        self.n_img, self.beam, self.T, self.rows, self.dc = n_img, beam, T, rows, decoding_constraint
        self.stats_tile = ops.rows_stats_tile(V) if self.rows_mode else 128
        self.ws = cap._alloc_step_ws(rows, Pb, self.stats_tile)
        # Everything that starts at zero lives in ONE zeroed arena (one fill launch instead of a dozen): the recurrent
        # state [h|c, layer, row, H], fp64 scores, word lists, lengths, the done flags and the live-image counters.
        def carve(arena, off, dtype, *shape):
            n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
            return arena[off:off + n].view(dtype).view(*shape), (off + n + 15) & ~15
        need = 4 * (2 * 2 * rows * H) + 2 * 8 * rows + 2 * 8 * rows * T + 2 * 4 * rows + 4 * n_img + 4 * (T + 1) + 16 * 8
        arena = torch.zeros(need, dtype=torch.uint8, device=dev)
        off = 0
        # recurrent state as ONE tensor [h|c, layer, row, H] per buffer: the per-step beam re-ordering is then a
        # single gather over [next ; current] rows instead of four index_selects and two wheres
        self.st_cur, off = carve(arena, off, torch.float32, 2, 2, rows, H)
        self.st_nxt = cap._new(2, 2, rows, H)
        self.logits = None if self.rows_mode else cap._new(rows, V)
        self.xt = cap._new(rows, Wd)
        # top-k ids and values share one byte buffer: ONE device->host copy per step (ids first: 8-byte aligned)
        nk = self.nk = rows * beam
        self.top_buf = torch.empty(nk * 12, dtype=torch.uint8, device=dev)
        self.top_idx = self.top_buf[:nk * 8].view(torch.int64).view(rows, beam)
        self.top_val = self.top_buf[nk * 8:].view(torch.float32).view(rows, beam)
        self.emb = p['word_embed.0.weight']
        self.mask_special = cap.pad_id != cap.eos_id
        cap.last_beam_steps = 0                    # decode steps executed (bench.py: latency per step)
        if self.device_merge:
            from ._lib import BeamMergeArgs, BeamSelectArgs, RowsExt
            span0 = off                                 # [scores | words | lengths | done | live]: ONE read-back at the end
            spans = {}

            def carve_host(name, dtype, npdtype, *shape):
                nonlocal off
                t, nxt_off = carve(arena, off, dtype, *shape)
                spans[name] = (off - span0, npdtype, shape)
                off = nxt_off
                return t
            sc2 = carve_host('score', torch.float64, np.float64, 2, rows)
            wd2 = carve_host('words', torch.int64, np.int64, 2, rows, T)
            ln2 = carve_host('length', torch.int32, np.int32, 2, rows)
            self.done = carve_host('done', torch.int32, np.int32, n_img)
            self.live = carve_host('live', torch.int32, np.int32, T + 1)
            self._result_span, self._result_views = arena[span0:off], spans
            self._result_host = None                      # pinned image of the span, filled by the graph itself (stage_result)
            self.score, self.words, self.length = [sc2[0], sc2[1]], [wd2[0], wd2[1]], [ln2[0], ln2[1]]
            la2 = torch.full((2, rows), cap.sos_id, dtype=torch.int64, device=dev)
            self.last = [la2[0], la2[1]]
            self.gather = torch.empty(rows, dtype=torch.int64, device=dev)
            a = self.args = BeamMergeArgs()
            a.n_img, a.beam, a.T, a.eos_id = n_img, beam, T, cap.eos_id
            a.top_val, a.top_idx = self.top_val.data_ptr(), self.top_idx.data_ptr()
            a.done, a.gather, a.live = self.done.data_ptr(), self.gather.data_ptr(), self.live.data_ptr()
            self.cur = 0
            self._plans = [None, None]                     # (step plan, merge arguments) of the even / odd steps
            if self.rows_mode:
                n_tile = self.ws['pmax'].shape[1]
                self.cand_val = cap._new(rows, n_tile, 8)
                self.cand_idx = cap._new(rows, n_tile, 8, dtype=torch.int32)
                self.src_row = self.gather                  # written by step t's select, read by step t + 1
                x = self.ext = RowsExt()
                x.src_row, x.stats_tile, x.beam, x.row_div = None, self.stats_tile, beam, beam   # (step 0: the identity)
                x.cand_val, x.cand_idx = self.cand_val.data_ptr(), self.cand_idx.data_ptr()
                x.pad_id, x.sos_id, x.unk_id = cap.pad_id, cap.sos_id, cap.unk_id
                x.mask_special, x.decoding_constraint = int(self.mask_special), int(decoding_constraint)
                b = self.args = BeamSelectArgs()
                b.n_img, b.beam, b.T, b.eos_id, b.n_tile, b.V = n_img, beam, T, cap.eos_id, n_tile, V
                b.part_max, b.part_sum = self.ws['pmax'].data_ptr(), self.ws['psum'].data_ptr()
                b.cand_val, b.cand_idx = x.cand_val, x.cand_idx
                b.done, b.src_row, b.live = self.done.data_ptr(), self.src_row.data_ptr(), self.live.data_ptr()
                b.top_val, b.top_idx = self.top_val.data_ptr(), self.top_idx.data_ptr()
                # the select re-orders the state as well: step t reads st_cur, writes st_nxt; its select gathers
                # st_nxt -> st_cur (same buffers, same plan, every step)
                b.state_in, b.state_out = self.st_nxt.data_ptr(), self.st_cur.data_ptr()
                b.state_planes, b.H = 4, H
            else:
                self.st_free = torch.empty_like(self.st_cur)   # third state buffer: target of the per-step re-ordering

    def step(self, t):
        """Decode step t on the device-side candidate table: nothing is read back (the reference's early exit,
        captioner.py:379-381, is the caller's look at the live-image counter every fourth step)."""
        if self.rows_mode:
            return self._step_rows(t)
        cap, a = self.cap, self.args
        if t > 0:
            # new state of row r = stepped ? nxt[parent] : cur[parent]  ==  [nxt ; cur][gather[r]]
            ops.beam_gather(self.st_nxt, self.st_cur, self.gather, self.st_free)
            self.st_cur, self.st_free = self.st_free, self.st_cur
        cap.last_beam_steps = t + 1
        cur = self.cur
        last_d = self.last[cur]
        if self.Pb.tab is None:
            ops.embed_relu_fwd(self.emb, last_d, self.xt)
        # Every pointer of a step repeats with period 2 (token / candidate tables ping-pong, the recurrent state
        # alternates between two of its three buffers): the step plan and the merge arguments of steps 0 and 1 are
        # kept and re-used, so that a later step costs four library calls and no struct filling - the eager loop is
        # then bound by the device (~110 us per step), not by ~55 us of host work in front of each step's first launch
        fast = self._plans[t & 1] if t >= 2 and not ops.TIMER.armed else None
        # live[t] == 0 (every image has ended): the step's contractions, scans and top-k return at once on the device
        # (isc_set_stream_gate) - the steps enqueued past the end of the search (up to three: the host looks at the counter
        # every fourth step) cost their launches only.  The small launches around them (state re-order, embedding gather,
        # merge) still run: they copy the frozen rows along.
        live_in = self.live.data_ptr() + 4 * t if (t > 0 and self.gated) else None
        with ops.stream_gate(live_in):
            if fast is not None:
                ops.step_fwd(fast[0])
            else:
                h_cur, c_cur, h_nxt, c_nxt = self.st_cur[0], self.st_cur[1], self.st_nxt[0], self.st_nxt[1]
                cap._step(self.p, self.Pb, self.ws, self.xt, h_cur, c_cur, h_nxt, c_nxt, logits=self.logits, tok=last_d)
            ops.beam_topk(self.logits, self.ws['pmax'], self.ws['psum'], last_d, self.beam, cap.pad_id, cap.sos_id,
                          cap.unk_id, self.mask_special, self.dc, self.top_val, self.top_idx)
        nxt = cur ^ 1
        if fast is not None:
            a = fast[1]
            a.t = t
        else:
            a.t = t
            a.score_in, a.score_out = self.score[cur].data_ptr(), self.score[nxt].data_ptr()
            a.last_in, a.last_out = self.last[cur].data_ptr(), self.last[nxt].data_ptr()
            a.words_in, a.words_out = self.words[cur].data_ptr(), self.words[nxt].data_ptr()
            a.len_in, a.len_out = self.length[cur].data_ptr(), self.length[nxt].data_ptr()
            if t < 2 and not ops.TIMER.armed and '_plan' in self.ws:
                plan = self.ws['_plan']
                self._plans[t] = (type(plan).from_buffer_copy(plan), type(a).from_buffer_copy(a))
        ops.beam_merge(a)
        self.cur = nxt

    def _step_rows(self, t):
        """Step t on the few-row kernels: the step reads st_cur and writes st_nxt, the select that closes it gathers the
        parents' rows of st_nxt back into st_cur (isc_beam_select_args.state_*)."""
        cap, a, x = self.cap, self.args, self.ext
        cap.last_beam_steps = t + 1
        cur = self.cur
        last_d = self.last[cur]
        if self.Pb.tab is None:
            ops.embed_relu_fwd(self.emb, last_d, self.xt)
        fast = self._plans[t & 1] if t >= 2 else None
        live_in = self.live.data_ptr() + 4 * t if t > 0 else None     # live[t] == 0: the step's launches return at once
        if fast is not None:
            plan, x, a = fast
            x.live_in = a.live_in = live_in
            ops.rows_step_fwd(plan, x)
        else:
            x.last_word = last_d.data_ptr()
            x.live_in = a.live_in = live_in
            sc, sn = self.st_cur, self.st_nxt
            cap._step(self.p, self.Pb, self.ws, self.xt, sc[0], sc[1], sn[0], sn[1], logits=None, tok=last_d, rows_ext=x)
        nxt = cur ^ 1
        a.t = t
        if fast is None:
            a.score_in, a.score_out = self.score[cur].data_ptr(), self.score[nxt].data_ptr()
            a.last_in, a.last_out = self.last[cur].data_ptr(), self.last[nxt].data_ptr()
            a.words_in, a.words_out = self.words[cur].data_ptr(), self.words[nxt].data_ptr()
            a.len_in, a.len_out = self.length[cur].data_ptr(), self.length[nxt].data_ptr()
            if t < 2 and '_plan' in self.ws:
                plan = self.ws['_plan']
                self._plans[t & 1] = (type(plan).from_buffer_copy(plan), type(x).from_buffer_copy(x), type(a).from_buffer_copy(a))
        ops.beam_select(a)
        self.cur = nxt

    def all_done(self, t):
        return int(self.live[t + 1].item()) == 0

    @staticmethod
    def result_bytes(n_img, beam, T):
        """Upper bound of the result span (scores, words, lengths, done flags, live counters + alignment)."""
        rows = n_img * beam
        return 2 * 8 * rows + 2 * 8 * rows * T + 2 * 4 * rows + 4 * n_img + 4 * (T + 1) + 16 * 8

    def stage_result(self, pinned):
        """Inside a capture, behind the last step: the result span -> `pinned` (a pinned host buffer allocated before the
        capture opened) as a copy node of the graph.  A replayed search then costs one stream synchronise and no copy
        call on the host (the pageable read-back was ~20 us of a 1 ms one-image search: tools/beam64_probe.py)."""
        n = self._result_span.numel()
        if pinned is None or pinned.numel() < n:
            return
        self._result_host = pinned[:n]
        self._result_host.copy_(self._result_span, non_blocking=True)

    def finish(self):
        cap, n_img, beam, T = self.cap, self.n_img, self.beam, self.T
        cap.cont_weights, cap.senti_weights, cap.cont_senti_weights = [], [], []
        # executed steps as the reference counts them: up to and including the step after which nobody was live
        steps = cap.last_beam_steps
        if self._result_host is not None:                # replayed from a graph that ends in the copy (stage_result)
            torch.cuda.current_stream().synchronize()
            hb = self._result_host.numpy()
        else:
            hb = self._result_span.cpu().numpy()         # one device->host copy: scores, words, lengths, counters

        def host(name):
            o, dt, shape = self._result_views[name]
            n = int(np.prod(shape)) * np.dtype(dt).itemsize
            return hb[o:o + n].view(dt).reshape(shape)
        lv = host('live')
        executed = steps
        for t in range(steps):
            if lv[t + 1] == 0:
                executed = t + 1
                break
        # the tables of the last step that WROTE them (step t writes buffer (t + 1) & 1): every enqueued step on the general
        # kernels (a frozen image's rows are copied along), the last executed one on the few-row kernels (later launches
        # return at once: isc_rows_ext.live_in)
        cur = (executed if self.rows_mode else steps) & 1
        # (plain lists once, then list slicing: 64 images x 5 beams as 320 numpy slices + generator joins took 0.55 ms
        # behind a 2.85 ms search - tools/beam64_probe.py)
        sc = host('score')[cur].reshape(n_img, beam).tolist()
        wd = host('words')[cur].reshape(n_img, beam, T).tolist()
        ln = host('length')[cur].reshape(n_img, beam).tolist()
        cap.last_beam_steps = executed
        i2w, eos = cap.idx2word, cap.eos_id
        captions, scores, ids = [], [], []
        for wd_i, ln_i, sc_i in zip(wd, ln, sc):
            cand = [words_k[:n_k] for words_k, n_k in zip(wd_i, ln_i)]
            captions.append([' '.join([i2w[w] for w in words_k if w != eos]) for words_k in cand])
            scores.append(sc_i)
            ids.append(cand)
        return captions, scores, ids

    def run_host_merge(self):
        cap, p, Pb, ws, T = self.cap, self.p, self.Pb, self.ws, self.T
        n_img, beam, rows, nk, dev = self.n_img, self.beam, self.rows, self.nk, cap._dev
        st_cur, st_nxt, logits, xt, emb, top_buf = self.st_cur, self.st_nxt, self.logits, self.xt, self.emb, self.top_buf
        merge = _VectorMerge(n_img, beam, T, cap.sos_id, cap.eos_id) if n_img >= 4 else \
            _ListMerge(n_img, beam, cap.sos_id, cap.eos_id)
        ctrl_h = torch.empty(2, rows, dtype=torch.int64).pin_memory()     # [last word ; gather index], one upload per step
        ctrl_np = ctrl_h.numpy()
        ctrl_np[0, :] = cap.sos_id
        ctrl_d = ctrl_h.to(dev, non_blocking=True)
        with ops.h3_weights_scope(dev):               # frozen weights: their f16 planes are built once per search
            for t in range(T):
                cap.last_beam_steps = t + 1
                last_d = ctrl_d[0]
                h_cur, c_cur, h_nxt, c_nxt = st_cur[0], st_cur[1], st_nxt[0], st_nxt[1]
                if Pb.tab is None:
                    ops.embed_relu_fwd(emb, last_d, xt)
                cap._step(p, Pb, ws, xt, h_cur, c_cur, h_nxt, c_nxt, logits=logits, tok=last_d)
                ops.beam_topk(logits, ws['pmax'], ws['psum'], last_d, beam, cap.pad_id, cap.sos_id, cap.unk_id,
                              self.mask_special, self.dc, self.top_val, self.top_idx)
                hb = top_buf.cpu().numpy()        # the single host read of this step
                ti = hb[:nk * 8].view(np.int64).reshape(rows, beam)
                tv = hb[nk * 8:].view(np.float32).reshape(rows, beam)
                # new state of row r = stepped ? nxt[parent] : cur[parent]  ==  [nxt ; cur][parent + (stepped ? 0 : rows)]
                any_live = merge.step(t, ti, tv, ctrl_np[0], ctrl_np[1])
                if not any_live:
                    break
                ctrl_d = ctrl_h.to(dev, non_blocking=True)
                st_cur = torch.cat([st_nxt, st_cur], dim=2).index_select(2, ctrl_d[1])
        cap.cont_weights, cap.senti_weights, cap.cont_senti_weights = [], [], []
        captions, scores, ids = [], [], []
        for i, cands in enumerate(merge.result()):
            captions.append([' '.join(cap.idx2word[w] for w in words if w != cap.eos_id) for (_, words) in cands])
            scores.append([s for (s, _) in cands])
            ids.append([list(words) for (_, words) in cands])
        return captions, scores, ids
