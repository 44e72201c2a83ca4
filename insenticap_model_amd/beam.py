"""Batched beam search behind `Captioner.sample` (reference: captioner.py:351-420).

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


def beam_search_batch(cap, fc_feats, att_feats, senti_words, senti_labels, beam, decoding_constraint, T):
    p = cap._p()
    n_img = fc_feats.shape[0]
    P = cap._prologue(p, 'beam', fc_feats, att_feats, None, senti_words,
                      senti_labels if senti_words is not None else None, want_table='cached')
    dev = cap._dev
    H, Wd, V = cap.att_lstm.hidden_size, cap.settings['word_emb_dim'], cap.vocab_size
    rows = n_img * beam
    # expand the step-invariant tensors to one copy per beam row (row = img*beam + k)
    rep = torch.arange(n_img, device=dev).repeat_interleave(beam)

    def expand(x):
        return None if x is None else x.index_select(0, rep).contiguous()
    Pb = type(P)()
    Pb.B, Pb.R, Pb.Mw = rows, P.R, P.Mw
    for name in ('fc_e', 'att_e3', 'att_p3', 'words_e3', 'words_p3', 'label_e', 'label_w', 'pre1'):
        setattr(Pb, name, expand(getattr(P, name)))
    Pb.tab = P.tab
    ws = cap._alloc_step_ws(rows, Pb)
    # recurrent state as ONE tensor [h|c, layer, row, H] per buffer: the per-step beam re-ordering is then a
    # single gather over [next ; current] rows instead of four index_selects and two wheres
    st_cur = cap._zeros(2, 2, rows, H)
    st_nxt = cap._new(2, 2, rows, H)
    logits = cap._new(rows, V)
    xt = cap._new(rows, Wd)
    # top-k ids and values share one byte buffer: ONE device->host copy per step (ids first: 8-byte aligned)
    nk = rows * beam
    top_buf = torch.empty(nk * 12, dtype=torch.uint8, device=dev)
    top_idx = top_buf[:nk * 8].view(torch.int64).view(rows, beam)
    top_val = top_buf[nk * 8:].view(torch.float32).view(rows, beam)
    emb = p['word_embed.0.weight']
    mask_special = cap.pad_id != cap.eos_id

    # host-side candidates per image: (score, last_word, words, ended_state_row)
    cands = [[(0.0, cap.sos_id, [])] for _ in range(n_img)]
    done = [False] * n_img
    last = [cap.sos_id] * rows
    ctrl_h = torch.empty(2, rows, dtype=torch.int64).pin_memory()     # [last word ; gather index], one upload per step
    ctrl_np = ctrl_h.numpy()
    ctrl_np[0, :] = last
    ctrl_d = ctrl_h.to(dev, non_blocking=True)
    cap.last_beam_steps = 0                    # decode steps executed (bench.py: latency per step)
    for t in range(T):
        cap.last_beam_steps = t + 1
        last_d = ctrl_d[0]
        h_cur, c_cur, h_nxt, c_nxt = st_cur[0], st_cur[1], st_nxt[0], st_nxt[1]
        if Pb.tab is None:
            ops.embed_relu_fwd(emb, last_d, xt)
        cap._step(p, Pb, ws, xt, h_cur, c_cur, h_nxt, c_nxt, logits=logits, tok=last_d)
        ops.beam_topk(logits, ws['pmax'], ws['psum'], last_d, beam, cap.pad_id, cap.sos_id, cap.unk_id,
                      mask_special, decoding_constraint, top_val, top_idx)
        hb = top_buf.cpu().numpy()        # the single host read of this step
        ti = hb[:nk * 8].view(np.int64).reshape(rows, beam).tolist()
        tv = hb[nk * 8:].view(np.float32).reshape(rows, beam).tolist()
        parent = list(range(rows))        # source row of every new row (state gather)
        stepped = [True] * rows           # False: carried candidate keeps its old state
        any_live = False
        for i in range(n_img):
            if done[i]:
                for k in range(beam):
                    stepped[i * beam + k] = False
                continue
            tmp = []                      # (score, last, words, src_row, was_stepped)
            all_ended = True
            for k, (score, lw, words) in enumerate(cands[i]):
                row = i * beam + k
                if t > 0 and lw == cap.eos_id:
                    tmp.append((score, lw, words, row, False))
                    continue
                all_ended = False
                for j in range(beam):
                    w = ti[row][j]
                    tmp.append((score + tv[row][j], w, words + [w], row, True))
            tmp = sorted(tmp, key=lambda x: x[0], reverse=True)[:beam]   # stable, as the reference
            cands[i] = [(s, lw, words) for (s, lw, words, _, _) in tmp]
            for k, (_, lw, _, src, st) in enumerate(tmp):
                parent[i * beam + k] = src
                stepped[i * beam + k] = st
                last[i * beam + k] = lw
            for k in range(len(tmp), beam):          # t == 0 with beam > candidates never happens
                stepped[i * beam + k] = False
            if all_ended:
                done[i] = True
            else:
                any_live = True
        if not any_live:
            break
        # new state of row r = stepped ? nxt[parent] : cur[parent]  ==  [nxt ; cur][parent + (stepped ? 0 : rows)]
        ctrl_np[0, :] = last
        ctrl_np[1, :] = [pr if st else pr + rows for pr, st in zip(parent, stepped)]
        ctrl_d = ctrl_h.to(dev, non_blocking=True)
        st_cur = torch.cat([st_nxt, st_cur], dim=2).index_select(2, ctrl_d[1])
    cap.cont_weights, cap.senti_weights, cap.cont_senti_weights = [], [], []
    captions, scores, ids = [], [], []
    for i in range(n_img):
        captions.append([' '.join(cap.idx2word[w] for w in words if w != cap.eos_id)
                         for (_, _, words) in cands[i]])
        scores.append([s for (s, _, _) in cands[i]])
        ids.append([list(words) for (_, _, words) in cands[i]])
    return captions, scores, ids
