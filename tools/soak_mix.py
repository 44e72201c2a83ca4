#!/usr/bin/env python3
"""Soak 2: graph-served RL training iterations interleaved with what a trainer does between them - an eager XE step, a greedy
evaluation roll-out (its own HIP graphs), a one-image beam search, a checkpoint round trip - on ONE captioner.
    python tools/soak_mix.py [rounds]"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Detector, synth, ops
from insenticap_model_amd.train import xe_train_step

dev = torch.device('cuda:0')
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
V, T, R = bench.V, bench.T, bench.R
st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
det = Detector(synth.make_idx2word(V), T, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
det.to(dev)
cap = det.captioner
B = 64
batches, split = synth.make_rl_batches(2, B, V, st, grid=(6, 6), seq_len=T, seed=90)
det.set_ciderd_scorer(split)
tt = torch.from_numpy
facts = [[(b[0], tt(b[1]).to(dev), tt(b[2]).to(dev), (tt(b[3][0]).to(dev), b[3][1]), tt(b[4]).to(dev), tt(b[5]).to(dev),
           b[6])] for b in batches]
s = synth.make_inputs(80, V, st, regions=R, seq_len=T, seed=91)
scs = [((tt(s['captions']).to(dev), s['lengths']), tt(s['cpt_words']).to(dev), tt(s['senti_words']).to(dev),
        tt(s['senti_labels']).to(dev))]
d = synth.make_inputs(32, V, st, regions=R, seq_len=T, seed=500)
fact = (None, tt(d['fc_feats']).to(dev), tt(d['att_feats']).to(dev), (tt(d['captions']).to(dev), d['lengths']), tt(d['cpt_words']).to(dev))
labels = tt(d['senti_labels']).to(dev)
ev, _ = bench.device_inputs(16, 33, dev)
t0 = time.perf_counter()
for r in range(rounds):
    for i in range(4):
        out = det((facts[i % 2], scs), 'fact', True)
        assert all(v == v for v in out.values()), out
    out = det(([(facts[r % 2][0][0], facts[r % 2][0][1], facts[r % 2][0][2], facts[r % 2][0][4], facts[r % 2][0][5],
                 torch.zeros(B, dtype=torch.int64, device=dev))], scs), 'senti', True)      # a 'senti' iteration: same graph object
    assert all(v == v for v in out.values()), out
    if r % 3 == 0:
        l = xe_train_step(cap, det.cap_optim, det.cap_xe_crit, det.cap_da_crit, fact, labels, scs[0], 0.25, 0.1)
        assert float(l['all_loss']) == float(l['all_loss'])
        cap.cpt_feats = cap.fc_feats = None
    cap.eval()
    with torch.no_grad():
        seq, lp, mk = cap(*ev, T, 1, mode='rl')
        caps_, scores = cap.sample(ev[0][r % 16], ev[1][r % 16], ev[3][r % 16], ev[4][r % 16:r % 16 + 1], 5, 1, T)
    assert bool(torch.isfinite(lp).all()) and len(caps_) == 5
    cap.train()
    if r % 10 == 5:
        buf = io.BytesIO()
        torch.save(cap.state_dict(), buf)
        buf.seek(0)
        cap.load_state_dict(torch.load(buf))
    if r % 10 == 0:
        torch.cuda.synchronize()
        print('round %d  %.1f s  allocated %.1f MB  rl graph: %d captures, %d replays, %d eager' % (
            r, time.perf_counter() - t0, torch.cuda.memory_allocated() / 1e6, det._rl_graph.captures, det._rl_graph.replays,
            det._rl_graph.eager_steps), flush=True)
torch.cuda.synchronize()
ops.check_numerics('soak')
print('soak_mix ok: %d rounds, %d private streams held' % (rounds, len(ops._OWNED_STREAMS)))
