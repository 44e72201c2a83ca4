#!/usr/bin/env python3
"""Soak: many graph-served RL iterations and XE iterations in one process; device memory and the private-stream pool must
not grow, every loss stays finite.   python tools/soak.py [rl_iterations [xe_iterations]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, Detector, synth, ops
from insenticap_model_amd.train_graph import XETrainGraph

dev = torch.device('cuda:0')
n_rl = int(sys.argv[1]) if len(sys.argv) > 1 else 600
n_xe = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
V, T, R = bench.V, bench.T, bench.R
st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
det = Detector(synth.make_idx2word(V), T, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
det.to(dev)
B = 128
batches, split = synth.make_rl_batches(3, B, V, st, grid=(6, 6), seq_len=T, seed=90)
det.set_ciderd_scorer(split)
tt = torch.from_numpy
facts = [[(b[0], tt(b[1]).to(dev), tt(b[2]).to(dev), (tt(b[3][0]).to(dev), b[3][1]), tt(b[4]).to(dev), tt(b[5]).to(dev),
           b[6])] for b in batches]
s = synth.make_inputs(80, V, st, regions=R, seq_len=T, seed=91)
scs = [((tt(s['captions']).to(dev), s['lengths']), tt(s['cpt_words']).to(dev), tt(s['senti_words']).to(dev),
        tt(s['senti_labels']).to(dev))]
mem0 = None
t0 = time.perf_counter()
for i in range(n_rl):
    out = det((facts[i % 3], scs), 'fact', True)
    assert all(v == v and abs(v) < 1e30 for v in out.values()), (i, out)
    if i == 20:
        torch.cuda.synchronize(); mem0 = torch.cuda.memory_allocated(); streams0 = len(ops._OWNED_STREAMS)
    if i % 100 == 0:
        print('rl iteration %d  %.1f s  allocated %.1f MB' % (i, time.perf_counter() - t0, torch.cuda.memory_allocated() / 1e6), flush=True)
torch.cuda.synchronize()
mem1 = torch.cuda.memory_allocated()
print('RL: %d iterations, %.2f ms each; allocated %.1f -> %.1f MB; private streams %d -> %d; replays %d' % (
    n_rl, (time.perf_counter() - t0) / n_rl * 1e3, mem0 / 1e6, mem1 / 1e6, streams0, len(ops._OWNED_STREAMS), det._rl_graph.replays))
assert mem1 <= mem0 * 1.02 + 64e6, 'device memory grew during the RL soak'
cap = det.captioner
optim, xc, dc = det.cap_optim, det.cap_xe_crit, det.cap_da_crit
d = synth.make_inputs(B, V, st, regions=R, seq_len=T, seed=500)
fact = (None, tt(d['fc_feats']).to(dev), tt(d['att_feats']).to(dev), (tt(d['captions']).to(dev), d['lengths']), tt(d['cpt_words']).to(dev))
labels = tt(d['senti_labels']).to(dev)
g = XETrainGraph(cap, optim, xc, dc, grad_clip=0.1, warmup=2)
t0 = time.perf_counter()
for i in range(n_xe):
    out = g.step(fact, labels, scs[0], 0.25)
    if i == 20:
        torch.cuda.synchronize(); mem0 = torch.cuda.memory_allocated()
    if i % 500 == 0:
        print('xe iteration %d  loss %.4f' % (i, float(out['xe_loss'])), flush=True)
torch.cuda.synchronize()
v = float(out['all_loss'])
print('XE: %d iterations, %.2f ms each; allocated %.1f -> %.1f MB; last loss %.4f' % (
    n_xe, (time.perf_counter() - t0) / n_xe * 1e3, mem0 / 1e6, torch.cuda.memory_allocated() / 1e6, v))
assert v == v and torch.cuda.memory_allocated() <= mem0 * 1.02 + 64e6
# and back: the detector's RL graph takes the captioner over again (its graphs were dropped: two eager iterations, a capture)
c0 = det._rl_graph.captures
for i in range(30):
    out = det((facts[i % 3], scs), 'fact', True)
    assert all(v == v for v in out.values())
for i in range(30):
    g.step(fact, labels, scs[0], 0.25)
torch.cuda.synchronize()
print('took turns: RL captures %d -> %d, XE captures %d' % (c0, det._rl_graph.captures, g.captures))
print('soak ok')
