#!/usr/bin/env python3
"""Soak 4: more input geometries than a training-graph object keeps (XETrainGraph: 4, RLTrainGraph: 2) - every step evicts the
least recently used one and its captured graphs; device memory must stay bounded.   python tools/soak_geoms.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth, ops
from insenticap_model_amd.train_graph import XETrainGraph

dev = torch.device('cuda:0')
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
V, R = bench.V, bench.R
st = synth.DEFAULT_SETTINGS
cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st).items()})
cap.to(dev).train()
optim, xc, dc = cap.get_optim_criterion(4e-4)
tt = lambda x: torch.from_numpy(x).to(dev)
def batch(T, seed):
    d = synth.make_inputs(64, V, st, regions=R, seq_len=T, seed=seed)
    s = synth.make_inputs(80, V, st, regions=R, seq_len=T, seed=seed + 1)
    return ((None, tt(d['fc_feats']), tt(d['att_feats']), (tt(d['captions']), d['lengths']), tt(d['cpt_words'])),
            tt(d['senti_labels']), ((tt(s['captions']), s['lengths']), tt(s['cpt_words']), tt(s['senti_words']), tt(s['senti_labels'])))
widths = (10, 12, 14, 16, 18, 20)
batches = {T: batch(T, 100 + T) for T in widths}
g = XETrainGraph(cap, optim, xc, dc, grad_clip=0.1, warmup=1)
mem = []
t0 = time.perf_counter()
for r in range(rounds):
    for T in widths:
        for _ in range(3):                   # eager, capture, one replay - then the next geometry evicts an older one
            out = g.step(*batches[T], 0.0)
    torch.cuda.synchronize()
    mem.append(torch.cuda.memory_allocated() / 1e6)
    print('round %d  %.1f s  allocated %.1f MB reserved %.1f MB  captures %d  geometries kept %d  streams %d' % (
        r, time.perf_counter() - t0, mem[-1], torch.cuda.memory_reserved() / 1e6, g.captures, len(g._geoms), len(ops._OWNED_STREAMS)), flush=True)
assert len(g._geoms) <= 4
assert mem[-1] <= mem[2] * 1.05 + 64, 'device memory keeps growing with evicted geometries: %s' % mem
print('soak_geoms ok')
