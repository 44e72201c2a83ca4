#!/usr/bin/env python3
"""XETrainGraph against the eager xe_train_step: same parameters after N steps (eval-mode dropout), then timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth
from insenticap_model_amd.train import xe_train_step
from insenticap_model_amd.train_graph import XETrainGraph
dev = torch.device('cuda:0')
V, R, T = bench.V, bench.R, bench.T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W = synth.make_weights(V, synth.DEFAULT_SETTINGS)


def make():
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    return cap.to(dev)


def batch(seed):
    d = synth.make_inputs(B, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=seed)
    s = synth.make_inputs(80, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=seed + 1000)
    tt = lambda x: torch.from_numpy(x).to(dev)
    fact = (None, tt(d['fc_feats']), tt(d['att_feats']), (tt(d['captions']), d['lengths']), tt(d['cpt_words']))
    scs = ((tt(s['captions']), s['lengths']), tt(s['cpt_words']), tt(s['senti_words']), tt(s['senti_labels']))
    return fact, tt(d['senti_labels']), scs

batches = [batch(10 + i) for i in range(3)]
N = 7
a, b = make().eval(), make().eval()
oa, xc, dc = a.get_optim_criterion(4e-4)
ob, xc2, dc2 = b.get_optim_criterion(4e-4)
g = XETrainGraph(b, ob, xc2, dc2, grad_clip=0.1)
LA = []
for i in range(N):
    f, l, s = batches[i % 3]
    la = xe_train_step(a, oa, xc, dc, f, l, s, 0.0, 0.1)
    LA.append([round(float(la[k]), 6) for k in ('xe_loss', 'da_loss', 'seq2seq_loss')])
for i in range(N):
    f, l, s = batches[i % 3]
    lb = g.step(f, l, s, 0.0)
    torch.cuda.synchronize()
    print(i, 'eager', LA[i], 'graph', [round(float(lb[k]), 6) for k in ('xe_loss', 'da_loss', 'seq2seq_loss')],
          'replays', g.replays, 'eager', g.eager_steps, 'captures', g.captures)
worst = 0.0
for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
    worst = max(worst, float((p - q).abs().max()))
print('max |param eager - param graph| after %d steps: %g' % (N, worst))
print('adam step counters', float(next(iter(oa.state.values()))['step']), float(next(iter(ob.state.values()))['step']))

# timing, train mode
a.train(); b.train()
f, l, s = batches[0]
for name, fn in (('eager', lambda: xe_train_step(a, oa, xc, dc, f, l, s, 0.0, 0.1)), ('graph', lambda: g.step(f, l, s, 0.0))):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('%s: %.2f ms/iter (host %.2f)' % (name, (t2 - t0) / 10 * 1e3, (t1 - t0) / 10 * 1e3))
print('graph stats: replays', g.replays, 'eager', g.eager_steps, 'captures', g.captures)
