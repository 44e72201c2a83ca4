#!/usr/bin/env python3
"""Host enqueue time vs wall time of the XE iteration (B=128 + 80): is the iteration host-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
import cProfile, pstats
dev = torch.device('cuda:0')
from insenticap_model_amd import Captioner, synth, dp
from insenticap_model_amd.train import xe_train_step
V, R, T = bench.V, bench.R, bench.T
cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev).train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
optim, xe_crit, da_crit = cap.get_optim_criterion(4e-4)
d = synth.make_inputs(B, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=500)
s = synth.make_inputs(80, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=600)
tt = lambda x: torch.from_numpy(x).to(dev)
fact = (None, tt(d['fc_feats']), tt(d['att_feats']), (tt(d['captions']), d['lengths']), tt(d['cpt_words']))
labels = tt(d['senti_labels'])
scs = ((tt(s['captions']), s['lengths']), tt(s['cpt_words']), tt(s['senti_words']), tt(s['senti_labels']))
step = lambda: xe_train_step(cap, optim, xe_crit, da_crit, fact, labels, scs, 0.0, 0.1, arena=None)
for _ in range(3):
    step()
torch.cuda.synchronize()
with bench.no_gc():
    for rep in range(3):
        host = []
        t0 = time.perf_counter()
        for _ in range(10):
            a = time.perf_counter(); step(); host.append(time.perf_counter() - a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('wall/iter %.2f ms  host enqueue/iter %.2f ms (min %.2f)  tail sync %.2f ms' % ((t2 - t0) / 10 * 1e3, sum(host) / 10 * 1e3, min(host) * 1e3, (t2 - t1) * 1e3))
    from insenticap_model_amd import autograd as AG
    pr = cProfile.Profile()
    orig = AG._backward
    def prof_backward(*a, **k):
        pr.enable()
        try:
            return orig(*a, **k)
        finally:
            pr.disable()
    AG._backward = prof_backward
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    print('profiled wall/iter %.2f' % ((time.perf_counter() - t0) / 5 * 1e3))
    pstats.Stats(pr).sort_stats('tottime').print_stats(40)
