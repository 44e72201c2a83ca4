#!/usr/bin/env python3
"""Eager XE training iterations (train.xe_train_step, no graphs): merged step chain vs one chain per unroll.
    python tools/r5_ab_eager.py [iterations [batch]]      ISC_PAIR=0|1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth
from insenticap_model_amd.train import xe_train_step

dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
V, R, T = bench.V, bench.R, bench.T
cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev).train()
cap.pair_unrolls = os.environ.get('ISC_PAIR', '1') != '0'
optim, xc, dc = cap.get_optim_criterion(4e-4)
d = synth.make_inputs(B, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=500)
s = synth.make_inputs(80, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=600)
tt = lambda x: torch.from_numpy(x).to(dev)
fact = (None, tt(d['fc_feats']), tt(d['att_feats']), (tt(d['captions']), d['lengths']), tt(d['cpt_words']))
scs = ((tt(s['captions']), s['lengths']), tt(s['cpt_words']), tt(s['senti_words']), tt(s['senti_labels']))
labels = tt(d['senti_labels'])
for _ in range(4):
    xe_train_step(cap, optim, xc, dc, fact, labels, scs, 0.0, 0.1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    xe_train_step(cap, optim, xc, dc, fact, labels, scs, 0.0, 0.1)
torch.cuda.synchronize()
print('eager ms/iter %.3f  pair %s B %d' % ((time.perf_counter() - t0) / iters * 1e3, cap.pair_unrolls, B))
