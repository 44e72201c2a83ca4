#!/usr/bin/env python3
"""Eager XE iteration (one chain per unroll, two streams) on batches sorted by caption length: ragged unroll on / off.
    python tools/ragged_probe.py [iterations [ss_prob]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth
from insenticap_model_amd.train import xe_train_step

dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ss = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
V, R, T = bench.V, bench.R, bench.T


for B in (128, 512, 1024):
    d = synth.sort_by_length(synth.make_inputs(B, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=500))
    s = synth.sort_by_length(synth.make_inputs(80, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=600))
    tt = lambda x: torch.from_numpy(x).to(dev)
    fact = (None, tt(d['fc_feats']), tt(d['att_feats']), (tt(d['captions']), d['lengths']), tt(d['cpt_words']))
    scs = ((tt(s['captions']), s['lengths']), tt(s['cpt_words']), tt(s['senti_words']), tt(s['senti_labels']))
    labels = tt(d['senti_labels'])
    for pair, ragged in ((True, False), (False, False), (False, True)):
        cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
        cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, synth.DEFAULT_SETTINGS).items()})
        cap.to(dev).train()
        cap.pair_unrolls, cap.ragged_unroll = pair, ragged
        optim, xc, dc = cap.get_optim_criterion(4e-4)
        for _ in range(3):
            xe_train_step(cap, optim, xc, dc, fact, labels, scs, ss, 0.1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = xe_train_step(cap, optim, xc, dc, fact, labels, scs, ss, 0.1)
        torch.cuda.synchronize()
        print('ss %.2f  B %4d  merged %-5s ragged %-5s  %.3f ms/iter  xe_loss %.5f  active %.2f' % (
            ss, B, pair, ragged, (time.perf_counter() - t0) / iters * 1e3, float(out['xe_loss']),
            sum(d['lengths']) / (T * B)), flush=True)
        del cap, optim
