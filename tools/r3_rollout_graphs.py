#!/usr/bin/env python3
"""Greedy roll-outs at small batches: eager vs enable_rollout_graphs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth
dev = torch.device('cuda:0')
V, R, T = bench.V, bench.R, bench.T
cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev).eval()
for B in (4, 128, 512, 1024):
    inputs, _ = bench.device_inputs(B, 700 + B, dev)
    for mode in ('eager', 'graphs'):
        cap.enable_rollout_graphs(mode == 'graphs')
        with torch.no_grad():
            for _ in range(4):
                out = cap(*inputs, T, 1, mode='rl')
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(10):
                    out = cap(*inputs, T, 1, mode='rl')
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 10)
            ts = []
            for i in range(10):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out = cap(*inputs, T, 1, mode='rl'); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print('B=%d %s: %.3f ms per roll-out (pipelined), single call %.3f ms' % (B, mode, best * 1e3, sorted(ts)[5] * 1e3))
