#!/usr/bin/env python3
"""XE training iterations (BASELINE config 2: B=128, +80 seq2seq rows) for rocprofv3 --kernel-trace --stats.
    python tools/profile_xe.py [iterations [batch]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()

dev = torch.device('cuda:0')
from insenticap_model_amd import Captioner, synth
cap = Captioner(synth.make_idx2word(bench.V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(bench.V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev)
print(bench.bench_xe_train(cap, dev, 0, 1, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 4,
                           B=int(sys.argv[2]) if len(sys.argv) > 2 else 128))
