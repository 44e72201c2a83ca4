#!/usr/bin/env python3
"""Batched beam-5 search over 64 images (BASELINE configs[2]) forced through all 20 steps, for rocprofv3 --kernel-trace
--stats; prints wall time per search.    python tools/profile_beam64.py [searches] [images]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth

dev = torch.device('cuda:0')
cap = Captioner(synth.make_idx2word(bench.V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(bench.V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev).eval()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 64
inputs, _ = bench.device_inputs(n_img, 100, dev)
fc, att, _, sw, lab = inputs
if os.environ.get('ISC_FORCE_FULL', '1') == '1':
    cap.eos_id = -7
cap.enable_beam_graphs(os.environ.get('ISC_BEAM_GRAPHS', '1') == '1')
with torch.no_grad():
    for i in range(3):
        cap.sample_batch(fc, att, sw, lab, 5, 1, bench.T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        cap.sample_batch(fc, att, sw, lab, 5, 1, bench.T)
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
print('beam-5 search over %d images, %d steps: %.3f ms per search, %.1f us per step, %.0f images/s'
      % (n_img, cap.last_beam_steps, el / n * 1e3, el / n / max(1, cap.last_beam_steps) * 1e6, n_img * n / el))
