#!/usr/bin/env python3
"""Single-image beam-5 searches forced through all 20 steps, for rocprofv3 --kernel-trace --stats; prints wall time per search."""
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
REG = int(os.environ.get('ISC_REGIONS', bench.R))
inputs, _ = bench.device_inputs(16, 100, dev, regions=REG)
fc, att, _, sw, lab = inputs
cap.eos_id = -7
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
with torch.no_grad():
    for i in range(3):
        cap.sample(fc[i], att[i], sw[i], lab[i:i + 1], 5, 1, bench.T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        cap.sample(fc[i % 16], att[i % 16], sw[i % 16], lab[i % 16:i % 16 + 1], 5, 1, bench.T)
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
print('beam-5 full 20-step search, %d regions: %.3f ms per image, %.1f us per step' % (REG, el / n * 1e3, el / n / bench.T * 1e6))
