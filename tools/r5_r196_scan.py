#!/usr/bin/env python3
"""Greedy roll-outs at 196 regions (the reference encoder's 14 x 14 grid) with the gated scan of <= 256-row steps on the
row kernel (walking the regions 36 at a time) vs on the 512-thread region walk: ms per roll-out at B = 8 ... 256."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth, _lib
dev = torch.device('cuda:0')
cap = Captioner(synth.make_idx2word(bench.V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(bench.V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev).eval()
cap.enable_rollout_graphs(False)
lib = _lib.load()
for B in (8, 64, 128, 256):
    inputs, _ = bench.device_inputs(B, 100, dev, regions=196)
    for regions in (256, 36, 256, 36):
        lib.isc_set_rows_scan_regions(regions)
        with torch.no_grad():
            for _ in range(3):
                cap(*inputs, bench.T, 1, mode='rl')
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                cap(*inputs, bench.T, 1, mode='rl')
            torch.cuda.synchronize()
        print('B %4d  row-kernel regions <= %3d: %.3f ms per roll-out' % (B, regions, (time.perf_counter() - t0) / 30 * 1e3))
lib.isc_set_rows_scan_regions(36)
