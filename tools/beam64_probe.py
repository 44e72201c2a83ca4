#!/usr/bin/env python3
"""Where a batched beam-5 search over 64 images spends its time: device (events around the call) vs the host's share
(staging, graph launch, read-back, caption strings).    python tools/beam64_probe.py [images]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, synth, beam as beam_mod

dev = torch.device('cuda:0')
cap = Captioner(synth.make_idx2word(bench.V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(bench.V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev).eval()
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 64
inputs, _ = bench.device_inputs(n_img, 100, dev)
fc, att, _, sw, lab = inputs
bump = float(os.environ.get('ISC_EOS_BUMP', '0'))          # > 0: push <EOS> so that the batch ends early (trained-model-like)
if bump > 0:
    with torch.no_grad():
        cap.classifier.bias[cap.eos_id] += bump
else:
    cap.eos_id = -7
for gate in (True, False):
    cap.beam_step_gate = gate
    with torch.no_grad():
        for i in range(3):
            cap.sample_batch(fc, att, sw, lab, 5, 1, bench.T)
        entries = [e for e in cap._beam_graphs.values() if isinstance(e, tuple)]
        ngraphs = [len(e[0]) for e in entries]
        fin = beam_mod._Search.finish
        t_fin = [0.0]

        def timed_finish(self):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fin(self)
            t_fin[0] += time.perf_counter() - t0
            return r
        beam_mod._Search.finish = timed_finish
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n, dev_ms = 20, 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            e0.record()
            cap.sample_batch(fc, att, sw, lab, 5, 1, bench.T)
            e1.record()
            torch.cuda.synchronize()
            dev_ms += e0.elapsed_time(e1)
        el = time.perf_counter() - t0
        beam_mod._Search.finish = fin
    print('steps executed %d; ' % cap.last_beam_steps, end='')
    print('gate %s: graphs per search %s; wall %.3f ms per search, between the events %.3f ms, finish() after the device is '
          'idle %.3f ms' % (gate, ngraphs, el / n * 1e3, dev_ms / n, t_fin[0] / n * 1e3))
